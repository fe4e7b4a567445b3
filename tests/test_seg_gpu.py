"""Bit-exact parity of the HIP segmentation kernels (through the C ABI) with the golden
fragments produced by the reference post/ws.py and with the C oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobby(rng, shape, sigma):
    from scipy.ndimage import gaussian_filter
    a = gaussian_filter(rng.random((3,) + shape), sigma=(0,) + sigma)
    a = (a - a.min()) / (a.max() - a.min())
    return (a * 255).astype(np.uint8)


def test_fragments_bit_exact_vs_reference_goldens(golden_dir):
    from bootstrapper_amd.post.ws import watershed_from_affinities
    d = np.load(os.path.join(golden_dir, "ws_cases.npz"))
    names = sorted({k.split("/")[0] for k in d.files})
    n_checked = 0
    for name in names:
        xy, msd, max_id = (int(v) for v in d[name + "/meta"])
        if not xy:
            continue  # 3-D flood mode: not implemented on the device (raises, tested below)
        affs = torch.from_numpy(d[name + "/affs"]).cuda()
        frags, mx = watershed_from_affinities(affs, fragments_in_xy=True, min_seed_distance=msd)
        assert mx == max_id, name
        assert np.array_equal(frags.cpu().numpy().astype(np.uint64), d[name + "/frags"].astype(np.uint64)), name
        n_checked += 1
    assert n_checked >= 12


def test_fragments_3d_mode_bit_exact(golden_dir):
    """fragments_in_xy=False (post/ws.py:97-110): the reference golden and larger volumes against the oracle, including a
    volume without any background voxel (scipy's EDT quirk) and one that is all background."""
    from bootstrapper_amd.post.ws import watershed_from_affinities
    from oracle import seg_ref as S
    d = np.load(os.path.join(golden_dir, "ws_cases.npz"))
    n3 = 0
    for name in sorted({k.split("/")[0] for k in d.files}):
        xy, msd, max_id = (int(v) for v in d[name + "/meta"])
        if xy:
            continue
        frags, mx = watershed_from_affinities(torch.from_numpy(d[name + "/affs"]).cuda(), fragments_in_xy=False, min_seed_distance=msd)
        assert mx == max_id and np.array_equal(frags.cpu().numpy().astype(np.uint64), d[name + "/frags"].astype(np.uint64)), name
        n3 += 1
    assert n3 >= 1
    rng = np.random.default_rng(12)
    cases = [(_blobby(rng, (20, 48, 40), (2, 3, 3)), 5), (_blobby(rng, (9, 33, 70), (1, 2, 2)), 3),
             (np.full((3, 6, 20, 20), 255, np.uint8), 4), (np.zeros((3, 4, 10, 12), np.uint8), 4)]
    from bootstrapper_amd.post.engine import SegEngine
    for affs, msd in cases:
        ref, ref_max = S.ws_fragments_u8(affs, False, msd)
        frags, mx = watershed_from_affinities(torch.from_numpy(affs).cuda(), fragments_in_xy=False, min_seed_distance=msd)   # flood on the host
        assert mx == ref_max and np.array_equal(frags.cpu().numpy().astype(np.uint64), ref)
        # the block pipeline's lanes keep the flood on the device (asynchronous): the same fragments
        f2, m2 = watershed_from_affinities(torch.from_numpy(affs).cuda(), fragments_in_xy=False, min_seed_distance=msd,
                                           engine=SegEngine(affs.shape[1:], 0, host_flood=False))
        assert m2 == ref_max and np.array_equal(f2.cpu().numpy().astype(np.uint64), ref)


@pytest.mark.parametrize("shape,sigma,msd", [((16, 128, 128), (1, 4, 4), 10), ((5, 160, 160), (1, 6, 6), 10),
                                             ((3, 97, 131), (1, 2, 2), 7), ((2, 200, 64), (0, 1, 1), 10)])
def test_fragments_bit_exact_vs_oracle(shape, sigma, msd):
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[1])
    affs = _blobby(rng, shape, sigma)
    ref, ref_max = S.ws_fragments_u8(affs, True, msd)
    eng = SegEngine(shape)
    frags, mx = eng.ws_fragments(torch.from_numpy(affs).cuda(), True, msd)
    assert int(mx.item()) == ref_max
    assert np.array_equal(frags.cpu().numpy().astype(np.uint64), ref)


@pytest.mark.parametrize("shape,sigma,msd,kind", [
    ((2, 300, 260), (0, 2, 2), 10, "blobs"),    # slices too large for the LDS form of the marker kernel: global scratch, three-array flood
    ((3, 40, 6), (0, 1, 1), 10, "blobs"),       # filter window wider than the rows: the general reflection
    ((2, 9, 1), (0, 1, 0), 3, "blobs"),         # one column (no reciprocal for the row index)
    ((2, 1, 23), (0, 0, 1), 3, "blobs"),        # one row
    ((2, 20, 30), None, 10, "full"),            # no background voxel at all (scipy's distance transform as if one sat at (-1, 0))
    ((2, 20, 30), None, 10, "empty"),           # nothing inside the mask
    ((2, 160, 160), None, 10, "half")])         # a straight edge: plateaus of equal distance, long runs of tied seeds
def test_fragments_of_odd_slices_bit_exact_vs_oracle(shape, sigma, msd, kind):
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[1] * 3 + shape[2])
    if kind == "blobs":
        affs = _blobby(rng, shape, sigma)
    elif kind == "full":
        affs = np.full((3,) + shape, 255, np.uint8)
    elif kind == "empty":
        affs = np.zeros((3,) + shape, np.uint8)
    else:
        affs = np.zeros((3,) + shape, np.uint8)
        affs[:, :, :, shape[2] // 3:] = 200
    ref, ref_max = S.ws_fragments_u8(affs, True, msd)
    frags, mx = SegEngine(shape).ws_fragments(torch.from_numpy(affs).cuda(), True, msd)
    assert int(mx.item()) == ref_max
    assert np.array_equal(frags.cpu().numpy().astype(np.uint64), ref)


def test_three_array_flood_in_subprocess():
    """BSMI_FLOOD_COMPACT=0 (read once per process): the flood on the mask / label / distance arrays instead of the 32-bit record"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_seg_gpu.py"), "-x", "-q", "-k",
                        "test_fragments_bit_exact_vs_oracle or test_fragments_white_noise_heap_spill"],
                       env=dict(os.environ, BSMI_FLOOD_COMPACT="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fragments_white_noise_heap_spill():
    """White-noise affinities: thousands of one-voxel seeds per slice -> exercises large heaps."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(2)
    affs = rng.integers(0, 256, size=(3, 2, 160, 160), dtype=np.uint8)
    ref, ref_max = S.ws_fragments_u8(affs, True, 2)
    eng = SegEngine(affs.shape[1:])
    frags, mx = eng.ws_fragments(torch.from_numpy(affs).cuda(), True, 2)
    assert int(mx.item()) == ref_max
    assert np.array_equal(frags.cpu().numpy().astype(np.uint64), ref)


@pytest.mark.parametrize("shape,sigma", [((12, 64, 64), (1, 3, 3)), ((32, 128, 128), (1, 4, 4)), ((4, 50, 70), (0, 1, 1))])
def test_agglomeration_bit_exact_vs_oracle(shape, sigma):
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[0])
    affs = _blobby(rng, shape, sigma)
    frags, _ = S.ws_fragments_u8(affs, True, 10)
    thresholds = [0.2, 0.35, 0.5]
    ref = S.agglomerate_mean_u8(affs, frags, thresholds)
    eng = SegEngine(shape)
    segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), thresholds)
    eng.status()
    got = segs.cpu().numpy().astype(np.uint64)
    for t in range(len(thresholds)):
        assert np.array_equal(got[t], ref[t]), f"threshold {thresholds[t]}"
    # the segmentations are non-trivial on this input
    assert len(np.unique(ref[2])) < len(np.unique(frags))


def test_agglomeration_edge_cases():
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    eng = SegEngine((4, 16, 16))
    # no fragments at all; one fragment; sparse ids with gaps
    affs = np.full((3, 4, 16, 16), 200, dtype=np.uint8)
    for frags in (np.zeros((4, 16, 16), np.uint64), np.full((4, 16, 16), 7, np.uint64)):
        ref = S.agglomerate_mean_u8(affs, frags, [0.5])
        segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), [0.5])
        eng.status()
        assert np.array_equal(segs[0].cpu().numpy().astype(np.uint64), ref[0])
    frags = np.zeros((4, 16, 16), np.uint64)
    frags[:, :8, :8] = 3; frags[:, :8, 8:] = 900; frags[:, 8:, :8] = 41; frags[:, 8:, 8:] = 40
    affs = np.random.default_rng(0).integers(0, 256, size=(3, 4, 16, 16), dtype=np.uint8)
    for thr in ([0.0], [0.3, 0.6, 0.9], [2.0]):
        ref = S.agglomerate_mean_u8(affs, frags, thr)
        segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), thr)
        eng.status()
        for t in range(len(thr)):
            assert np.array_equal(segs[t].cpu().numpy().astype(np.uint64), ref[t])


def test_mirror_api_generator():
    """post/ws.py and waterz.agglomerate as the reference's call sites use them: float affinities that are exactly
    u8 / 255 (post/watershed.py:259-262), return_seeds, fragments updated in place, and the blockwise call with
    thresholds [0, 1.0], discretize_queue=256, merge history and region graph (waterz_agglom.py:131-170)."""
    from bootstrapper_amd.post.ws import watershed_from_affinities
    from bootstrapper_amd.post.waterz import agglomerate
    from bootstrapper_amd.post.merge_tree import MergeTree
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(9)
    affs = _blobby(rng, (6, 48, 48), (1, 3, 3))
    a = torch.from_numpy(affs).cuda()
    frags, n = watershed_from_affinities(a, fragments_in_xy=True, min_seed_distance=10)
    ref_frags, ref_n, ref_seeds = S.ws_fragments_u8(affs, True, 10, return_seeds=True)
    assert n == ref_n
    # what the reference's drivers hand over: float32 u8 / 255, max_affinity_value left at 1.0; seeds on request
    af = affs.astype(np.float32) / np.float32(255.0)
    f2, n2, seeds = watershed_from_affinities(af, fragments_in_xy=True, return_seeds=True, min_seed_distance=10)
    assert n2 == ref_n and torch.equal(f2, frags)
    assert np.array_equal(seeds.cpu().numpy().view(np.uint64), ref_seeds)
    f3, n3, seeds3 = watershed_from_affinities(torch.from_numpy(af).cuda().double(), fragments_in_xy=False, return_seeds=True, min_seed_distance=5)
    r3, rn3, rs3 = S.ws_fragments_u8(affs, False, 5, return_seeds=True)
    assert n3 == rn3 and np.array_equal(f3.cpu().numpy().view(np.uint64), r3) and np.array_equal(seeds3.cpu().numpy().view(np.uint64), rs3)
    # anything else -- shifted / continuous floats, more than three channels -- is reduced to ws.py's own boundary mask first
    # (ws.py:64,77 and :100: the 3-D mode takes the mean over ALL channels); the oracle is fed that mask, computed with numpy
    shifted = af + rng.normal(0, 0.05, af.shape).astype(np.float32)
    f4, n4 = watershed_from_affinities(shifted, fragments_in_xy=True, min_seed_distance=4)
    m4 = (0.5 * (shifted[-1] + shifted[-2]) > 0.5).astype(np.uint8) * 255
    r4, rn4 = S.ws_fragments_u8(np.stack([m4] * 3), True, 4)
    assert n4 == rn4 and np.array_equal(f4.cpu().numpy().view(np.uint64), r4)
    five = np.concatenate([shifted, shifted[:2] * np.float32(0.8)]).astype(np.float32)
    f5, n5 = watershed_from_affinities(five, fragments_in_xy=False, min_seed_distance=5)
    m5 = (np.mean(five, axis=0) > 0.5).astype(np.uint8) * 255
    r5, rn5 = S.ws_fragments_u8(np.stack([m5] * 3), False, 5)
    assert n5 == rn5 and np.array_equal(f5.cpu().numpy().view(np.uint64), r5)
    assert not np.array_equal(m5, (np.mean(five[-3:], axis=0) > 0.5).astype(np.uint8) * 255)    # not the last three channels
    u5 = (np.clip(five, 0, 1) * 255).astype(np.uint8)
    f6, n6 = watershed_from_affinities(u5, max_affinity_value=255, fragments_in_xy=False, min_seed_distance=5)
    m6 = (np.mean(u5, axis=0) > 0.5 * 255).astype(np.uint8) * 255
    r6, rn6 = S.ws_fragments_u8(np.stack([m6] * 3), False, 5)
    assert n6 == rn6 and np.array_equal(f6.cpu().numpy().view(np.uint64), r6)
    ref = S.agglomerate_mean_u8(affs, ref_frags, [0.2, 0.5])
    work = frags.clone()
    for seg, r in zip(agglomerate(af, [0.2, 0.5], fragments=work), ref):
        assert seg is work and np.array_equal(seg.cpu().numpy().astype(np.uint64), r)      # in place, the same array every time
    # the histogram-quantile scorers of the non-blockwise path (post/watershed.py:230-243): device region graph + histograms,
    # host merge loop, device relabel -- against the oracle; a scorer outside the reference's table is refused
    from bootstrapper_amd.post.waterz import MERGE_FUNCTIONS
    assert len(MERGE_FUNCTIONS) == 11 and MERGE_FUNCTIONS["mean"].startswith("OneMinus<MeanAffinity")
    for name, (q, initmax) in (("hist_quant_50", (50, False)), ("hist_quant_10_initmax", (10, True)), ("hist_quant_90", (90, False)),
                               ("hist_quant_75_initmax", (75, True))):
        want = S.agglomerate_hist_u8(affs, ref_frags, [0.2, 0.5, 0.8], q, initmax)
        work = frags.clone()
        got = [seg.cpu().numpy().astype(np.uint64) for seg in agglomerate(a, [0.2, 0.5, 0.8], fragments=work, scoring_function=MERGE_FUNCTIONS[name])]
        for g, w in zip(got, want):
            assert np.array_equal(g, w), name
        assert len(np.unique(want[2])) < len(np.unique(want[0])) <= len(np.unique(ref_frags))
    with pytest.raises(NotImplementedError):
        next(agglomerate(a, [0.2], fragments=frags, scoring_function="OneMinus<MaxAffinity<RegionGraphType, ScoreValue>>"))
    with pytest.raises(NotImplementedError):   # merge history needs the discretized queue, which only the mean scorer has
        next(agglomerate(a, [0.2], fragments=frags, scoring_function=MERGE_FUNCTIONS["hist_quant_50"], discretize_queue=256, return_merge_history=True))
    # the blockwise call (numpy fragments, modified in place like waterz does)
    e_ref, s_ref, m_ref, ms_ref = S.rag_merge_scores_u8(affs, ref_frags, 1.0, 256)
    fr_np = ref_frags.copy()
    gen = agglomerate(a, thresholds=[0, 1.0], fragments=fr_np, discretize_queue=256, return_merge_history=True, return_region_graph=True)
    seg0, hist0, rag0 = next(gen)
    assert seg0 is fr_np and hist0 == [] and np.array_equal(seg0, ref_frags)
    assert [(g["u"], g["v"]) for g in rag0] == [tuple(x) for x in e_ref.tolist()]
    seg1, hist1, rag1 = next(gen)
    for _ in gen:
        pass
    assert [(h["a"], h["b"]) for h in hist1] == [tuple(x) for x in m_ref.tolist()] and all(h["c"] == h["a"] for h in hist1)
    np.testing.assert_array_equal(np.array([h["score"] for h in hist1], np.float32), ms_ref)
    nodes = np.unique(ref_frags)
    mt = MergeTree(nodes[nodes > 0])
    for h in hist1:
        mt.merge(h["a"], h["b"], h["c"], h["score"])
    got = mt.find_merges(e_ref[:, 0], e_ref[:, 1])
    assert np.array_equal(np.isnan(got), np.isnan(s_ref)) and np.array_equal(got[~np.isnan(got)].astype(np.float32), s_ref[~np.isnan(s_ref)])
    # the segmentation at 1.0 is the partition the history implies, and the remaining region graph only links distinct segments
    assert len(np.unique(seg1)) == len(np.unique(ref_frags)) - len(hist1)
    assert all(g["u"] < g["v"] and g["score"] >= 0 for g in rag1)
    ids1 = set(np.unique(seg1).tolist())
    assert all(g["u"] in ids1 and g["v"] in ids1 for g in rag1)
    # initial region-graph scores are the mean-affinity scores of the initial edges
    eng = SegEngine((6, 48, 48))
    eng.rag_merge_scores(a, torch.from_numpy(ref_frags.view(np.int64)).cuda(), 1e-30, 256)
    sums, counts = eng.rag_edge_stats(len(e_ref))
    np.testing.assert_array_equal(np.array([g["score"] for g in rag0], np.float32),
                                  (np.float32(1.0) - (sums / (255.0 * counts)).astype(np.float32)).astype(np.float32))


@pytest.mark.parametrize("shape,crop,filt,min_size", [
    ((12, 96, 96), ((2, 16, 16), (8, 64, 64)), 0.5, 64),
    ((6, 130, 70), ((0, 0, 0), (6, 130, 70)), 0.45, 0),
    ((9, 64, 80), ((1, 3, 5), (7, 50, 61)), 0.0, 30),
    ((4, 40, 40), ((0, 0, 0), (4, 40, 40)), 0.0, 0),
])
def test_fragment_postprocess_bit_exact_vs_oracle(shape, crop, filt, min_size):
    """filter_avg_fragments + remove_small_objects + crop + measure.label + offset (watershed_frags.py:181-224)."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[2])
    affs = _blobby(rng, shape, (1, 3, 3))
    ref_frags, _ = S.ws_fragments_u8(affs, True, 5)
    ref_f = S.filter_fragments_u8(affs, ref_frags, filt, min_size)
    (oz, oy, ox), (cd, ch, cw) = crop
    ref_lab, ref_n = S.label26(np.ascontiguousarray(ref_f[oz:oz + cd, oy:oy + ch, ox:ox + cw]))
    offset = 7_000_000_000
    ref_lab = np.where(ref_lab > 0, ref_lab + np.uint64(offset), np.uint64(0))

    eng = SegEngine(shape)
    a = torch.from_numpy(affs).cuda()
    frags, _ = eng.ws_fragments(a, True, 5)
    lab, num = eng.postprocess_fragments(a, frags, filt, min_size, crop[0], crop[1], offset)
    eng.status()
    assert int(num.item()) == ref_n
    assert np.array_equal(frags.cpu().numpy().astype(np.uint64), ref_f)        # filtered in place
    assert np.array_equal(lab.cpu().numpy().astype(np.uint64), ref_lab)

    size, sums = eng.label_stats(lab, offset, ref_n)
    torch.cuda.synchronize()
    l = lab.cpu().numpy() - offset
    idx = np.indices(l.shape)
    for k in rng.choice(ref_n, size=min(ref_n, 20), replace=False):
        m = l == k + 1
        assert int(size[k]) == int(m.sum())
        assert [int(v) for v in sums[k]] == [int(idx[d][m].sum()) for d in range(3)]


def test_fragment_postprocess_serpentine_components():
    """Union-find stress: one-voxel-wide spirals and equal ids split by the crop into several components."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(3)
    shape = (5, 64, 64)
    frags = rng.integers(0, 4, size=shape).astype(np.uint64)      # few ids -> percolating, tangled components
    frags[2] = 0
    frags[3, ::2, :] = 9
    frags[3, 1::2, :] = 0
    frags[3, 1::4, 0] = 9
    frags[3, 3::4, -1] = 9                                          # a serpentine of id 9 through the slice
    affs = np.full((3,) + shape, 200, np.uint8)
    ref_lab, ref_n = S.label26(frags)
    eng = SegEngine(shape)
    f = torch.from_numpy(frags.astype(np.int64)).cuda()
    lab, num = eng.postprocess_fragments(torch.from_numpy(affs).cuda(), f, 0.0, 0)
    eng.status()
    assert int(num.item()) == ref_n
    assert np.array_equal(lab.cpu().numpy().astype(np.uint64), ref_lab)


def _as_u64(t):
    return t.cpu().numpy().view(np.uint64) if t.dtype == torch.int64 else t.cpu().numpy()


@pytest.mark.parametrize("shape,sigma,msd,bins,id_base", [
    ((8, 96, 96), (1, 3, 3), 5, 256, 0),
    ((6, 128, 80), (1, 2, 2), 4, 256, 3 * 2_097_152),        # ids of a later block
    ((12, 64, 64), (1, 4, 4), 6, 16, (1 << 40) + 5),           # coarse bins, ids beyond 32 bits
    ((3, 50, 70), (0, 1, 1), 3, 1, 0),                         # one bin: pure FIFO
])
def test_rag_merge_scores_bit_exact_vs_oracle(shape, sigma, msd, bins, id_base):
    """waterz_agglom.py:106-170: initial RAG, merge history and per-edge merge scores."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[1] + bins)
    affs = _blobby(rng, shape, sigma)
    frags, _ = S.ws_fragments_u8(affs, True, msd)
    # scatter the ids over two id ranges the way neighbouring blocks do
    big = frags > np.median(frags[frags > 0])
    frags = np.where(frags > 0, frags + np.uint64(id_base) + np.where(big, np.uint64(2_097_152), np.uint64(0)), np.uint64(0))
    e_ref, s_ref, m_ref, ms_ref = S.rag_merge_scores_u8(affs, frags, 1.0, bins)
    eng = SegEngine(shape)
    e, s, m, ms = eng.rag_merge_scores(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.view(np.int64)).cuda(),
                                       1.0, bins, return_merges=True)
    assert np.array_equal(_as_u64(e), e_ref)
    assert np.array_equal(_as_u64(m), m_ref)
    assert np.array_equal(ms.cpu().numpy().view(np.uint32), ms_ref.view(np.uint32))
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))     # NaNs included


@pytest.mark.parametrize("shape,sigma,msd,bins,thr,levels", [
    ((12, 96, 96), (1, 3, 3), 5, 256, 1.0, 0), ((4, 160, 160), (1, 5, 5), 10, 256, 1.0, 0),
    ((6, 80, 80), (1, 3, 3), 5, 256, 0.45, 0), ((8, 64, 72), (1, 2, 2), 3, 16, 1.0, 0),
    # saturated / coarsely quantised affinities: parallel edges tie, the tie rule decides the order (tests/test_host_logic.py)
    ((8, 80, 80), (1, 2, 2), 3, 256, 1.0, 3), ((8, 64, 72), (1, 2, 2), 3, 16, 1.0, 2), ((6, 96, 96), (1, 3, 3), 5, 256, 1.0, 6)])
def test_rag_scores_on_the_host_equal_the_device_loop_and_the_oracle(shape, sigma, msd, bins, thr, levels):
    """The block pipeline's edge scoring: the device exports the region graph (bsmi_rag_graph_u8), the merge loop and the
    merge-tree look-ups run on host threads (bsmi_rag_merge_scores_host) -- the same edges and bit for bit the same scores as the
    device loop (bsmi_rag_merge_scores_u8) and the oracle, several graphs in one call."""
    from bootstrapper_amd.post.engine import SegEngine, rag_merge_scores_host
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[1] + bins)
    graphs, refs = [], []
    eng = SegEngine(shape)
    for g in range(3):
        affs = _blobby(rng, shape, sigma)
        if levels:
            q = np.clip((affs.astype(np.float64) / 255 - 0.5) * 2.5 + 0.5, 0, 1)
            affs = (np.round(q * (levels - 1)) / (levels - 1) * 255).astype(np.uint8)
        frags, _ = S.ws_fragments_u8(affs, True, msd)
        big = frags > np.median(frags[frags > 0])
        frags = np.where(frags > 0, frags + np.uint64(1000 * g) + np.where(big, np.uint64(2_097_152), np.uint64(0)), np.uint64(0))
        e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, thr, bins)
        a, f = torch.from_numpy(affs).cuda(), torch.from_numpy(frags.view(np.int64)).cuda()
        e_dev, s_dev = eng.rag_merge_scores(a, f, thr, bins)
        assert np.array_equal(_as_u64(e_dev), e_ref) and np.array_equal(s_dev.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
        cap = len(e_ref) + 7
        edges = torch.zeros((cap, 2), dtype=torch.int64, device="cuda")
        sums = torch.zeros(cap, dtype=torch.int64, device="cuda")
        cnts = torch.zeros(cap, dtype=torch.int32, device="cuda")
        counts = torch.zeros(4, dtype=torch.int64, device="cuda")
        eng.rag_graph_async(a, f, edges, sums, cnts, counts)
        eng.status()
        assert int(counts[0]) == len(e_ref)
        graphs.append((edges.cpu().numpy(), sums.cpu().numpy(), cnts.cpu().numpy()))
        refs.append((e_ref, s_ref))
    cap = max(len(g[0]) for g in graphs)
    E = np.zeros((3, cap, 2), np.int64); Sm = np.zeros((3, cap), np.int64); Cn = np.zeros((3, cap), np.int32)
    for g, (e, s, c) in enumerate(graphs):
        E[g, :len(e)], Sm[g, :len(s)], Cn[g, :len(c)] = e, s, c
    ne = np.array([len(r[0]) for r in refs])
    sc = rag_merge_scores_host(ne, E, Sm, Cn, thr, bins, threads=2)
    for g, (e_ref, s_ref) in enumerate(refs):
        assert np.array_equal(E[g, :ne[g]].view(np.uint64), e_ref)
        assert np.array_equal(sc[g, :ne[g]].view(np.uint32), s_ref.view(np.uint32))     # NaNs included
    # a too small buffer is reported with the count the block needs, as by the device loop's call
    small = torch.zeros((4, 2), dtype=torch.int64, device="cuda")
    eng.rag_graph_async(a, f, small, torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda"), counts)
    with pytest.raises(Exception):
        eng.status()
    assert int(counts[0]) == ne[2]


def test_rag_merge_scores_threshold_leaves_unmerged_edges():
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(11)
    shape = (6, 80, 80)
    affs = _blobby(rng, shape, (1, 3, 3))
    frags, _ = S.ws_fragments_u8(affs, True, 5)
    e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, 0.45, 256)
    assert np.isnan(s_ref).any() and (~np.isnan(s_ref)).any()
    eng = SegEngine(shape)
    e, s = eng.rag_merge_scores(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.view(np.int64)).cuda(), 0.45, 256)
    assert np.array_equal(_as_u64(e), e_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


def test_lut_relabel():
    from bootstrapper_amd.post.engine import lut_relabel
    rng = np.random.default_rng(0)
    keys = np.unique(rng.integers(1, 1 << 45, size=5000)).astype(np.int64)
    vals = rng.integers(1, 1 << 45, size=keys.size).astype(np.int64)
    lab = rng.choice(np.concatenate([keys, [0, 7, (1 << 50) + 1]]), size=(7, 33, 41)).astype(np.int64)
    lut = dict(zip(keys.tolist(), vals.tolist()))
    ref = np.vectorize(lambda v: lut.get(v, v))(lab)
    out = lut_relabel(torch.from_numpy(lab).cuda(), torch.from_numpy(keys), torch.from_numpy(vals))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_lut_relabel_multi_equals_single_columns():
    """every threshold's LUT applied in one pass (runs of equal ids share a look-up) = one lut_relabel per column"""
    from bootstrapper_amd.post.engine import lut_relabel, lut_relabel_multi
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(1, 1 << 45, size=3000)).astype(np.int64)
    vals = rng.integers(1, 1 << 45, size=(3, keys.size)).astype(np.int64)
    # compact regions (long runs along x), zeros, ids without a key, a length that is no multiple of the run
    base = rng.choice(np.concatenate([keys, [0, 0, 7, (1 << 50) + 1]]), size=(5, 37, 9)).astype(np.int64)
    lab = np.repeat(base, 7, axis=2)[:, :, :61]
    out = lut_relabel_multi(torch.from_numpy(lab).cuda(), torch.from_numpy(keys), torch.from_numpy(vals))
    for t in range(3):
        ref = lut_relabel(torch.from_numpy(lab).cuda(), torch.from_numpy(keys), torch.from_numpy(vals[t]))
        assert torch.equal(out[t], ref)
    one = lut_relabel_multi(torch.from_numpy(lab).cuda(), torch.from_numpy(keys), torch.from_numpy(vals[:1]))
    assert torch.equal(one[0], out[0])
    none = lut_relabel_multi(torch.from_numpy(lab).cuda(), torch.zeros(0, dtype=torch.int64), torch.zeros((2, 0), dtype=torch.int64))
    assert torch.equal(none[0], torch.from_numpy(lab).cuda()) and torch.equal(none[1], none[0])


def test_cc_affs_bit_exact_vs_reference_goldens_and_oracle(golden_dir):
    """`bs segment --cc` labelling: goldens produced by the reference post/cc.py, then larger volumes against the oracle,
    then debris removal."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    d = np.load(os.path.join(golden_dir, "cc_cases.npz"))
    for name in sorted({k.split("/")[0] for k in d.files}):
        affs, thr = d[name + "/affs"], float(d[name + "/thr"])
        eng = SegEngine(affs.shape[1:])
        frags, seg, num = eng.cc_affs(torch.from_numpy(affs).cuda(), thr, 0)
        eng.status()
        assert np.array_equal(frags.cpu().numpy().astype(np.uint32), d[name + "/seg"]), name
        assert torch.equal(frags, seg) and int(num.item()) == int(d[name + "/seg"].max())
    rng = np.random.default_rng(8)
    for shape, sigma, thr, debris in [((20, 96, 80), (1, 2, 2), 0.5, 40), ((7, 130, 70), (0, 1, 1), 0.62, 5), ((33, 40, 40), (1, 3, 3), 0.45, 0)]:
        affs = _blobby(rng, shape, sigma)
        ref, n = S.cc_affs_u8(affs, thr)
        eng = SegEngine(shape)
        frags, seg, num = eng.cc_affs(torch.from_numpy(affs).cuda(), thr, debris)
        eng.status()
        assert int(num.item()) == n and np.array_equal(frags.cpu().numpy().astype(np.uint32), ref)
        counts = np.bincount(ref.ravel())
        keep = counts >= debris
        keep[0] = False
        want = np.where(keep[ref], ref, 0)
        assert np.array_equal(seg.cpu().numpy().astype(np.uint32), want)


def test_merge_rule_hand_built_graphs():
    """The two hand-built graphs of tests/test_oracle_seg.py on which waterz's shared-neighbour rule (the dearer edge is
    merged into the cheaper one, which keeps its place in the queue) and the rule this engine had before give different
    segmentations; both queues, both forms of the merge loop are covered by the variant run of this file."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    from tests.test_oracle_seg import _three_node_case
    eng = SegEngine((2, 8, 8))
    affs, frags = _three_node_case()
    segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), [0.5, 0.6])
    eng.status()
    assert segs[0].cpu().numpy().tolist() == [[[1, 1], [3, 3]]]
    assert segs[1].cpu().numpy().tolist() == [[[1, 1], [1, 1]]]
    e, s, m, ms = eng.rag_merge_scores(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.view(np.int64)).cuda(), 0.6, 256,
                                       return_merges=True)
    assert m.cpu().numpy().tolist() == [[1, 2], [1, 3]]
    frags = np.array([[[1, 2, 2], [3, 3, 4], [1, 1, 4]]], dtype=np.uint64)
    affs = np.zeros((3, 1, 3, 3), dtype=np.uint8)
    affs[2, 0, 0, 1] = 240; affs[1, 0, 1, 0] = 51; affs[1, 0, 1, 1] = 230; affs[2, 0, 1, 2] = 128
    affs[1, 0, 2, 0] = 51; affs[1, 0, 2, 1] = 230
    ref = S.agglomerate_mean_u8(affs, frags, [0.55])[0]
    segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), [0.55])
    eng.status()
    assert np.array_equal(segs[0].cpu().numpy().astype(np.uint64), ref)
    assert ref[0, 1, 2] == 4 and len(np.unique(ref)) == 2


def test_tie_rich_agglomeration_bit_exact_vs_oracle():
    """Few affinity levels: many equal scores and many shared neighbours with equal stored scores (the branch where the
    b-side edge survives)."""
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    rng = np.random.default_rng(23)
    shape = (6, 40, 40)
    eng = SegEngine(shape)
    for case in range(4):
        affs = rng.choice(np.array([0, 64, 128, 191, 255], dtype=np.uint8), size=(3,) + shape)
        frags = np.kron(rng.integers(0 if case % 2 else 1, 60, size=(3, 10, 10)), np.ones((2, 4, 4), dtype=np.int64)).astype(np.uint64)
        thr = [0.3, 0.5, 0.7]
        ref = S.agglomerate_mean_u8(affs, frags, thr)
        segs = eng.agglomerate_mean(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.astype(np.int64)).cuda(), thr)
        eng.status()
        for t in range(3):
            assert np.array_equal(segs[t].cpu().numpy().astype(np.uint64), ref[t]), (case, t)
        for bins in (256, 16):
            e_ref, s_ref, m_ref, ms_ref = S.rag_merge_scores_u8(affs, frags, 0.8, bins)
            e, s, m, ms = eng.rag_merge_scores(torch.from_numpy(affs).cuda(), torch.from_numpy(frags.view(np.int64)).cuda(), 0.8, bins,
                                               return_merges=True)
            assert np.array_equal(e.cpu().numpy().view(np.uint64), e_ref)
            assert np.array_equal(m.cpu().numpy().view(np.uint64), m_ref), (case, bins)
            np.testing.assert_array_equal(ms.cpu().numpy(), ms_ref)
            np.testing.assert_array_equal(s.cpu().numpy(), s_ref)


def test_overflow_of_an_earlier_call_is_remembered():
    """An overflow is reported by the next status() even when later calls on the same handle went well (the per-call flag
    word is reset by every call; the pipeline checks once per lane after many blocks)."""
    from bootstrapper_amd.post.engine import SegEngine
    from bootstrapper_amd._lib import BsmiError
    shape = (2, 8, 8)
    eng = SegEngine(shape)
    affs = torch.full((3,) + shape, 200, dtype=torch.uint8, device="cuda")
    bad = torch.full(shape, 10_000, dtype=torch.int64, device="cuda")     # id beyond the direct-address table of this workspace
    good = torch.ones(shape, dtype=torch.int64, device="cuda")
    eng.agglomerate_mean(affs, bad, [0.5])
    eng.agglomerate_mean(affs, good, [0.5])
    with pytest.raises(BsmiError):
        eng.status()
    eng.agglomerate_mean(affs, good, [0.5])
    eng.status()                                                          # cleared by the failed check
