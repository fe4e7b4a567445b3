"""Host-side logic against golden vectors produced by running the reference
(tools/gen_goldens_host.py).  CPU only."""
import json
import math
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "host_cases.json")) as f:
        return json.load(f)


def test_build_name_and_fmt(cases):
    from bootstrapper_amd.post.naming import build_name, fmt
    for c in cases["naming"]:
        assert build_name(c["params"]) == c["name"]
    for c in cases["fmt"]:
        assert fmt(c["value"]) == c["fmt"]


def test_get_seg_config(cases, tmp_path):
    from bootstrapper_amd.segment import get_seg_config
    for i, c in enumerate(cases["seg_config"]):
        p = tmp_path / f"c{i}.toml"
        p.write_text(c["toml"])
        kw = dict(c["kwargs"])
        if "param" in kw:
            kw["param"] = tuple(kw["param"])
        if c.get("error"):
            with pytest.raises(ValueError) as e:
                get_seg_config(str(p), c["method"], **kw)
            assert str(e.value) == c["message"]
        else:
            assert get_seg_config(str(p), c["method"], **kw) == c["config"]


def test_merge_tree(cases):
    from bootstrapper_amd.post.merge_tree import MergeTree
    for c in cases["merge_tree"]:
        mt = MergeTree(c["leaves"])
        for a, b, t, s in c["merges"]:
            mt.merge(a, b, t, s)
        got = mt.find_merges(c["us"], c["vs"])
        for g, ref in zip(got, c["scores"]):
            assert (ref is None and math.isnan(g)) or (ref is not None and g == ref)
    assert MergeTree([1, 2]).find_merge(1, 2) is None


def test_train_host_helpers(tmp_path):
    """setup_train's checks (train.py:13-120), AddAffinities / BalanceLabels restatements, default initialisation keys."""
    import torch
    from bootstrapper_amd import train as T
    cfg = tmp_path / "t.toml"
    cfg.write_text('setup_dir = "x"\nmax_iterations = 3\nsamples = []\n')
    with pytest.raises(ValueError, match="No training samples provided"):
        T.setup_train(str(cfg))
    cfg.write_text(f'setup_dir = "x"\nmax_iterations = 3\n[[samples]]\nraw = "{tmp_path}/nope"\nlabels = "{tmp_path}/nope"\n')
    with pytest.raises(ValueError, match="Raw dataset path .* does not exist"):
        T.setup_train(str(cfg))

    from oracle import train_ref as TR
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 4, size=(5, 9, 8))
    nhood = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0]]
    affs, mask = TR.affinities_from_labels(labels, nhood)
    for e, (dz, dy, dx) in enumerate(nhood):
        for z, y, x in [(0, 0, 0), (2, 4, 3), (4, 8, 7), (1, 2, 0), (3, 0, 5)]:
            zz, yy, xx = z + dz, y + dy, x + dx
            inside = 0 <= zz < 5 and 0 <= yy < 9 and 0 <= xx < 8
            want = float(inside and labels[z, y, x] == labels[zz, yy, xx] and labels[z, y, x] > 0)
            assert float(affs[e, z, y, x]) == want and float(mask[e, z, y, x]) == float(inside)
    w = TR.balance_labels(affs, mask)
    frac = float((affs * mask).sum() / mask.sum())
    frac = min(max(frac, 0.05), 0.95)
    assert np.isclose(float(w[affs > 0].max()), 1 / (2 * frac)) and float(np.abs(w[mask == 0]).max()) == 0.0
    # grow boundary: two touching labels lose one voxel each along their border, not along unknown (0) regions
    lab = np.zeros((1, 6, 8), np.int64)
    lab[0, 1:5, 1:4], lab[0, 1:5, 4:7] = 5, 9
    g = TR.grow_boundary(lab, (lab > 0).astype(np.uint8), 1, only_xy=True)
    assert (g[0, 1:5, 3:5] == 0).all() and (g[0, 1:5, 1:3] == 5).all() and (g[0, 1:5, 5:7] == 9).all()

    from tests.test_lib_cpu import AFFS_NET_CONFIG
    sd = T.default_init(AFFS_NET_CONFIG)
    assert sd["unet.l_conv.3.conv_pass.2.weight"].shape == (1500, 1500, 3, 3, 3)
    assert sd["unet.r_conv.0.2.conv_pass.0.weight"].shape == (300, 1800, 3, 3, 3) and sd["affs_head.residual.0.weight"].shape == (6, 12, 1, 1, 1)
    assert abs(float(np.abs(sd["unet.l_conv.0.conv_pass.0.weight"]).max()) - 1 / np.sqrt(27)) < 0.01
    from bootstrapper_amd.unet import Model
    import ctypes as C
    from bootstrapper_amd._lib import lib, check
    m = Model(AFFS_NET_CONFIG)
    for k, v in sd.items():      # every key and shape is one the engine expects (no device work: load only)
        a = np.ascontiguousarray(v, dtype=np.float32)
        check(lib.bsmi_unet_load_weight(m._h, k.encode(), a.ctypes.data_as(C.c_void_p), (C.c_int64 * a.ndim)(*a.shape), a.ndim))


def test_block_task_retries_and_failure_accounting():
    """bootstrapper_amd.blockwise against the contract of the reference's blockwise.py:12-22 and daisy's max_retries=5
    (predict.py:38): a block that raises twice and then succeeds completes; one that always raises is counted failed after
    six attempts; downstream blocks that read a failed block are orphaned; any of these ends in the reference's RuntimeError."""
    import pytest
    from bootstrapper_amd.blockwise import run_blocks, check_task_states, FatalBlockError, TaskState
    calls = {}

    def flaky(blk):
        calls[blk] = calls.get(blk, 0) + 1
        if blk == 3 and calls[blk] <= 2:
            raise ValueError("transient")
        if blk == 5:
            raise IOError("always")
    st = run_blocks("PredictBlockwiseTask", list(range(8)), flaky)
    assert calls[3] == 3 and calls[5] == 6 and all(calls[b] == 1 for b in (0, 1, 2, 4, 6, 7))
    assert st.as_tuple() == (8, 7, 1, 0, [5])
    with pytest.raises(RuntimeError, match=r"task PredictBlockwiseTask: 1 failed, 0 orphaned of 8 blocks"):
        check_task_states({"PredictBlockwiseTask": st})
    # downstream task: blocks 4, 5, 6 read the failed block 5
    ran = []
    st2 = run_blocks("agglom", list(range(8)), ran.append, upstream_failed=st.failed_blocks, depends_on=lambda b: (b - 1, b, b + 1))
    assert ran == [0, 1, 2, 3, 7] and st2.as_tuple() == (8, 5, 0, 3, [])
    merged = TaskState("agglom").merge(st2).merge(TaskState.from_tuple("agglom", (4, 4, 0, 0, [])))
    with pytest.raises(RuntimeError, match=r"task agglom: 0 failed, 3 orphaned of 12 blocks"):
        check_task_states({"agglom": merged})
    check_task_states({"ok": run_blocks("ok", [1, 2], lambda b: None)})

    def fatal(blk):
        raise FatalBlockError("device lost")
    with pytest.raises(FatalBlockError):
        run_blocks("t", [0, 1], fatal)


def test_histogram_quantile_merge_loop_on_the_host():
    """The host merge loop behind `merge_function = hist_quant_<Q>[_initmax]` (csrc/agglo_host.cpp, through the C ABI; no GPU):
    on region graphs of small random volumes it produces the clusters of the oracle (oracle/seg_ref.c, itself checked against
    the literal model in tests/test_oracle_seg.py)."""
    import ctypes as C
    from bootstrapper_amd._lib import lib, check
    from oracle import seg_ref as S
    from tests.agglo_model import region_graph
    rng = np.random.default_rng(31)
    for case in range(8):
        shape = (int(rng.integers(1, 4)), int(rng.integers(5, 10)), int(rng.integers(5, 10)))
        frags = rng.integers(0 if case % 2 else 1, int(rng.integers(5, 16)), size=shape).astype(np.uint64)
        affs = rng.choice(np.array([0, 40, 80, 128, 200, 255]) if case % 3 else np.arange(256), size=(3,) + shape).astype(np.uint8)
        ids = np.array(sorted(set(frags.ravel().tolist()) - {0}), dtype=np.uint64)
        rank = {int(v): i for i, v in enumerate(ids)}
        g = region_graph(affs, frags, values=True)
        keys = sorted(g)
        eu = np.array([rank[k[0]] for k in keys], dtype=np.uint32)
        ev = np.array([rank[k[1]] for k in keys], dtype=np.uint32)
        thr = np.array([0.3, 0.55, 0.8], dtype=np.float32)
        for q, initmax in ((50, False), (10, True), (90, False), (75, True)):
            hist = np.zeros((len(keys), 256), dtype=np.uint32)
            for e, k in enumerate(keys):
                np.add.at(hist[e], g[k], 1)
            roots = np.zeros((3, len(ids)), dtype=np.uint32)
            check(lib.bsmi_agglomerate_hist_graph(len(ids), len(keys), eu.ctypes.data, ev.ctypes.data, hist.ctypes.data, q, int(initmax),
                                                  thr.ctypes.data_as(C.POINTER(C.c_float)), 3, roots.ctypes.data))
            want = S.agglomerate_hist_u8(affs, frags, thr, q, initmax)
            for t in range(3):
                lut = np.zeros(int(ids.max()) + 1, dtype=np.uint64)
                lut[ids.astype(np.int64)] = ids[roots[t]]
                assert np.array_equal(lut[frags.astype(np.int64)], want[t]), (case, q, initmax, t)


def test_tensorboard_event_file_round_trip(tmp_path):
    """tb_events: TFRecord framing and the Event / Summary fields TensorBoard reads scalars from (reference training.py:130
    logs `train_loss` through Lightning's TensorBoardLogger).  Known answers: CRC-32C of "123456789" is 0xE3069283 (RFC 3720
    B.4); the record of an empty payload."""
    import struct
    from bootstrapper_amd import tb_events as T
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(b"") == 0
    assert T.record_bytes(b"")[:8] == struct.pack("<Q", 0) and len(T.record_bytes(b"abc")) == 8 + 4 + 3 + 4
    with T.ScalarWriter(str(tmp_path / "log")) as w:
        for step, v in [(10, 0.5), (20, 0.25), (300000, 1e-3)]:
            w.add_scalar("train_loss", v, step)
        path = w.path
    assert os.path.basename(os.path.dirname(path)) == "version_0" and os.path.basename(path).startswith("events.out.tfevents.")
    version, scalars = T.read_scalars(path)
    assert version == "brain.Event:2"
    assert [(s, t) for s, t, _, _ in scalars] == [(10, "train_loss"), (20, "train_loss"), (300000, "train_loss")]
    assert np.allclose([v for _, _, v, _ in scalars], [0.5, 0.25, 1e-3], rtol=1e-7)
    with T.ScalarWriter(str(tmp_path / "log")) as w2:   # a second run gets its own version directory
        assert os.path.basename(w2.dir) == "version_1"
    raw = bytearray(open(path, "rb").read())
    raw[-6] ^= 1                                         # a flipped payload bit is caught by the checksum
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        T.read_scalars(str(bad))


def _numpy_region_graph(affs, frags):
    """edges (ascending id pairs), affinity sums and voxel-pair counts of the faces between different fragments: the pair
    (p, p - 1 along axis d) counts with affs[d][p] (seg.hip agg_edges_kernel; waterz's region graph)"""
    keys, vals = [], []
    for d in range(3):
        hi = [slice(None)] * 3
        lo = [slice(None)] * 3
        hi[d], lo[d] = slice(1, None), slice(None, -1)
        a, b, w = frags[tuple(hi)], frags[tuple(lo)], affs[d][tuple(hi)]
        m = (a > 0) & (b > 0) & (a != b)
        u, v = np.minimum(a[m], b[m]), np.maximum(a[m], b[m])
        keys.append(np.stack([u, v], axis=1))
        vals.append(w[m].astype(np.uint64))
    keys, vals = np.concatenate(keys), np.concatenate(vals)
    uniq, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    sums = np.bincount(inv, weights=vals.astype(np.float64), minlength=len(uniq)).astype(np.uint64)
    cnts = np.bincount(inv, minlength=len(uniq)).astype(np.uint32)
    return uniq.astype(np.uint64), sums, cnts


@pytest.mark.parametrize("shape,sigma,msd,bins,thr,levels", [
    ((8, 64, 64), (1, 3, 3), 5, 256, 1.0, 0), ((6, 80, 80), (1, 3, 3), 5, 256, 0.45, 0),
    ((10, 48, 56), (1, 2, 2), 3, 16, 1.0, 0), ((3, 50, 70), (0, 1, 1), 3, 1, 1.0, 0),
    # tie-heavy graphs: affinities quantised to a few levels (u8 predictions saturate at 0 / 255), so that parallel edges meet
    # with EQUAL stored scores and which of the two lives on decides the merge order (the host loop walked the
    # shorter list and compared one way; 23 of 60 such graphs differed from the oracle)
    ((8, 64, 64), (1, 3, 3), 5, 256, 1.0, 2), ((6, 80, 80), (1, 2, 2), 3, 256, 1.0, 3), ((6, 80, 80), (1, 3, 3), 5, 256, 0.6, 4),
    ((10, 48, 56), (1, 2, 2), 3, 16, 1.0, 3), ((4, 50, 70), (0, 1, 1), 3, 1, 1.0, 2), ((8, 64, 64), (1, 2, 2), 3, 256, 1.0, 8)])
def test_host_merge_loop_of_the_edge_scoring_equals_the_oracle(shape, sigma, msd, bins, thr, levels):
    """bsmi_rag_merge_scores_host (csrc/agglo_host.cpp: the block pipeline's edge scoring on host threads) against the C
    restatement of waterz_agglom.py:106-170, on region graphs built here in numpy: the same edges, bit for bit the same
    scores (NaN = never merged), with coarse and single-bin queues and a threshold that leaves edges unmerged."""
    from scipy.ndimage import gaussian_filter
    from bootstrapper_amd.post.engine import rag_merge_scores_host
    from oracle import seg_ref as S
    rng = np.random.default_rng(shape[1] * 7 + bins + 1000 * levels)
    graphs, refs = [], []
    for g in range(3):
        a = gaussian_filter(rng.random((3,) + shape), sigma=(0,) + sigma)
        a = (a - a.min()) / (a.max() - a.min())
        if levels:
            a = np.clip((a - 0.5) * 2.5 + 0.5, 0, 1)                     # saturate like a trained net's output
            a = np.round(a * (levels - 1)) / (levels - 1)
        affs = (a * 255).astype(np.uint8)
        frags, _ = S.ws_fragments_u8(affs, True, msd)
        frags = np.where(frags > 0, frags + np.uint64((1 << 40) * g), np.uint64(0))     # ids beyond 32 bits
        e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, thr, bins)
        e, sums, cnts = _numpy_region_graph(affs, frags)
        assert np.array_equal(e, e_ref)
        graphs.append((e, sums, cnts))
        refs.append(s_ref)
    cap = max(len(g[0]) for g in graphs) + 3
    E = np.zeros((3, cap, 2), np.uint64); Sm = np.zeros((3, cap), np.uint64); Cn = np.ones((3, cap), np.uint32)
    for g, (e, s, c) in enumerate(graphs):
        E[g, :len(e)], Sm[g, :len(e)], Cn[g, :len(e)] = e, s, c
    ne = np.array([len(g[0]) for g in graphs])
    # the device hands a graph over in the order of its hash table: two of the three arrive shuffled and are sorted in place
    for g in (0, 2):
        perm = rng.permutation(ne[g])
        E[g, :ne[g]], Sm[g, :ne[g]], Cn[g, :ne[g]] = E[g, perm], Sm[g, perm], Cn[g, perm]
    sc = rag_merge_scores_host(ne, E, Sm, Cn, thr, bins, threads=2)
    for g, s_ref in enumerate(refs):
        assert np.array_equal(E[g, :ne[g]], graphs[g][0]) and np.array_equal(Sm[g, :ne[g]], graphs[g][1]) and np.array_equal(Cn[g, :ne[g]], graphs[g][2])
        assert np.array_equal(sc[g, :ne[g]].view(np.uint32), s_ref.view(np.uint32))
        assert np.isnan(sc[g, ne[g]:]).all()
    if thr < 1.0:
        assert np.isnan(refs[0]).any() and (~np.isnan(refs[0])).any()


def test_queue_bin_rule_and_cc_inclusivity_are_switchable():
    """The two documented choices where waterz / funlib.segment are unpinned (segment config keys
    `queue_bins_formula`, `cc_inclusive`): the alternative bin rule in the host merge loop equals the oracle run with the same
    rule (and is a different merge order on a coarse queue), the strict connected-components rule is the inclusive one at the
    float32 just below the threshold."""
    from scipy.ndimage import gaussian_filter
    from bootstrapper_amd.post.engine import rag_merge_scores_host
    from bootstrapper_amd.post.watershed import connected_components_multi
    from oracle import seg_ref as S
    rng = np.random.default_rng(77)
    differ = 0
    for bins in (16, 256):
        a = gaussian_filter(rng.random((3, 6, 64, 64)), sigma=(0, 1, 2, 2))
        affs = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
        frags, _ = S.ws_fragments_u8(affs, True, 3)
        e, sums, cnts = _numpy_region_graph(affs, frags)
        out = {}
        for rule in ("n_minus_1", "n"):
            e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, 1.0, bins, bins_formula=rule)
            E, Sm, Cn = e[None].copy(), sums[None].copy(), cnts[None].copy()
            sc = rag_merge_scores_host(np.array([len(e)]), E, Sm, Cn, 1.0, bins, bins_formula=rule)
            assert np.array_equal(e_ref, e) and np.array_equal(sc[0].view(np.uint32), s_ref.view(np.uint32)), (bins, rule)
            out[rule] = s_ref
        differ += not np.array_equal(out["n"].view(np.uint32), out["n_minus_1"].view(np.uint32))
    assert differ >= 1
    with pytest.raises(ValueError):
        rag_merge_scores_host(np.array([0]), np.zeros((1, 1, 2), np.uint64), np.zeros((1, 1), np.uint64), np.ones((1, 1), np.uint32), bins_formula="floor")
    # connected components: score == threshold joins under the inclusive rule only
    nodes = np.arange(1, 7, dtype=np.uint64)
    edges = np.array([[1, 2], [2, 3], [4, 5], [5, 6]], dtype=np.uint64)
    scores = np.array([0.5, 0.25, np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(0.5), np.float32(0))], dtype=np.float32)
    incl = connected_components_multi(nodes, edges, scores, [0.5])[0]
    strict = connected_components_multi(nodes, edges, scores, [float(np.nextafter(np.float32(0.5), np.float32(-np.inf)))])[0]
    assert list(incl) == [1, 1, 1, 4, 5, 5] and list(strict) == [1, 2, 2, 4, 5, 5]
    assert np.array_equal(incl, S.connected_components(nodes, edges, scores, 0.5))
