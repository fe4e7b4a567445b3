"""Pin oracle/seg_ref.c (watershed fragments) bit-for-bit against golden vectors produced
by running the reference post/ws.py (tools/gen_goldens_ws.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import seg_ref as S


def _cases(golden_dir):
    d = np.load(os.path.join(golden_dir, "ws_cases.npz"))
    names = sorted({k.split("/")[0] for k in d.files})
    return d, names


def test_fragments_bit_exact_vs_reference(golden_dir):
    d, names = _cases(golden_dir)
    assert len(names) >= 14
    for name in names:
        xy, msd, max_id = (int(v) for v in d[name + "/meta"])
        frags, mx, seeds = S.ws_fragments_u8(d[name + "/affs"], bool(xy), msd, return_seeds=True)
        assert mx == max_id, name
        assert np.array_equal(seeds, d[name + "/seeds"].astype(np.uint64)), name
        assert np.array_equal(frags, d[name + "/frags"].astype(np.uint64)), name


def test_agglomeration_invariants():
    """waterz parity is unpinned (no runnable reference); check the structural invariants of
    the specified algorithm: segmentations are coarsenings of the fragments, nested across
    ascending thresholds, threshold 0 leaves fragments untouched, and a huge threshold merges
    every connected group of fragments."""
    rng = np.random.default_rng(3)
    from scipy.ndimage import gaussian_filter
    a = gaussian_filter(rng.random((3, 10, 48, 48)), sigma=(0, 1, 3, 3))
    a = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    frags, _ = S.ws_fragments_u8(a, True, 10)
    segs = S.agglomerate_mean_u8(a, frags, [0.0, 0.2, 0.5, 1.5])
    assert np.array_equal(segs[0], frags)
    prev = frags
    for s in segs:
        assert np.array_equal(s == 0, frags == 0)
        # coarsening: each previous label maps to exactly one new label
        pairs = np.unique(np.stack([prev.ravel(), s.ravel()]), axis=1)
        assert len(np.unique(pairs[0])) == pairs.shape[1]
        # root id is the smallest fragment id of the segment
        assert np.all(s <= frags)
        prev = s
    n = [len(np.unique(s)) for s in segs]
    assert n[0] >= n[1] >= n[2] >= n[3]
    assert n[3] < n[0]


def test_blockwise_primitives_vs_library_goldens(golden_dir):
    """seg_fragment_means / filter / remove-small / label26 against scipy + skimage outputs
    (tools/gen_goldens_blockwise.py)."""
    d = np.load(os.path.join(golden_dir, "blockwise_cases.npz"))
    for name in ("a", "b", "c"):
        frags, affs, ids = d[name + "/frags"], d[name + "/affs"], d[name + "/ids"]
        means = S.fragment_means_u8(affs, frags, ids)
        assert np.array_equal(means, d[name + "/means"]), name  # bit-exact float64
        for thr in (0.1, 0.35, 0.5):
            got = S.filter_fragments_u8(affs, frags, thr, 0)
            ref = frags.copy()
            ref[np.isin(ref, d[name + f"/filtered_{thr}"])] = 0
            assert np.array_equal(got, ref), (name, thr)
        for ms in (1, 8, 40, 200):
            got = S.filter_fragments_u8(affs, frags, 0.0, ms)
            assert np.array_equal(got, d[name + f"/debris_{ms}"]), (name, ms)
        lab, num = S.label26(frags[1:-1, 2:-3, 3:-2])
        assert num == int(d[name + "/crop_num"][0])
        assert np.array_equal(lab, d[name + "/crop_label"])
    for name in ("checker", "diag"):
        lab, num = S.label26(d[name + "/frags"])
        assert num == int(d[name + "/num"][0]) and np.array_equal(lab, d[name + "/label"])


def test_float_boundary_mask_equals_integer_threshold():
    """The reference drivers hand float affinities (u8/255 as float32 in post/watershed.py:245-248, float64 in
    post/blockwise/watershed_frags.py:198-205) to post/ws.py:75 `0.5*(a_y+a_x) > 0.5`; the oracle and the device
    threshold the u8 values `a_y + a_x >= 256`.  Check all 65536 pairs agree in both float widths."""
    a = np.arange(256, dtype=np.uint8)
    A, B = np.meshgrid(a, a, indexing="ij")
    integer = (A.astype(np.int64) + B.astype(np.int64)) >= 256
    for dt in (np.float32, np.float64):
        fa, fb = A.astype(dt) / dt(255.0), B.astype(dt) / dt(255.0)
        assert np.array_equal(0.5 * (fa + fb) > 0.5 * 1.0, integer)


def test_rag_merge_scores_match_merge_tree_replay_and_cc():
    """The oracle's per-edge merge score equals replaying its own merge history through the MergeTree mirror
    (pinned to the reference's post/merge_tree.py by tests/golden/host_cases.json), and connected components
    at a threshold reproduce the clusters of a full agglomeration when scores happen to be monotone."""
    from scipy.ndimage import gaussian_filter
    from bootstrapper_amd.post.merge_tree import MergeTree
    rng = np.random.default_rng(5)
    a = gaussian_filter(rng.random((3, 6, 72, 72)), sigma=(0, 1, 3, 3))
    affs = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    frags, _ = S.ws_fragments_u8(affs, True, 5)
    edges, scores, merges, mscores = S.rag_merge_scores_u8(affs, frags, 1.0, 256)
    nodes = np.unique(frags)
    nodes = nodes[nodes > 0]
    assert len(merges) <= len(nodes) - 1 and len(edges) > len(nodes)
    assert np.all(edges[:, 0] < edges[:, 1])
    keys = edges[:, 0].astype(object) * (1 << 64) + edges[:, 1].astype(object)
    assert all(keys[i] < keys[i + 1] for i in range(len(keys) - 1))
    mt = MergeTree(nodes)
    for (x, y), sc in zip(merges, mscores):
        mt.merge(x, y, x, sc)
    ref = mt.find_merges(edges[:, 0], edges[:, 1])
    assert np.array_equal(np.isnan(ref), np.isnan(scores))
    assert np.array_equal(ref[~np.isnan(ref)].astype(np.float32), scores[~np.isnan(scores)])
    # connected components: partition check against a tiny python union-find
    thr = 0.4
    comp = S.connected_components(nodes, edges, scores, thr)
    parent = {int(n): int(n) for n in nodes}
    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x
    for (u, v), sc in zip(edges, scores):
        if sc <= thr:
            ru, rv = find(int(u)), find(int(v))
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
    assert [find(int(n)) for n in nodes] == comp.tolist()


def test_cc_oracle_vs_reference_goldens(golden_dir):
    """post/cc.py (run by tools/gen_goldens_cc.py) against the C restatement, including the float threshold -> u8 cut."""
    d = np.load(os.path.join(golden_dir, "cc_cases.npz"))
    names = sorted({k.split("/")[0] for k in d.files})
    assert len(names) >= 6
    for name in names:
        affs, thr = d[name + "/affs"], float(d[name + "/thr"])
        assert np.array_equal(affs > S.cc_cut(thr), d[name + "/hard"]), name
        seg, n = S.cc_affs_u8(affs, thr)
        assert np.array_equal(seg, d[name + "/seg"]) and n == int(d[name + "/seg"].max()), name
    for thr in (0.0, 0.2, 0.35, 0.5, 1 / 3, 0.999, 1.0):
        u = np.arange(256, dtype=np.uint8)
        assert np.array_equal((u.astype(np.float32) / 255.0) > thr, u > S.cc_cut(thr)), thr


def _three_node_case():
    """Fragments 1 | 2 over 3 3 on a 1 x 2 x 2 grid: edges {1,2} (x-affinity at voxel (0,0,1)), {1,3} and {2,3}
    (y-affinities at (0,1,0) and (0,1,1))."""
    frags = np.array([[[1, 2], [3, 3]]], dtype=np.uint64)
    affs = np.zeros((3, 1, 2, 2), dtype=np.uint8)
    affs[2, 0, 0, 1] = 230   # {1,2}: score 0.098
    affs[1, 0, 1, 0] = 26    # {1,3}: score 0.898
    affs[1, 0, 1, 1] = 204   # {2,3}: score 0.200
    return affs, frags


def test_shared_neighbour_keeps_the_cheaper_edge():
    """waterz mergeRegions merges the dearer of two parallel edges into the cheaper one, which keeps its queue
    position.  After 1 <- 2 the edges {1,3} (0.898) and {2,3} (0.2) are one edge of true score 0.549: it must be
    met at 0.2, rescored and merged below the threshold 0.6.  (Keeping the a-side edge at 0.898 would end the loop
    with fragment 3 unmerged: the rule this restatement had before.)"""
    affs, frags = _three_node_case()
    segs = S.agglomerate_mean_u8(affs, frags, [0.5, 0.6])
    assert np.array_equal(segs[0], np.array([[[1, 1], [3, 3]]], dtype=np.uint64))
    assert np.array_equal(segs[1], np.ones((1, 2, 2), dtype=np.uint64))
    edges, scores, merges, mscores = S.rag_merge_scores_u8(affs, frags, 0.6, 256)
    assert merges.tolist() == [[1, 2], [1, 3]]
    np.testing.assert_allclose(mscores, [1 - 230 / 255, 1 - 230 / 510], rtol=1e-6)


def test_merge_order_follows_the_cheaper_edge():
    """Four fragments in a row of columns 1 2 3 4 over a second row 1 1 3 4 ... : the merged edge {1,3} (true score
    0.45) must be handled before {3,4} (0.5), after which {3,4} and {1,4} are parallel and too dear to merge; with the
    merged edge waiting at the dearer score 0.8, 3 <- 4 would happen first and everything would end up in one segment."""
    frags = np.array([[[1, 2, 2], [3, 3, 4], [1, 1, 4]]], dtype=np.uint64)
    affs = np.zeros((3, 1, 3, 3), dtype=np.uint8)
    affs[2, 0, 0, 1] = 240            # {1,2} x: 0.059
    affs[1, 0, 1, 0] = 51             # {1,3} y at (1,0): 0.8
    affs[1, 0, 1, 1] = 230            # {2,3} y at (1,1): 0.098
    affs[1, 0, 1, 2] = 0              # {2,4} y at (1,2): 1.0
    affs[2, 0, 1, 2] = 128            # {3,4} x at (1,2): 0.498
    affs[1, 0, 2, 0] = 51             # {1,3} y at (2,0)
    affs[1, 0, 2, 1] = 230            # {1,3} y at (2,1)
    affs[2, 0, 2, 2] = 0              # {1,4} x at (2,2): 1.0
    from tests.agglo_model import agglomerate
    want = agglomerate(affs, frags, [0.55])[0]
    got = S.agglomerate_mean_u8(affs, frags, [0.55])[0]
    assert np.array_equal(got, want)
    assert len(np.unique(got)) == 2 and got[0, 1, 2] == 4      # 4 stays on its own


def test_oracle_equals_literal_python_model():
    """The C restatement against the dictionary-based model of the same specification (tests/agglo_model.py) on small
    random volumes with many ties: exact queue (segmentations at three thresholds) and bin queue (merge history)."""
    from tests.agglo_model import agglomerate, merge_history
    rng = np.random.default_rng(17)
    for case in range(12):
        shape = (int(rng.integers(1, 4)), int(rng.integers(4, 9)), int(rng.integers(4, 9)))
        nfr = int(rng.integers(4, 14))
        frags = rng.integers(0 if case % 3 == 0 else 1, nfr, size=shape).astype(np.uint64)
        levels = np.array([0, 40, 80, 128, 200, 255]) if case % 2 else np.arange(256)
        affs = rng.choice(levels, size=(3,) + shape).astype(np.uint8)
        thr = [0.3, 0.55, 0.8]
        got = S.agglomerate_mean_u8(affs, frags, thr)
        want = agglomerate(affs, frags, thr)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), case
        for nbins in (256, 8):
            hist = merge_history(affs, frags, 0.7, nbins)
            _, _, merges, mscores = S.rag_merge_scores_u8(affs, frags, 0.7, nbins)
            assert [(a, b) for a, b, _ in hist] == [tuple(m) for m in merges.tolist()], (case, nbins)
            np.testing.assert_array_equal(np.array([s for _, _, s in hist], dtype=np.float32), mscores)


def test_histogram_quantile_oracle_equals_literal_model():
    """OneMinus<HistogramQuantileAffinity<., q, ., 256, init_with_max>> (reference post/watershed.py:230-243: q in 10, 25, 50,
    75, 90, each with and without init_with_max): the C restatement against the literal model, and the scoring rule itself on
    hand-made histograms (parity unpinned: waterz is absent; the rule is stated in oracle/seg_ref.c)."""
    from tests.agglo_model import agglomerate, _quantile_score
    # pivot = q * n / 100 + 1, 1-based: 4 values, q = 50 -> the 3rd smallest; q = 90 -> the 4th; q = 10 -> the 1st
    assert _quantile_score([10, 20, 30, 40], 50) == np.float32(1) - np.float32(30.5) / np.float32(256)
    assert _quantile_score([10, 20, 30, 40], 90) == np.float32(1) - np.float32(40.5) / np.float32(256)
    assert _quantile_score([10, 20, 30, 40], 10) == np.float32(1) - np.float32(10.5) / np.float32(256)
    rng = np.random.default_rng(23)
    differs = 0
    for case in range(10):
        shape = (int(rng.integers(1, 4)), int(rng.integers(4, 9)), int(rng.integers(4, 9)))
        frags = rng.integers(0 if case % 3 == 0 else 1, int(rng.integers(4, 14)), size=shape).astype(np.uint64)
        levels = np.array([0, 40, 80, 128, 200, 255]) if case % 2 else np.arange(256)
        affs = rng.choice(levels, size=(3,) + shape).astype(np.uint8)
        thr = [0.3, 0.55, 0.8]
        mean = S.agglomerate_mean_u8(affs, frags, thr)
        for q in (10, 50, 90):
            for initmax in (False, True):
                got = S.agglomerate_hist_u8(affs, frags, thr, q, initmax)
                want = agglomerate(affs, frags, thr, quantile=q, init_with_max=initmax)
                for g, w in zip(got, want):
                    assert np.array_equal(g, w), (case, q, initmax)
                differs += any(not np.array_equal(g, m) for g, m in zip(got, mean))
    assert differs > 20      # the scorers are not the mean scorer in disguise
