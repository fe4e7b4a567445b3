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
