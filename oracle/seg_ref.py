"""ORACLE loader (test infrastructure): ctypes binding of oracle/seg_ref.c.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSMI_ORACLE_SO: another build of the same source (tests/test_sanitizers.py points it at `make -C oracle asan`'s library)
_SO = os.environ.get("BSMI_ORACLE_SO") or os.path.join(_HERE, "_build", "libsegref.so")


def _load():
    src = os.path.join(_HERE, "seg_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    lib = C.CDLL(_SO)
    vp = C.c_void_p
    lib.seg_ws_fragments_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.seg_ws_fragments_u8.restype = C.c_int
    lib.seg_agglomerate_mean_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.seg_agglomerate_mean_u8.restype = C.c_int
    lib.seg_agglomerate_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int]
    lib.seg_agglomerate_u8.restype = C.c_int
    lib.seg_fragment_means_u8.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, vp]
    lib.seg_fragment_means_u8.restype = None
    lib.seg_filter_fragments_u8.argtypes = [vp, vp, C.c_int64, C.c_double, C.c_int64]
    lib.seg_filter_fragments_u8.restype = None
    lib.seg_label26.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.seg_label26.restype = C.c_int64
    lib.seg_rag_merge_scores_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, C.c_int64, vp, vp, vp]
    lib.seg_rag_merge_scores_u8.restype = C.c_int64
    lib.seg_connected_components.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, C.c_float, vp]
    lib.seg_connected_components.restype = None
    lib.seg_cc_cut.argtypes = [C.c_double]
    lib.seg_cc_cut.restype = C.c_int
    lib.seg_cc_affs_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.seg_cc_affs_u8.restype = C.c_int64
    lib.seg_set_bin_rule.argtypes = [C.c_int]
    lib.seg_set_bin_rule.restype = None
    lib.seg_count_labels.argtypes = [vp, C.c_int64]
    lib.seg_count_labels.restype = C.c_int64
    return lib


_lib = _load()


def ws_fragments_u8(affs_u8, fragments_in_xy=True, min_seed_distance=10, return_seeds=False):
    """post/ws.py:38-112 on uint8 affinities [3][D][H][W] -> (fragments u64, max_id[, seeds])."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    _, D, H, W = a.shape
    frags = np.zeros((D, H, W), dtype=np.uint64)
    seeds = np.zeros((D, H, W), dtype=np.uint64) if return_seeds else None
    mx = C.c_uint64(0)
    rc = _lib.seg_ws_fragments_u8(a.ctypes.data, D, H, W, int(bool(fragments_in_xy)), int(min_seed_distance),
                                  frags.ctypes.data, C.addressof(mx), seeds.ctypes.data if return_seeds else None)
    assert rc == 0
    return (frags, mx.value, seeds) if return_seeds else (frags, mx.value)


def agglomerate_mean_u8(affs_u8, frags, thresholds):
    """waterz.agglomerate(affs, thresholds, fragments, OneMinus<MeanAffinity>) restatement
    -> list of uint64 segmentations, one per threshold."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    f = np.ascontiguousarray(frags, dtype=np.uint64)
    _, D, H, W = a.shape
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    segs = np.zeros((len(thr), D, H, W), dtype=np.uint64)
    rc = _lib.seg_agglomerate_mean_u8(a.ctypes.data, f.ctypes.data, D, H, W, thr.ctypes.data, len(thr),
                                      segs.ctypes.data)
    assert rc == 0
    return [segs[i] for i in range(len(thr))]


def agglomerate_hist_u8(affs_u8, frags, thresholds, quantile, init_with_max=False):
    """waterz.agglomerate(..., OneMinus<HistogramQuantileAffinity<RegionGraphType, quantile, ScoreValue, 256, init_with_max>>)
    restatement (reference post/watershed.py:230-243) -> list of uint64 segmentations, one per threshold."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    f = np.ascontiguousarray(frags, dtype=np.uint64)
    _, D, H, W = a.shape
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    segs = np.zeros((len(thr), D, H, W), dtype=np.uint64)
    rc = _lib.seg_agglomerate_u8(a.ctypes.data, f.ctypes.data, D, H, W, thr.ctypes.data, len(thr), segs.ctypes.data,
                                 int(quantile), int(bool(init_with_max)))
    assert rc == 0
    return [segs[i] for i in range(len(thr))]


def fragment_means_u8(affs_u8, frags, ids):
    """watershed_frags.py:148-152: per-fragment mean of the 3-channel average affinity (float64)."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    f = np.ascontiguousarray(frags, dtype=np.uint64)
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    means = np.zeros(len(ids), dtype=np.float64)
    _lib.seg_fragment_means_u8(a.ctypes.data, f.ctypes.data, f.size, ids.ctypes.data, len(ids), means.ctypes.data)
    return means


def filter_fragments_u8(affs_u8, frags, filter_value, min_size):
    """filter_avg_fragments + remove_small_objects (watershed_frags.py:181-192); returns a copy."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    f = np.array(frags, dtype=np.uint64, order="C", copy=True)
    _lib.seg_filter_fragments_u8(a.ctypes.data, f.ctypes.data, f.size, float(filter_value), int(min_size))
    return f


def label26(x):
    """skimage.measure.label(x, return_num=True) -> (labels uint32, num)."""
    x = np.ascontiguousarray(x, dtype=np.uint64)
    lab = np.zeros(x.shape, dtype=np.uint32)
    n = _lib.seg_label26(x.ctypes.data, x.shape[0], x.shape[1], x.shape[2], lab.ctypes.data)
    return lab, int(n)


def rag_merge_scores_u8(affs_u8, frags, threshold=1.0, discretize_queue=256, bins_formula="n_minus_1"):
    """waterz_agglom.py:106-170 for one block -> (edges uint64 [ne][2] ascending, scores float32 [ne]
    (NaN = never merged), merges uint64 [nm][2] (survivor, absorbed), merge scores float32 [nm]).
    bins_formula: "n_minus_1" (the specification) or "n" (the documented alternative; process-wide while the call runs)."""
    _lib.seg_set_bin_rule({"n_minus_1": 0, "n": 1}[bins_formula])
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    f = np.ascontiguousarray(frags, dtype=np.uint64)
    _, D, H, W = a.shape
    cap = 3 * f.size + 1
    edges = np.zeros((cap, 2), dtype=np.uint64)
    scores = np.zeros(cap, dtype=np.float32)
    merges = np.zeros((f.size + 1, 2), dtype=np.uint64)
    mscores = np.zeros(f.size + 1, dtype=np.float32)
    nm = C.c_int64(0)
    ne = _lib.seg_rag_merge_scores_u8(a.ctypes.data, f.ctypes.data, D, H, W, float(threshold), int(discretize_queue),
                                      edges.ctypes.data, scores.ctypes.data, cap, merges.ctypes.data,
                                      mscores.ctypes.data, C.addressof(nm))
    _lib.seg_set_bin_rule(0)
    return edges[:ne].copy(), scores[:ne].copy(), merges[:nm.value].copy(), mscores[:nm.value].copy()


def connected_components(nodes, edges, scores, threshold):
    """funlib.segment connected_components restatement (score <= threshold joins; component = smallest id)."""
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    assert np.all(nodes[1:] > nodes[:-1]), "nodes must be ascending"
    edges = np.ascontiguousarray(edges, dtype=np.uint64).reshape(-1, 2)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    out = np.zeros(len(nodes), dtype=np.uint64)
    _lib.seg_connected_components(nodes.ctypes.data, len(nodes), edges.ctypes.data, scores.ctypes.data, len(scores),
                                  float(threshold), out.ctypes.data)
    return out


def cc_cut(threshold):
    """u8 cut equivalent to `(u8 / 255 as float32) > threshold` (post/connected_components.py:49-52,77)."""
    return int(_lib.seg_cc_cut(float(threshold)))


def cc_affs_u8(affs_u8, threshold):
    """post/cc.py:7-74 on uint8 affinities [3][D][H][W] -> (labels uint32, count)."""
    a = np.ascontiguousarray(affs_u8[:3], dtype=np.uint8)
    _, D, H, W = a.shape
    seg = np.zeros((D, H, W), dtype=np.uint32)
    n = _lib.seg_cc_affs_u8(a.ctypes.data, D, H, W, cc_cut(threshold), seg.ctypes.data)
    return seg, int(n)
