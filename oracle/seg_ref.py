"""ORACLE loader (test infrastructure): ctypes binding of oracle/seg_ref.c.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsegref.so")


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
