"""Runs under LD_PRELOAD=libasan.so (tests/test_sanitizers.py starts it): the host C++ of libbsmi (`make -C
bootstrapper_amd/csrc asan`: chunk codecs, merge loops, the 3-D flood) and the C oracle (`make -C oracle asan`) with
AddressSanitizer + UBSan.  A finding aborts the process; wrong results fail the asserts.  No GPU, no libbsmi.so."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Codec(C.Structure):   # bsmi_codec, include/bsmi_io.h
    _fields_ = [("id", C.c_int32), ("level", C.c_int32), ("cname", C.c_int32), ("shuffle", C.c_int32),
                ("typesize", C.c_int32), ("blocksize", C.c_int32)]


RAW, ZLIB, GZIP, ZSTD, LZ4, BLOSC = range(6)


def main():
    host = C.CDLL(os.path.join(ROOT, "bootstrapper_amd", "libbsmi_host_asan.so"))
    host.bsmi_codec_bound.restype = C.c_size_t
    host.bsmi_codec_bound.argtypes = [C.c_void_p, C.c_size_t]
    host.bsmi_codec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    host.bsmi_codec_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    host.bsmi_rag_merge_scores_host.argtypes = [C.c_int, C.c_void_p] + [C.c_void_p] * 3 + [C.c_float, C.c_int, C.c_void_p, C.c_int]
    host.bsmi_agglomerate_hist_graph.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    host.bsmi_san_host_flood3.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3
    rng = np.random.default_rng(11)

    def decode(codec, frame, cap):
        """exact-size heap buffers on both sides, so that one byte out of bounds is a report"""
        src = np.frombuffer(bytes(frame), dtype=np.uint8).copy()
        dst = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t()
        rc = host.bsmi_codec_decode(C.byref(codec), src.ctypes.data, src.size, dst.ctypes.data, dst.size, C.byref(n))
        return rc, dst[:n.value]

    def encode(codec, data):
        src = np.ascontiguousarray(data).view(np.uint8).reshape(-1).copy()
        dst = np.empty(max(1, host.bsmi_codec_bound(C.byref(codec), src.size)), dtype=np.uint8)
        n = C.c_size_t()
        rc = host.bsmi_codec_encode(C.byref(codec), src.ctypes.data, src.size, dst.ctypes.data, dst.size, C.byref(n))
        assert rc == 0
        return dst[:n.value].copy()

    # ---- 1. codecs: third-party frames (tests/golden/codec_cases.npz), then the same frames damaged -------------------------
    g = np.load(os.path.join(ROOT, "tests", "golden", "codec_cases.npz"))
    plain = {k[len("plain__"):]: g[k] for k in g.files if k.startswith("plain__")}
    frames = {k: g[k] for k in g.files if not k.startswith("plain__")}
    by_kind = {"blosc": Codec(BLOSC, 5, 1, 1, 1, 0), "zstd": Codec(ZSTD, 1, 0, 0, 1, 0), "lz4": Codec(LZ4, 1, 0, 0, 1, 0)}
    n_ok = n_bad = n_rejected = 0
    for key, frame in frames.items():
        kind, pname = key.split("__")[:2]
        want = plain[pname]
        rc, out = decode(by_kind[kind], frame, want.size)          # capacity = exactly the decoded size
        assert rc == 0 and np.array_equal(out, want), key
        n_ok += 1
        if want.size:
            rc, _ = decode(by_kind[kind], frame, max(0, want.size - 1 - int(rng.integers(0, min(64, want.size)))))   # too small a buffer
            assert rc != 0, key
        fb = np.frombuffer(bytes(frame), dtype=np.uint8)
        for trial in range(6):                                      # truncated, bit-flipped, overwritten
            bad = fb.copy()
            if trial < 2 and bad.size > 1:
                bad = bad[: int(rng.integers(0, bad.size))]
            elif trial < 4 and bad.size:
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, bad.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif bad.size > 8:
                a = int(rng.integers(0, bad.size - 4))
                bad[a:a + 4] = rng.integers(0, 256, 4, dtype=np.uint8)
            rc, out = decode(by_kind[kind], bad, want.size)
            n_bad += 1
            n_rejected += rc != 0
    # round trips of the encoders (every codec, shuffles, item sizes, odd lengths)
    payloads = [np.zeros(0, np.uint8), rng.integers(0, 256, 1, dtype=np.uint8), rng.integers(0, 4, 70001, dtype=np.uint8),
                np.repeat(rng.integers(0, 1 << 40, 3000, dtype=np.uint64), 7), rng.random(4099).astype(np.float32),
                rng.integers(0, 256, 300000, dtype=np.uint8)]
    for p in payloads:
        for codec in (Codec(RAW, 0, 0, 0, 1, 0), Codec(ZLIB, 1, 0, 0, 1, 0), Codec(GZIP, 1, 0, 0, 1, 0), Codec(ZSTD, 1, 0, 0, 1, 0),
                      Codec(LZ4, 1, 0, 0, 1, 0), Codec(BLOSC, 5, 1, 1, p.itemsize, 0), Codec(BLOSC, 5, 1, 2, p.itemsize, 0),
                      Codec(BLOSC, 5, 1, 0, p.itemsize, 0), Codec(BLOSC, 3, 4, 1, p.itemsize, 4096), Codec(BLOSC, 5, 3, 1, p.itemsize, 0)):
            raw = p.view(np.uint8).reshape(-1)
            rc, out = decode(codec, encode(codec, p), raw.size)
            assert rc == 0 and np.array_equal(out, raw), (codec.id, codec.cname, codec.shuffle, p.dtype, p.size)
    print(f"codecs: {n_ok} golden frames decoded, {n_rejected} of {n_bad} damaged frames rejected (the others decode to other bytes), {len(payloads) * 10} round trips")

    # ---- 2. merge loops ----------------------------------------------------------------------------------------------------
    from oracle import seg_ref as S            # BSMI_ORACLE_SO: the sanitizer build
    from scipy.ndimage import gaussian_filter
    from test_host_logic import _numpy_region_graph
    n_graphs = 0
    for levels, bins, thr in ((0, 256, 1.0), (2, 256, 1.0), (3, 16, 1.0), (4, 1, 1.0), (0, 256, 0.4)):
        a = gaussian_filter(rng.random((3, 6, 56, 60)), sigma=(0, 1, 2, 2))
        a = (a - a.min()) / (a.max() - a.min())
        if levels:
            a = np.round(np.clip((a - 0.5) * 2.5 + 0.5, 0, 1) * (levels - 1)) / (levels - 1)
        affs = (a * 255).astype(np.uint8)
        frags, _ = S.ws_fragments_u8(affs, True, 3)
        e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, thr, bins)
        e, sums, cnts = _numpy_region_graph(affs, frags)
        assert np.array_equal(e, e_ref)
        perm = rng.permutation(len(e))
        E, Sm, Cn = e[perm].copy(), sums[perm].copy(), cnts[perm].copy()
        sc = np.full(len(e), np.nan, np.float32)
        ne = np.array([len(e)], np.uint64)
        one = lambda arr: (C.c_void_p * 1)(arr.ctypes.data)
        rc = host.bsmi_rag_merge_scores_host(1, ne.ctypes.data, one(E), one(Sm), one(Cn), thr, bins, one(sc), 2)
        assert rc == 0 and np.array_equal(E, e) and np.array_equal(sc.view(np.uint32), s_ref.view(np.uint32)), (levels, bins, thr)
        # histogram-quantile loop on the same graph
        ids = np.unique(frags[frags > 0])
        rank = {int(v): i for i, v in enumerate(ids)}
        eu = np.array([rank[int(x)] for x in e[:, 0]], np.uint32)
        ev = np.array([rank[int(x)] for x in e[:, 1]], np.uint32)
        hist = np.zeros((len(e), 256), np.uint32)
        key_of = {(int(x), int(y)): i for i, (x, y) in enumerate(e)}
        for d in range(3):
            hi = [slice(None)] * 3; lo = [slice(None)] * 3
            hi[d], lo[d] = slice(1, None), slice(None, -1)
            fa, fb, w = frags[tuple(hi)], frags[tuple(lo)], affs[d][tuple(hi)]
            m = (fa > 0) & (fb > 0) & (fa != fb)
            for x, y, v in zip(np.minimum(fa[m], fb[m]), np.maximum(fa[m], fb[m]), w[m]):
                hist[key_of[(int(x), int(y))], int(v)] += 1
        thr3 = np.array([0.2, 0.5, 0.8], np.float32)
        for q, initmax in ((50, 0), (75, 1)):
            roots = np.zeros((3, len(ids)), np.uint32)
            h2 = hist.copy()
            rc = host.bsmi_agglomerate_hist_graph(len(ids), len(e), eu.ctypes.data, ev.ctypes.data, h2.ctypes.data, q, initmax, thr3.ctypes.data, 3, roots.ctypes.data)
            assert rc == 0
            want = S.agglomerate_hist_u8(affs, frags, thr3, q, bool(initmax))
            for t in range(3):
                lut = np.zeros(int(ids.max()) + 1, np.uint64)
                lut[ids.astype(np.int64)] = ids[roots[t]]
                assert np.array_equal(lut[frags.astype(np.int64)], want[t]), (levels, q, initmax, t)
        n_graphs += 1
    print(f"merge loops: {n_graphs} graphs, mean scorer (bin queue) and two histogram-quantile scorers each, equal to the oracle")

    # ---- 3. the 3-D flood ---------------------------------------------------------------------------------------------------
    D, H, W = 7, 30, 26
    mask = (gaussian_filter(rng.random((D, H, W)), 1.5) > 0.49).astype(np.uint8)
    d2 = (rng.integers(1, 60, (D, H, W)) * mask).astype(np.int32)
    lab = np.zeros((D, H, W), np.int32)
    seeds = np.argwhere(mask)[rng.choice(int(mask.sum()), 12, replace=False)]
    for i, (z, y, x) in enumerate(seeds):
        lab[z, y, x] = i + 1
    assert host.bsmi_san_host_flood3(D, H, W, mask.ctypes.data, d2.ctypes.data, lab.ctypes.data) == 0
    from scipy.ndimage import label as cc_label
    comp, _ = cc_label(mask)
    reached = np.isin(comp, np.unique(comp[tuple(seeds.T)]))
    assert np.array_equal(lab > 0, reached) and not (lab[mask == 0]).any()
    print("flood: every voxel of a seeded component labelled, nothing outside the mask")

    # ---- 4. the oracle's remaining entry points -------------------------------------------------------------------------------
    affs = (gaussian_filter(rng.random((3, 9, 40, 44)), sigma=(0, 1, 2, 2)) * 4 % 1 * 255).astype(np.uint8)
    for xy in (True, False):
        frags, mx = S.ws_fragments_u8(affs, xy, 4)
        assert frags.max() <= mx
    frags, _ = S.ws_fragments_u8(affs, True, 4)
    S.agglomerate_mean_u8(affs, frags, [0.2, 0.6])
    keep = S.filter_fragments_u8(affs, frags, 0.3, 8) if hasattr(S, "filter_fragments_u8") else None
    if hasattr(S, "label26"):
        S.label26(frags)
    if hasattr(S, "cc_affs_u8"):
        S.cc_affs_u8(affs, 0.5)
    e_ref, s_ref, _, _ = S.rag_merge_scores_u8(affs, frags, 1.0, 256)
    if hasattr(S, "connected_components"):
        nodes = np.unique(frags[frags > 0])
        S.connected_components(nodes, e_ref, np.nan_to_num(s_ref, nan=2.0), 0.5)
    print("oracle: fragments (both modes), agglomeration, clean-up, labelling, edge scoring, connected components ran clean")
    print("SANITIZERS OK")


if __name__ == "__main__":
    main()
