"""Handle around the libbsmi segmentation workspace (one per host thread per GPU)."""
import ctypes as C

import torch

from .. import _lib
from .._lib import lib, check


class SegEngine:
    def __init__(self, max_shape, device=0, host_flood=True):
        """host_flood: the one sequential flood of the 3-D fragments mode (fragments_in_xy = False) runs on the host and
        ws_fragments returns when it is done (0.16 s per 128^3 block; the device's single-wave replay of the same loop takes
        12.8 s, 0.8 s per block with 16 lanes side by side); False keeps the call asynchronous on the device."""
        self.device = int(device)
        self.max_shape = tuple(int(s) for s in max_shape)
        self._h = C.c_void_p()
        check(lib.bsmi_seg_create(self.device, _lib.i64x3(self.max_shape), C.byref(self._h)))
        check(lib.bsmi_seg_set_host_flood(self._h, 1 if host_flood else 0))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.bsmi_seg_destroy(h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)

    def ws_fragments(self, affs_u8, fragments_in_xy=True, min_seed_distance=10, return_seeds=False):
        """affs_u8: uint8 CUDA tensor [3][D][H][W] -> (fragments int64 [D][H][W] holding the
        uint64 ids, max_id tensor int64[1][, seeds int64 [D][H][W]]); asynchronous on the current stream (except the 3-D
        mode of an engine with host_flood, which returns when the fragments are written)."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        a = affs_u8.contiguous()
        shape = a.shape[1:]
        frags = torch.empty(tuple(shape), dtype=torch.int64, device=a.device)
        max_id = torch.zeros(1, dtype=torch.int64, device=a.device)
        seeds = torch.empty(tuple(shape), dtype=torch.int64, device=a.device) if return_seeds else None
        check(lib.bsmi_ws_fragments_seeds_u8(self._h, C.c_void_p(a.data_ptr()), _lib.i64x3(shape),
                                             1 if fragments_in_xy else 0, int(min_seed_distance),
                                             C.c_void_p(frags.data_ptr()), C.c_void_p(max_id.data_ptr()),
                                             C.c_void_p(seeds.data_ptr()) if return_seeds else None, self._stream()))
        return (frags, max_id, seeds) if return_seeds else (frags, max_id)

    def agglomerate_mean(self, affs_u8, frags, thresholds):
        """-> int64 CUDA tensor [len(thresholds)][D][H][W]; asynchronous on the current stream
        (call status() to synchronise and check the workspace did not overflow)."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]):
            raise ValueError("fragments must be an int64 tensor of shape (D, H, W)")
        a = affs_u8.contiguous()
        f = frags.contiguous()
        thr = (C.c_float * len(thresholds))(*[float(t) for t in thresholds])
        segs = torch.empty((len(thresholds),) + tuple(f.shape), dtype=torch.int64, device=a.device)
        check(lib.bsmi_agglomerate_mean_u8(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(f.data_ptr()),
                                           _lib.i64x3(f.shape), thr, len(thresholds),
                                           C.c_void_p(segs.data_ptr()), self._stream()))
        return segs

    def agglomerate_hist(self, affs_u8, frags, thresholds, quantile, init_with_max=False):
        """agglomerate_mean with OneMinus<HistogramQuantileAffinity<., quantile, ., 256, init_with_max>> as the scorer
        (reference post/watershed.py:230-243); the merge loop runs on the host: synchronises."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]):
            raise ValueError("fragments must be an int64 tensor of shape (D, H, W)")
        a = affs_u8.contiguous()
        f = frags.contiguous()
        thr = (C.c_float * len(thresholds))(*[float(t) for t in thresholds])
        segs = torch.empty((len(thresholds),) + tuple(f.shape), dtype=torch.int64, device=a.device)
        check(lib.bsmi_agglomerate_hist_u8(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(f.data_ptr()), _lib.i64x3(f.shape), thr,
                                           len(thresholds), int(quantile), 1 if init_with_max else 0, C.c_void_p(segs.data_ptr()),
                                           self._stream()))
        return segs

    def postprocess_fragments(self, affs_u8, frags, filter_value=0.0, min_size=0, crop_offset=(0, 0, 0),
                              crop_shape=None, id_offset=0, out=None, num=None):
        """Blockwise fragment clean-up (reference post/blockwise/watershed_frags.py:181-224): filter
        by mean affinity, drop debris, crop to the write ROI, relabel 26-connected components in
        raster order and add `id_offset`.  `frags` is filtered in place.  -> (labels int64
        [crop_shape], num_labels int64[1]); asynchronous on the current stream."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]) or not frags.is_contiguous():
            raise ValueError("fragments must be a contiguous int64 tensor of shape (D, H, W)")
        a = affs_u8.contiguous()
        shape = tuple(frags.shape)
        crop_shape = tuple(shape) if crop_shape is None else tuple(int(c) for c in crop_shape)
        # out / num: caller-owned destinations (a pipeline that keeps every block's results without a host round trip)
        if out is None:
            out = torch.empty(crop_shape, dtype=torch.int64, device=a.device)
        elif out.dtype != torch.int64 or tuple(out.shape) != crop_shape or not out.is_contiguous():
            raise ValueError("out must be a contiguous int64 tensor of the crop shape")
        if num is None:
            num = torch.zeros(1, dtype=torch.int64, device=a.device)
        check(lib.bsmi_frag_postprocess_u8(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(frags.data_ptr()),
                                           _lib.i64x3(shape), float(filter_value), int(min_size),
                                           _lib.i64x3(crop_offset), _lib.i64x3(crop_shape), int(id_offset),
                                           C.c_void_p(out.data_ptr()), C.c_void_p(num.data_ptr()), self._stream()))
        return out, num

    def label_stats(self, labels, id_offset, num, size=None, sums=None):
        """Voxel count and z/y/x index sums of labels id_offset+1..id_offset+num (RAG node
        attributes, watershed_frags.py:230-246) -> (size int64[num], sums int64[num][3])."""
        if labels.dtype != torch.int64 or labels.dim() != 3 or not labels.is_cuda:
            raise ValueError("labels must be an int64 CUDA tensor of shape (D, H, W)")
        lab = labels.contiguous()
        num = int(num)
        # size / sums: caller-owned destinations of at least `num` entries (`num` may then be an upper bound)
        if size is None:
            size = torch.empty(max(num, 1), dtype=torch.int64, device=lab.device)
        if sums is None:
            sums = torch.empty((max(num, 1), 3), dtype=torch.int64, device=lab.device)
        if size.numel() < num or sums.numel() < 3 * num or not size.is_contiguous() or not sums.is_contiguous():
            raise ValueError("size / sums too small")
        check(lib.bsmi_label_stats(self._h, C.c_void_p(lab.data_ptr()), _lib.i64x3(lab.shape), int(id_offset), num,
                                   C.c_void_p(size.data_ptr()), C.c_void_p(sums.data_ptr()), self._stream()))
        return size[:num], sums[:num]

    def rag_merge_scores(self, affs_u8, frags, threshold=1.0, discretize_queue=256, return_merges=False):
        """Initial RAG edges of a block and the score at which each edge's fragments merge
        (reference post/blockwise/waterz_agglom.py:106-170).  Synchronises.  -> (edges int64 [ne][2]
        holding the uint64 ids, scores float32 [ne] with NaN = never merged[, merges int64 [nm][2],
        merge_scores float32 [nm]]) as CUDA tensors."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]):
            raise ValueError("fragments must be an int64 tensor of shape (D, H, W)")
        a, f = affs_u8.contiguous(), frags.contiguous()
        cap = self._rag_cap = getattr(self, "_rag_cap", None) or max(1024, f.numel() // 4)
        dev = a.device
        edges = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        scores = torch.empty(cap, dtype=torch.float32, device=dev)
        ncap = max(1024, f.numel() // 8 + 1024)
        merges = torch.empty((ncap, 2), dtype=torch.int64, device=dev) if return_merges else None
        mscores = torch.empty(ncap, dtype=torch.float32, device=dev) if return_merges else None
        counts = torch.zeros(4, dtype=torch.int64, device=dev)
        check(lib.bsmi_rag_merge_scores_u8(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(f.data_ptr()), _lib.i64x3(f.shape),
                                           float(threshold), int(discretize_queue), C.c_void_p(edges.data_ptr()),
                                           C.c_void_p(scores.data_ptr()), cap,
                                           C.c_void_p(merges.data_ptr()) if return_merges else None,
                                           C.c_void_p(mscores.data_ptr()) if return_merges else None,
                                           C.c_void_p(counts.data_ptr()), self._stream()))
        self.status()
        ne, nm = int(counts[0]), int(counts[1])
        if return_merges:
            return edges[:ne], scores[:ne], merges[:nm], mscores[:nm]
        return edges[:ne], scores[:ne]

    def rag_agglomerate(self, affs_u8, frags, threshold, discretize_queue=256):
        """Epsilon agglomeration of `frags` IN PLACE (reference watershed_frags.py:158-177); asynchronous."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3 or not affs_u8.is_contiguous():
            raise ValueError("affs must be a contiguous uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]) or not frags.is_contiguous():
            raise ValueError("fragments must be a contiguous int64 tensor of shape (D, H, W)")
        check(lib.bsmi_rag_agglomerate_u8(self._h, C.c_void_p(affs_u8.data_ptr()), C.c_void_p(frags.data_ptr()), _lib.i64x3(frags.shape),
                                          float(threshold), int(discretize_queue), self._stream()))
        return frags

    def rag_edge_stats(self, n_edges):
        """(affinity sums, voxel-pair counts) int64 [n_edges] of the initial edges of the last rag_merge_scores call."""
        dev = torch.device("cuda", self.device)
        sums = torch.zeros(max(1, n_edges), dtype=torch.int64, device=dev)
        counts = torch.zeros(max(1, n_edges), dtype=torch.int64, device=dev)
        check(lib.bsmi_rag_edge_stats(self._h, C.c_void_p(sums.data_ptr()), C.c_void_p(counts.data_ptr()), int(n_edges), self._stream()))
        self.status()
        return sums[:n_edges].cpu().numpy(), counts[:n_edges].cpu().numpy()

    def rag_merge_scores_async(self, affs_u8, frags, threshold, discretize_queue, edges, scores, counts, merges=None,
                               merge_scores=None):
        """rag_merge_scores into caller-owned CUDA buffers, asynchronous on the current stream: edges int64 [cap][2],
        scores float32 [cap], counts int64 [>= 3] (edges, merges, nodes); status() reports a too small buffer."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3 or not affs_u8.is_contiguous():
            raise ValueError("affs must be a contiguous uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]) or not frags.is_contiguous():
            raise ValueError("fragments must be a contiguous int64 tensor of shape (D, H, W)")
        if not (edges.is_contiguous() and scores.is_contiguous() and counts.is_contiguous()) or edges.shape[0] != scores.shape[0]:
            raise ValueError("edges / scores / counts must be contiguous and of one capacity")
        check(lib.bsmi_rag_merge_scores_u8(self._h, C.c_void_p(affs_u8.data_ptr()), C.c_void_p(frags.data_ptr()),
                                           _lib.i64x3(frags.shape), float(threshold), int(discretize_queue),
                                           C.c_void_p(edges.data_ptr()), C.c_void_p(scores.data_ptr()), int(edges.shape[0]),
                                           C.c_void_p(merges.data_ptr()) if merges is not None else None,
                                           C.c_void_p(merge_scores.data_ptr()) if merge_scores is not None else None,
                                           C.c_void_p(counts.data_ptr()), self._stream()))

    def rag_graph_async(self, affs_u8, frags, edges, sums, pair_counts, counts):
        """The block's region graph without the merge loop, into caller-owned CUDA buffers, asynchronous on the current stream:
        edges int64 [cap][2] (id pairs, ascending), sums int64 [cap], pair_counts int32 [cap], counts int64 [>= 3] (edges, 0,
        nodes); status() reports a too small buffer.  `rag_merge_scores_host` scores such graphs on the host."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3 or not affs_u8.is_contiguous():
            raise ValueError("affs must be a contiguous uint8 CUDA tensor of shape (3, D, H, W)")
        if frags.dtype != torch.int64 or tuple(frags.shape) != tuple(affs_u8.shape[1:]) or not frags.is_contiguous():
            raise ValueError("fragments must be a contiguous int64 tensor of shape (D, H, W)")
        if (edges.dtype != torch.int64 or sums.dtype != torch.int64 or pair_counts.dtype != torch.int32 or counts.dtype != torch.int64
                or not (edges.is_contiguous() and sums.is_contiguous() and pair_counts.is_contiguous() and counts.is_contiguous())
                or not edges.shape[0] == sums.shape[0] == pair_counts.shape[0]):
            raise ValueError("edges int64 [cap][2], sums int64 [cap], pair_counts int32 [cap], counts int64: contiguous, one capacity")
        check(lib.bsmi_rag_graph_u8(self._h, C.c_void_p(affs_u8.data_ptr()), C.c_void_p(frags.data_ptr()), _lib.i64x3(frags.shape),
                                    C.c_void_p(edges.data_ptr()), C.c_void_p(sums.data_ptr()), C.c_void_p(pair_counts.data_ptr()),
                                    int(edges.shape[0]), C.c_void_p(counts.data_ptr()), self._stream()))

    def cc_affs(self, affs_u8, threshold=0.5, remove_debris=0):
        """Thresholded-affinity connected components (reference post/cc.py; post/connected_components.py:77-101).
        -> (fragments int64, segmentation int64 (debris removed), count int64[1]); asynchronous."""
        if affs_u8.dtype != torch.uint8 or not affs_u8.is_cuda or affs_u8.dim() != 4 or affs_u8.shape[0] != 3:
            raise ValueError("affs must be a uint8 CUDA tensor of shape (3, D, H, W)")
        a = affs_u8.contiguous()
        shape = tuple(a.shape[1:])
        frags = torch.empty(shape, dtype=torch.int64, device=a.device)
        seg = torch.empty(shape, dtype=torch.int64, device=a.device)
        num = torch.zeros(1, dtype=torch.int64, device=a.device)
        check(lib.bsmi_cc_affs_u8(self._h, C.c_void_p(a.data_ptr()), _lib.i64x3(shape), threshold_to_cut(threshold), int(remove_debris),
                                  C.c_void_p(frags.data_ptr()), C.c_void_p(seg.data_ptr()), C.c_void_p(num.data_ptr()), self._stream()))
        return frags, seg, num

    def label_table(self, labels, z0=0):
        """Distinct non-zero ids (ascending), voxel counts and first / last z slice of an int64 CUDA label block
        (`bs refine` statistics, reference refine.py:98-109, 228-250).  Synchronises.  -> numpy (ids u64, counts, zmin, zmax)."""
        if labels.dtype != torch.int64 or labels.dim() != 3 or not labels.is_cuda:
            raise ValueError("labels must be an int64 CUDA tensor of shape (D, H, W)")
        lab = labels.contiguous()
        cap = max(1024, lab.numel() // 8 + 1024)
        dev = lab.device
        ids = torch.empty(cap, dtype=torch.int64, device=dev)
        counts = torch.empty(cap, dtype=torch.int64, device=dev)
        zmin = torch.empty(cap, dtype=torch.int32, device=dev)
        zmax = torch.empty(cap, dtype=torch.int32, device=dev)
        n = torch.zeros(1, dtype=torch.int64, device=dev)
        check(lib.bsmi_label_table_u64(self._h, C.c_void_p(lab.data_ptr()), _lib.i64x3(lab.shape), int(z0), C.c_void_p(ids.data_ptr()),
                                       C.c_void_p(counts.data_ptr()), C.c_void_p(zmin.data_ptr()), C.c_void_p(zmax.data_ptr()), cap,
                                       C.c_void_p(n.data_ptr()), self._stream()))
        self.status()
        k = int(n.item())
        return (ids[:k].cpu().numpy().view("uint64"), counts[:k].cpu().numpy(), zmin[:k].cpu().numpy(), zmax[:k].cpu().numpy())

    def status(self):
        check(lib.bsmi_seg_status(self._h, self._stream()))


QUEUE_BINS_FORMULAS = {"n_minus_1": 0, "n": 1}   # include/bsmi.h BSMI_QUEUE_BINS_*


def rag_merge_scores_host(n_edges, edges, sums, pair_counts, threshold=1.0, discretize_queue=256, threads=0, bins_formula="n_minus_1"):
    """waterz_agglom.py:106-170 for many blocks at once on host threads (csrc/agglo_host.cpp): graphs as SegEngine.rag_graph_async
    exports them, copied to the host -- edges uint64 / int64 [G][cap][2], sums [G][cap], pair_counts uint32 / int32 [G][cap]
    (numpy, C-contiguous, writable), n_edges [G].  bins_formula: "n_minus_1" (bin = (int)(score (N-1)), the specification) or "n"
    (min(N-1, (int)(score N))): waterz's own rule is unpinned.  The device leaves a graph's edges in the order of its hash table: every graph
    is first sorted by (id, id) IN PLACE -- edges, sums and pair_counts permuted together --, which is the order the scores come
    back in.  -> scores float32 [G][cap] (entries past n_edges[g] untouched: NaN)."""
    import numpy as np
    ne = np.ascontiguousarray(n_edges, dtype=np.uint64)
    G = int(ne.shape[0])
    for a in (edges, sums, pair_counts):
        if not (isinstance(a, np.ndarray) and a.flags.c_contiguous and a.flags.writeable):
            raise ValueError("edges, sums and pair_counts must be C-contiguous writable numpy arrays (they are sorted in place)")
    e = edges.view(np.uint64)
    s = sums.view(np.uint64)
    c = pair_counts.view(np.uint32)
    if e.shape[:1] != (G,) or e.ndim != 3 or e.shape[2] != 2 or s.shape != e.shape[:2] or c.shape != e.shape[:2] or int(ne.max(initial=0)) > e.shape[1]:
        raise ValueError("edges [G][cap][2], sums [G][cap], pair_counts [G][cap], n_edges [G] <= cap")
    scores = np.full(e.shape[:2], np.nan, dtype=np.float32)
    ptrs = lambda a, stride: (C.c_void_p * G)(*[a.ctypes.data + g * stride for g in range(G)])
    if bins_formula not in QUEUE_BINS_FORMULAS:
        raise ValueError(f"queue_bins_formula {bins_formula!r}: one of {sorted(QUEUE_BINS_FORMULAS)}")
    check(lib.bsmi_rag_merge_scores_host_rule(G, ne.ctypes.data_as(C.c_void_p), ptrs(e, e.strides[0]), ptrs(s, s.strides[0]), ptrs(c, c.strides[0]),
                                              float(threshold), int(discretize_queue), QUEUE_BINS_FORMULAS[bins_formula],
                                              ptrs(scores, scores.strides[0]), int(threads)))
    return scores


def lut_relabel(labels, keys, vals, out=None):
    """out[p] = vals[k] where keys[k] == labels[p] (keys ascending int64/uint64 ids); volara Relabel
    (reference post/watershed.py:187-202).  CUDA int64 tensors; asynchronous on the current stream."""
    if labels.dtype != torch.int64 or not labels.is_cuda:
        raise ValueError("labels must be an int64 CUDA tensor")
    lab = labels.contiguous()
    keys = keys.to(device=lab.device, dtype=torch.int64).contiguous()
    vals = vals.to(device=lab.device, dtype=torch.int64).contiguous()
    if keys.numel() != vals.numel():
        raise ValueError("keys and vals differ in length")
    out = torch.empty_like(lab) if out is None else out
    stream = C.c_void_p(torch.cuda.current_stream(lab.device).cuda_stream)
    check(lib.bsmi_lut_relabel(lab.device.index, C.c_void_p(lab.data_ptr()), lab.numel(), C.c_void_p(keys.data_ptr()),
                               C.c_void_p(vals.data_ptr()), keys.numel(), C.c_void_p(out.data_ptr()), stream))
    return out


def lut_relabel_multi(labels, keys, vals, out=None):
    """lut_relabel for several value columns at once -- one segmentation per threshold out of one fragment volume: vals int64
    [T][m] (CUDA or host), -> out int64 [T] + labels.shape.  One look-up per run of equal ids serves every column."""
    if labels.dtype != torch.int64 or not labels.is_cuda:
        raise ValueError("labels must be an int64 CUDA tensor")
    lab = labels.contiguous()
    keys = keys.to(device=lab.device, dtype=torch.int64).contiguous()
    vals = vals.to(device=lab.device, dtype=torch.int64).contiguous()
    if vals.dim() != 2 or vals.shape[1] != keys.numel():
        raise ValueError("vals must have shape (columns, len(keys))")
    T = int(vals.shape[0])
    if out is None:
        out = torch.empty((T,) + tuple(lab.shape), dtype=torch.int64, device=lab.device)
    if not out.is_contiguous() or tuple(out.shape) != (T,) + tuple(lab.shape) or out.dtype != torch.int64:
        raise ValueError("out must be a contiguous int64 tensor of shape (columns,) + labels.shape")
    stream = C.c_void_p(torch.cuda.current_stream(lab.device).cuda_stream)
    check(lib.bsmi_lut_relabel_multi(lab.device.index, C.c_void_p(lab.data_ptr()), lab.numel(), C.c_void_p(keys.data_ptr()),
                                     C.c_void_p(vals.data_ptr()), keys.numel(), T, C.c_void_p(out.data_ptr()), stream))
    return out


def threshold_to_cut(threshold):
    """The uint8 cut equivalent to the reference's `affs.astype(float32) / 255.0 > threshold`
    (post/connected_components.py:49-52,77): the largest v with not (float32(v) / 255 > float32(threshold)), -1 if none."""
    import numpy as np
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    below = np.nonzero(~(v > np.float32(threshold)))[0]
    return int(below.max()) if below.size else -1
