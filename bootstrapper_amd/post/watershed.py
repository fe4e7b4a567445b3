"""`bs segment --ws` driver on the device.

Behavioural mirror of /root/reference/bootstrapper/post/watershed.py:206-366
(`simple_watershed`, `watershed_segmentation`): read the first three affinity channels of the ROI,
optional mask, fragments -> `<fragments_dataset>/<build_name(frag_params)>`, agglomeration at every
threshold -> `<seg_dataset_prefix>/<build_name(params)>`, parameters recorded as `bs_params`.
The arithmetic runs in libbsmi (bootstrapper_amd.post.ws / .waterz).  `waterz_pipeline` mirrors
post/watershed.py:8-203: blockwise fragments with context, per-block RAG edge scoring, global
thresholded connected components, LUT, relabel (bootstrapper_amd.post.blockwise).
"""
import os

import numpy as np

from ..zarr_io import open_ds, prepare_ds
from .naming import build_name, dump_params


def _warn_3d_fragments(fragments_in_xy, shape):
    """fragments_in_xy = false (post/ws.py:98-110) floods a whole block as ONE sequential priority queue: bit-exact here, on the
    host (csrc/flood_host.cpp: 0.16 s per 128^3 block, one block at a time) where the per-section default takes 12 ms on the
    device with 16 blocks in flight -- say so instead of looking slow."""
    if not fragments_in_xy and int(np.prod(shape)) > (1 << 21):
        import sys
        print(f"warning: fragments_in_xy = false floods a block of {tuple(int(v) for v in shape)} voxels as one sequential queue "
              "on the host (0.2 s per 128^3 block, block after block); the default fragments_in_xy = true runs every section as its own queue on the device",
              file=sys.stderr)


def simple_watershed(config, device=0):
    import torch
    from .ws import watershed_from_affinities
    from .waterz import agglomerate

    affs = open_ds(config["affs_dataset"])
    thresholds = config.get("thresholds", [0.2, 0.35, 0.5])
    fragments_in_xy = config.get("fragments_in_xy", True)
    min_seed_distance = config.get("min_seed_distance", 10)
    merge_function = config.get("merge_function", "mean")
    sigma, noise_eps, bias = config.get("sigma"), config.get("noise_eps"), config.get("bias")
    from .waterz import MERGE_FUNCTIONS
    if merge_function not in MERGE_FUNCTIONS:   # the reference's table (post/watershed.py:230-243) raises KeyError here
        raise KeyError(merge_function)
    if affs.dtype != np.uint8:
        raise NotImplementedError("the device path takes uint8 affinities (what `bs predict` stores)")

    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    sl = affs.roi_to_slices(*roi)
    data = affs[sl][:3]
    if data.shape[0] == 2:  # 2-channel affinities get an all-zero z channel (post/watershed.py:305-308)
        data = np.concatenate([np.zeros_like(data[:1]), data])
    dev = torch.device("cuda", device)
    a = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    if config.get("mask_dataset"):
        mask = open_ds(config["mask_dataset"])
        m = torch.from_numpy((mask[mask.roi_to_slices(*roi)] > 0).astype(np.uint8)).to(dev)
        a = a * m

    frag_params = {"fragments_in_xy": fragments_in_xy, "min_seed_distance": min_seed_distance,
                   "sigma": sigma, "noise_eps": noise_eps, "bias": bias}
    _warn_3d_fragments(fragments_in_xy, a.shape[1:])
    if any([sigma, noise_eps, bias]):   # post/watershed.py:285-303: the watershed sees the shifted affinities
        from .shifts import boundary_mask_affinities
        src = boundary_mask_affinities(a, fragments_in_xy, sigma, noise_eps, bias, dtype=torch.float32)
    else:
        src = a
    frags, _ = watershed_from_affinities(src, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance)
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    common = dict(offset=roi[0], voxel_size=affs.voxel_size, axis_names=affs.axis_names[1:], units=affs.units,
                  dtype=np.uint64)
    out = prepare_ds(frags_name, shape=frags.shape, **common)
    out[:] = frags.cpu().numpy().astype(np.uint64)
    dump_params(frags_name, {"method": "ws", "blockwise": False, **frag_params})

    written = [frags_name]
    for threshold, seg in zip(thresholds, agglomerate(a, thresholds, fragments=frags.clone(),     # a copy, as post/watershed.py:336
                                                      scoring_function=MERGE_FUNCTIONS[merge_function])):
        params = {"merge_function": merge_function, "threshold": threshold, **frag_params}
        seg_name = os.path.join(config["seg_dataset_prefix"], build_name(params))
        out = prepare_ds(seg_name, shape=seg.shape, **common)
        out[:] = seg.cpu().numpy().astype(np.uint64)
        dump_params(seg_name, {"method": "ws", "blockwise": False, **params})
        written.append(seg_name)
    return written


def connected_components(nodes, edges, scores, threshold):
    """`funlib.segment.graphs.impl.connected_components` (post/watershed.py:182) through the C ABI (host code)."""
    import ctypes as C
    from .._lib import lib, check
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    edges = np.ascontiguousarray(edges, dtype=np.uint64).reshape(-1, 2)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    out = np.zeros(len(nodes), dtype=np.uint64)
    check(lib.bsmi_connected_components(C.c_void_p(nodes.ctypes.data), len(nodes), C.c_void_p(edges.ctypes.data),
                                        C.c_void_p(scores.ctypes.data), len(scores), float(threshold),
                                        C.c_void_p(out.ctypes.data)))
    return out


def connected_components_multi(nodes, edges, scores, thresholds):
    """connected_components for every threshold in one library call -> uint64 [len(thresholds)][len(nodes)]."""
    import ctypes as C
    from .._lib import lib, check
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    edges = np.ascontiguousarray(edges, dtype=np.uint64).reshape(-1, 2)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    out = np.zeros((len(thr), len(nodes)), dtype=np.uint64)
    check(lib.bsmi_connected_components_multi(C.c_void_p(nodes.ctypes.data), len(nodes), C.c_void_p(edges.ctypes.data),
                                              C.c_void_p(scores.ctypes.data), len(scores), C.c_void_p(thr.ctypes.data), len(thr),
                                              C.c_void_p(out.ctypes.data)))
    return out


IO_WORKERS = 8   # pieces in flight between the store and the device (each request decodes / encodes its chunks on up to
                 # zarr_io.IO_THREADS native threads, no GIL; with 3 the 32 GB of fragments + segmentations of a 1024^3 volume
                 # left in 2.2 s, bounded by the three requests' encoders)


PIECE_BYTES = 256 << 20   # staging buffers between the store and the device: page-locked, pooled for the life of the process


class _PinnedPool:
    """Page-locked staging buffers, handed out and taken back (locking a gigabyte takes the runtime a few tenths of a second:
    the readers' buffers serve the writers afterwards instead of each stage locking its own)."""

    def __init__(self):
        import threading
        self.free, self.lock = [], threading.Lock()

    def take(self, nbytes):
        import torch
        with self.lock:
            for i, b in enumerate(self.free):
                if b.numel() >= nbytes:
                    return self.free.pop(i)
        return torch.empty(max(int(nbytes), PIECE_BYTES), dtype=torch.uint8, pin_memory=True)

    def give(self, buf):
        with self.lock:
            self.free.append(buf)


_PINNED = _PinnedPool()


def _fill_affinities(seg, affs, origin, z0, mask=None, y0=0):
    """The box's affinities WITH their context margins straight from the dataset (`to_ndarray(read_roi, fill_value=0)`,
    watershed_frags.py:196-201: zeros beyond the array, real data beyond the ROI), first three channels, masked.
    Streamed: the reference's workers read block by block; here one layer of blocks at a time is decoded by the native
    chunk codecs -- the chunk's first three channels only, straight into a page-locked staging buffer (`read_into`: no
    chunk-sized arrays in the interpreter) -- and copied into the resident slab while the next layers are being read;
    the host never holds more than IO_WORKERS layers."""
    import concurrent.futures as cf
    import threading
    import torch
    from .blockwise import read_with_fill
    begin = tuple(o + lo - c for o, lo, c in zip(origin, (z0, y0, 0), seg.ctx))
    full = tuple(s + 2 * c for s, c in zip(seg.shape, seg.ctx))
    step = max(1, seg.block[0])
    nch = min(3, affs.shape[0])
    ch0 = 3 - nch    # 2-channel affinities get an all-zero z channel (post/watershed.py:305-308): the slab starts zeroed
    vol = affs.shape[-3:]
    staging, streams = {}, {}

    def load(za):
        zb = min(full[0], za + step)
        b = (begin[0] + za,) + begin[1:]
        e = (begin[0] + zb,) + tuple(bb + f for bb, f in zip(begin[1:], full[1:]))
        if mask is not None:   # (rare) the masked form through the interpreter
            a = read_with_fill(affs, b, e, lead=(affs.shape[0],))[:3]
            t = torch.from_numpy(np.ascontiguousarray(a)).to(seg.dev)
            m = read_with_fill(mask, b, e)
            t = t * torch.from_numpy((m > 0).astype(np.uint8)).to(seg.dev)
            seg.affs[ch0:, za:zb].copy_(t)
            torch.cuda.current_stream(seg.dev).synchronize()
            return
        lo = [max(bb, 0) for bb in b]
        hi = [min(ee, n) for ee, n in zip(e, vol)]
        if any(h <= l for l, h in zip(lo, hi)):
            return   # wholly outside the array: the slab is zero there already
        tid = threading.get_ident()
        if tid not in staging:
            staging[tid] = _PINNED.take(nch * step * full[1] * full[2])
            from ..volume import io_stream
            streams[tid] = io_stream(seg.dev)
        ext = tuple(h - l for l, h in zip(lo, hi))
        host = staging[tid][:nch * ext[0] * ext[1] * ext[2]].view((nch,) + ext)
        from .. import _trace
        with _trace.span("reader: decode", True):
            affs.read_into((slice(0, nch),) + tuple(slice(l, h) for l, h in zip(lo, hi)), host.numpy())
        dst = seg.affs[(slice(ch0, 3),) + tuple(slice(l - bb + off, h - bb + off) for l, h, bb, off in zip(lo, hi, b, (za, 0, 0)))]
        with torch.cuda.stream(streams[tid]):
            dst.copy_(host, non_blocking=True)
        streams[tid].synchronize()
    # pieces of at most PIECE_BYTES: whole block layers where they fit, else a layer in runs of sections
    step = max(1, min(step, PIECE_BYTES // max(1, nch * full[1] * full[2])))
    try:
        with cf.ThreadPoolExecutor(max_workers=IO_WORKERS, thread_name_prefix="bsmi-read") as pool:
            for f in [pool.submit(load, za) for za in range(0, full[0], step)]:
                f.result()
    finally:
        for b in staging.values():
            _PINNED.give(b)


class _LayerWriter:
    """Write-behind of a rank's resident volumes: a device tensor [Z][Y][X] goes to its dataset one layer of blocks at a
    time -- device -> host copy on a side stream into page-locked memory, then the library's threads gather, encode and write the
    layer's chunks straight from that buffer (`write_from`) -- while the caller carries on (the next stage's kernels, the next
    dataset).  `drain()` waits for everything and re-raises a failure."""

    def __init__(self, dev, step):
        import concurrent.futures as cf
        import torch
        self.dev, self.step = dev, max(1, int(step))
        self.pool = cf.ThreadPoolExecutor(max_workers=IO_WORKERS, thread_name_prefix="bsmi-write")
        self.streams = {}
        self.pinned = {}   # thread -> page-locked staging buffer (a pageable destination makes the runtime stage the copy itself)
        self.devbuf = {}   # thread -> (scratch, frame slots, frame sizes) of the device-side Blosc encoder
        self.futures = []
        self.torch = torch

    @staticmethod
    def _device_frames_ok(ds, z, y, part):
        """can this piece's chunks be made into Blosc frames on the device (csrc/blosc_dev.hip)?  The dataset must be what
        `prepare_ds` creates by default for ids (Blosc lz4, byte shuffle, 8-byte items, whole 256 KiB blocks per chunk), the piece
        must start on chunk boundaries and reach the array's end or a chunk boundary."""
        import os as _os
        from .. import _lib
        if _os.environ.get("BSMI_DEVICE_FRAMES", "1") == "0" or len(ds.shape) != 3 or ds.dtype.itemsize != 8:
            return False
        c = ds.codec
        if c.id != _lib.CODEC_BLOSC or c.cname != _lib.BLOSC_LZ4 or c.shuffle != 1 or c.blocksize not in (0, 256 * 1024) or c.level == 0:
            return False
        if ds.chunk_nbytes % (256 * 1024) or ds.chunk_nbytes // (256 * 1024) > 1024 or (ds.fill_value not in (0, None)):
            return False
        cz, cy, cx = ds.chunks
        if z % cz or y % cy or part.stride(2) != 1:
            return False
        return all((o + e) % c_ == 0 or o + e == n for o, e, c_, n in ((z, part.shape[0], cz, ds.shape[0]), (y, part.shape[1], cy, ds.shape[1])))             and part.shape[2] == ds.shape[2]

    def _write_device_frames(self, ds, part, z, y, st, tid):
        """the piece's chunks as Blosc frames made on the device: two launches, the frame sizes and then the frames themselves
        (a few percent of the piece) come to the host, the files are written from there"""
        import ctypes as C
        import os as _os
        from .. import _lib
        torch = self.torch
        cz, cy, cx = ds.chunks
        grid = [(iz, iy, ix) for iz in range(0, part.shape[0], cz) for iy in range(0, part.shape[1], cy) for ix in range(0, part.shape[2], cx)]
        n = len(grid)
        origins = (C.c_int64 * (3 * n))(*[v for g in grid for v in g])
        extents = (C.c_int64 * (3 * n))(*[min(c_, s - o) for g in grid for o, c_, s in zip(g, (cz, cy, cx), part.shape)])
        slot = int(_lib.lib.bsmi_blosc_dev_frame_bound(ds.chunk_nbytes))
        slot = (slot + 255) // 256 * 256
        need_s = int(_lib.lib.bsmi_blosc_dev_scratch_bytes(n, ds.chunk_nbytes))
        bufs = self.devbuf.get(tid)
        if bufs is None or bufs[0].numel() < need_s or bufs[1].numel() < n * slot:
            bufs = self.devbuf[tid] = (torch.empty(need_s, dtype=torch.uint8, device=self.dev), torch.empty(n * slot, dtype=torch.uint8, device=self.dev),
                                       torch.empty(max(n, 64), dtype=torch.int32, device=self.dev))
        scratch, frames, sizes = bufs
        if sizes.numel() < n:
            sizes = torch.empty(n, dtype=torch.int32, device=self.dev)
            self.devbuf[tid] = (scratch, frames, sizes)
        shape3 = (C.c_int64 * 3)(cz, cy, cx)
        _lib.check(_lib.lib.bsmi_blosc_encode_dev_u64(self.dev.index, C.c_void_p(part.data_ptr()), part.stride(0), part.stride(1), n, origins, extents, shape3,
                                                       C.c_void_p(scratch.data_ptr()), need_s, C.c_void_p(frames.data_ptr()), slot, C.c_void_p(sizes.data_ptr()),
                                                       C.c_void_p(st.cuda_stream)))
        with torch.cuda.stream(st):
            host_sizes = sizes[:n].to("cpu", non_blocking=False).numpy().astype(np.int64)
        total = int(host_sizes.sum())
        buf = self.pinned.get(tid)
        if buf is None or buf.numel() < total:
            if buf is not None:
                _PINNED.give(buf)
            buf = self.pinned[tid] = _PINNED.take(total)
        offs = np.concatenate([[0], np.cumsum(host_sizes)])
        with torch.cuda.stream(st):
            for i in range(n):
                buf[int(offs[i]):int(offs[i + 1])].copy_(frames[i * slot:i * slot + int(host_sizes[i])], non_blocking=True)
        st.synchronize()
        host = buf.numpy()
        zi, yi = z // cz, y // cy
        for i, (iz, iy, ix) in enumerate(grid):
            path = ds._chunk_path((zi + iz // cz, yi + iy // cy, ix // cx))
            tmp = f"{path}.tmp{_os.getpid()}.{tid % 100000}"
            with open(tmp, "wb") as f:
                f.write(memoryview(host[int(offs[i]):int(offs[i + 1])]))
            _os.replace(tmp, path)

    def _write(self, ds, src, za, zb, ya, yb, z0, y0, ready):
        import threading
        from .. import _trace
        torch = self.torch
        ready.synchronize()   # host-side wait (a stream parked behind a device-side wait costs the running kernels)
        tid = threading.get_ident()
        if tid not in self.streams:
            from ..volume import io_stream
            self.streams[tid] = io_stream(self.dev)
        st = self.streams[tid]
        part = src[za:zb, ya:yb]
        if self._device_frames_ok(ds, z0 + za, y0 + ya, part):
            with _trace.span("writer: frames made on the device + files written", True):
                os.makedirs(ds.path, exist_ok=True)
                self._write_device_frames(ds, part, z0 + za, y0 + ya, st, tid)
            return
        nbytes = part.numel() * part.element_size()
        buf = self.pinned.get(tid)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                _PINNED.give(buf)
            buf = self.pinned[tid] = _PINNED.take(nbytes)
        host = buf[:nbytes].view(part.dtype).view(part.shape)
        with torch.cuda.stream(st):
            host.copy_(part, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st)
        done.synchronize()
        with _trace.span("writer: encode + write", True):
            ds.write_from((slice(z0 + za, z0 + zb), slice(y0 + ya, y0 + yb)), host.numpy().view(np.uint64))

    def submit(self, ds, src, z0, y0):
        """queue `src` (int64 device tensor holding uint64 ids; the producer ran on the current stream) for ds[z0:, y0:]"""
        ready = self.torch.cuda.Event()
        ready.record(self.torch.cuda.current_stream(self.dev))
        # pieces of whole chunks and at most PIECE_BYTES: a layer of blocks, cut along y at chunk rows where it is larger
        cy = int(ds.chunks[1])
        rows = max(cy, PIECE_BYTES // max(1, self.step * src.shape[2] * 8) // cy * cy)
        for za in range(0, src.shape[0], self.step):
            for ya in range(0, src.shape[1], rows):
                self.futures.append(self.pool.submit(self._write, ds, src, za, min(src.shape[0], za + self.step), ya,
                                                     min(src.shape[1], ya + rows), z0, y0, ready))

    def drain(self):
        try:
            for f in self.futures:
                f.result()
        finally:
            self.futures = []

    def close(self):
        self.pool.shutdown()
        for b in self.pinned.values():
            _PINNED.give(b)
        self.pinned = {}


def worker_grid(config):
    """(Rz, Ry) of `run_waterz_pipeline`'s workers for this config: `num_workers` ranks over the block layers and block rows
    of the ROI, as many of them as have blocks (bootstrapper_amd.volume.rank_grid)."""
    from ..volume import rank_grid
    affs = open_ds(config["affs_dataset"])
    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    sl = affs.roi_to_slices(*roi)[-3:]
    total = tuple(s.stop - s.start for s in sl)
    block = tuple(config["block_shape"]) if config.get("block_shape") else tuple(affs.chunks[1:])
    world = int(config.get("num_workers", 1) or 1) if config.get("blockwise", False) else 1
    return rank_grid(world, -(-total[0] // int(block[0])), -(-total[1] // int(block[1])))


def waterz_pipeline(config, device=None, rank=0, world=1, group=None, grid=None, obj_group=None):
    """post/watershed.py:8-203 for one worker of `world` (a box of whole blocks each: `grid` = (Rz, Ry) parts along z and y,
    default slabs of block layers; bootstrapper_amd.volume): blockwise fragments with context, per-block RAG edge scoring,
    global thresholded connected components, LUT, relabel.  Returns the list of datasets written (fragments first)."""
    import torch
    from .. import _trace
    from ..blockwise import check_task_states, TaskState
    from ..volume import SlabSegmenter, slab_layers
    from .blockwise import RagStore
    from .naming import dump_lut_params

    affs = open_ds(config["affs_dataset"])
    if affs.dtype != np.uint8:
        raise NotImplementedError("the device path takes uint8 affinities (what `bs predict` stores)")
    thresholds = config.get("thresholds", [0.2, 0.35, 0.5])
    merge_function = config.get("merge_function", "mean")
    if merge_function != "mean":   # post/blockwise/waterz_agglom.py:23-36: the blockwise task knows the mean scorer only
        raise NotImplementedError(f"merge_function {merge_function!r}: the blockwise pipeline scores edges by their mean affinity "
                                  "(reference post/blockwise/waterz_agglom.py:23-36); the histogram-quantile scorers belong to the "
                                  "non-blockwise `simple_watershed` (blockwise = false)")
    blockwise = config.get("blockwise", False)
    # choices of waterz / funlib.segment that this repository cannot check (both absent): config keys, so that
    # whoever runs tools/gen_goldens_waterz.py and finds the other answer switches without touching code
    tie_order = config.get("queue_tie_order", "edge_key")
    if tie_order != "edge_key":
        raise NotImplementedError(f"queue_tie_order {tie_order!r}: equal scores leave the exact queue in the order of the edges' initial "
                                  "(smaller id, larger id) key and an N-bin queue first in, first out -- the one specified order "
                                  "(oracle/seg_ref.c); tests/test_waterz_pin.py says whether waterz agrees once its vectors exist")
    frag_params = {
        "fragments_in_xy": config.get("fragments_in_xy", True),
        "min_seed_distance": config.get("min_seed_distance", 10),
        "seed_eps": config.get("seed_eps"),
        "epsilon_agglomerate": config.get("epsilon_agglomerate", 0.0),
        "sigma": config.get("sigma"),
        "noise_eps": config.get("noise_eps"),
        "bias": config.get("bias"),
        "filter_fragments": config.get("filter_fragments", 0.0),
        "remove_debris": config.get("remove_debris", 0),
    }
    voxel_size = affs.voxel_size
    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    sl = affs.roi_to_slices(*roi)[-3:]
    origin = tuple(s.start for s in sl)
    total_shape = tuple(s.stop - s.start for s in sl)
    if blockwise:
        block_size = tuple(config["block_shape"]) if config.get("block_shape") else tuple(affs.chunks[1:])
        ctx = tuple(config["context"]) if config.get("context") else tuple(max(1, b // 8) for b in block_size)
    else:
        block_size, ctx = total_shape, (0, 0, 0)
    block_size = tuple(int(b) for b in block_size)   # not clipped to the ROI: ids are block id * voxels of a whole block
    if rank == 0:
        _warn_3d_fragments(frag_params["fragments_in_xy"], [min(b, t) + 2 * c for b, t, c in zip(block_size, total_shape, ctx)])

    device = rank % max(1, torch.cuda.device_count()) if device is None else device
    layers, rows = -(-total_shape[0] // block_size[0]), -(-total_shape[1] // block_size[1])
    grid = (world, 1) if grid is None else tuple(grid)
    rz, ry = divmod(rank, grid[1])
    zs, zc = slab_layers(layers, grid[0])
    ys, yc = slab_layers(rows, grid[1])
    z0, z1 = zs[rz] * block_size[0], min(total_shape[0], (zs[rz] + zc[rz]) * block_size[0])
    y0, y1 = ys[ry] * block_size[1], min(total_shape[1], (ys[ry] + yc[ry]) * block_size[1])
    if z1 <= z0 or y1 <= y0:
        raise ValueError(f"a {grid[0]} x {grid[1]} grid of workers for {layers} layer(s) x {rows} row(s) of blocks leaves rank {rank} without blocks "
                         "(run_waterz_pipeline sizes the grid with bootstrapper_amd.volume.rank_grid)")
    mask = open_ds(config["mask_dataset"]) if config.get("mask_dataset") else None
    # A box that does not fit the device (43 B per voxel: 2048^3 is 369 GB) is taken in passes of block layers that do
    # (`_waterz_streamed`; `hbm_budget_gb` in the config bounds the device memory a pass may plan for -- default: what is free).
    nthr = len(thresholds)
    n_yx = -(-(y1 - y0) // block_size[1]) * -(-total_shape[2] // block_size[2])
    def slab_bytes(nl):
        return SlabSegmenter.hbm_bytes((min(nl * block_size[0], z1 - z0), y1 - y0, total_shape[2]), ctx, nthr, nl * n_yx,
                                       int(config.get("label_cap", 1 << 16)), int(config.get("edge_cap", 1 << 17)))
    free_b, _tot = torch.cuda.mem_get_info(torch.device("cuda", device))
    free_b += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    limit = min(free_b, int(float(config["hbm_budget_gb"]) * 2**30)) if config.get("hbm_budget_gb") is not None else free_b
    my_layers = -(-(z1 - z0) // block_size[0])
    if blockwise and slab_bytes(my_layers) > limit:
        if world > 1:
            raise MemoryError(f"rank {rank}'s box of {my_layers} block layer(s) needs {slab_bytes(my_layers) / 2**30:.1f} GiB of HBM, {limit / 2**30:.1f} GiB may be "
                              "used: passes over a box that does not fit are implemented for one worker per volume (num_workers = 1), or use more workers / GPUs")
        per_pass = max([nl for nl in range(1, my_layers) if slab_bytes(nl + 1) <= limit] or [0])
        if per_pass < 1:
            raise MemoryError(f"not even two layers of blocks ({slab_bytes(2) / 2**30:.1f} GiB) fit the {limit / 2**30:.1f} GiB of HBM that may be used: smaller blocks, or a larger hbm_budget_gb")
        return _waterz_streamed(config, affs, mask, device, thresholds, merge_function, blockwise, frag_params, roi, voxel_size, origin,
                                total_shape, block_size, ctx, layers, per_pass)
    t_slab = _trace.span("segment: slab + lane workspaces allocated")
    t_slab.__enter__()
    seg = SlabSegmenter((z1 - z0, y1 - y0, total_shape[2]), block_size, ctx, layers, zs[rz], thresholds, frag_params["fragments_in_xy"],
                        frag_params["min_seed_distance"], frag_params["filter_fragments"], frag_params["remove_debris"], 256,
                        n_lanes=int(config.get("lanes", 20)), device=device, rank=rank, world=world, group=group, exchange_affs=False,
                        label_cap=int(config.get("label_cap", 1 << 16)), edge_cap=int(config.get("edge_cap", 1 << 17)),
                        grid=grid, total_rows=rows, row0=ys[ry], obj_group=obj_group,
                        cc_inclusive=bool(config.get("cc_inclusive", True)), queue_bins_formula=config.get("queue_bins_formula", "n_minus_1"),
                        epsilon_agglomerate=frag_params["epsilon_agglomerate"], sigma=frag_params["sigma"],
                        noise_eps=frag_params["noise_eps"], bias=frag_params["bias"], seed_eps=frag_params["seed_eps"], lazy_outputs=True)
    # the stitch's buffers: the interior copy of the fragments and ONE segmentation (written out threshold by threshold: 8.6 GB
    # each for a 1024^3 volume, where all thresholds at once were 26 GB).  Allocated here, not beside the read: the runtime
    # serialises an allocation with the reader's host-to-device copies (measured: the read 0.55 -> 1.5 s)
    seg.ensure_outputs(one=True)
    t_slab.__exit__(None, None, None)
    # The lanes' workspaces, the slab-sized reductions and the relabel kernels are touched for the first time while the
    # affinities stream in (SlabSegmenter.prime: block 0's two tasks on every lane, a collect and a stitch, on whatever the slab
    # holds at that moment -- results discarded; first touch of 40 GB of fresh allocations was half a second of the block stage)
    import threading

    def warm():
        torch.cuda.set_device(seg.dev)
        with _trace.span("segment: lanes warmed beside the read"):
            try:
                seg.prime_lanes()
            except Exception:  # noqa: BLE001 - a warm-up on half-read data: an overflow there means nothing
                pass
            for lane in seg.lanes:   # ... and its sticky overflow flags are read (= cleared) before the real run
                try:
                    lane["engine"].status()
                except Exception:  # noqa: BLE001
                    pass
    warm_thread = threading.Thread(target=warm, name="bsmi-warm") if world == 1 and os.environ.get("BSMI_SEG_WARM", "1") != "0" else None
    if warm_thread is not None:
        warm_thread.start()
    with _trace.span("segment: affinities read into the slab"):
        _fill_affinities(seg, affs, origin, z0, mask, y0)
    if warm_thread is not None:
        warm_thread.join()
        seg.frags.zero_()
        seg.nums.zero_()
        seg.counts_dev.zero_()
    og = obj_group if obj_group is not None else group   # pickled objects: never through an RCCL group

    # fragments + edge scores of this worker's blocks (post/watershed.py:118-153), accounted like daisy tasks
    with _trace.span("segment: fragments + edge scores of the blocks"):
        states = seg.run_blocks_accounted()
    if world > 1:
        import torch.distributed as dist
        mine = {k: v.as_tuple() for k, v in states.items()}
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=og)
        states = {k: TaskState(k) for k in mine}
        for p in parts:
            for k, t in p.items():
                states[k].merge(TaskState.from_tuple(k, t))
    check_task_states(states)

    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    common = dict(offset=roi[0], voxel_size=voxel_size, axis_names=affs.axis_names[1:], units=affs.units, dtype=np.uint64,
                  chunk_shape=tuple(min(b, t) for b, t in zip(block_size, total_shape)))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier(group=group)
    if rank == 0:
        prepare_ds(frags_name, shape=total_shape, **common)
        dump_params(frags_name, {"method": "ws", "blockwise": blockwise, **frag_params})
    barrier()
    # write-behind: fragments now, the segmentations as the stitch produces them; layers of blocks stream out on a pool while
    # the RAG export and the stitch run (the reference's workers write block by block: watershed_frags.py:222-246)
    writer = _LayerWriter(seg.dev, block_size[0])
    writer.submit(open_ds(frags_name, "r+"), seg.interior(seg.frags), z0, y0)

    # RAG to the database (rank 0 gathers nodes and all edges, scored or not: post/watershed.py:100-117 db config)
    import concurrent.futures as cf
    aux = cf.ThreadPoolExecutor(max_workers=4, thread_name_prefix="bsmi-aux")   # the database and the LUT files, beside the stitch
    aux_jobs = []
    ids, pos, size = seg.node_table()
    pos = np.asarray(roi[0], np.float64) + (pos + np.array([z0, y0, 0], np.float64)) * np.asarray(voxel_size, np.float64)
    mine = (ids, pos, size, seg.rag_edges, seg.rag_scores)
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0, group=og)
    else:
        parts = [mine]
    db = config.get("db") or {}
    if rank == 0 and "db_file" in db:
        def export():
            with _trace.span("segment: RAG -> SQLite (beside the stitch)"):
                rag = RagStore()
                for p in parts:
                    rag.add_nodes(p[0], p[1], p[2])
                    rag.add_edges(p[3], p[4])
                rag.to_sqlite(db["db_file"])
        aux_jobs.append(aux.submit(export))

    # global segmentation: thresholded connected components -> LUT -> relabel (post/watershed.py:155-203)
    written = [frags_name]
    lut_dir = config["lut_dir"]

    def emit(t, seg_t):
        """threshold t's segmentation (in the segmenter's one buffer): LUT file, dataset, chunks written before the buffer is reused"""
        threshold = thresholds[t]
        params = {"merge_function": merge_function, "threshold": threshold, **frag_params}
        name = build_name(params)
        recorded = {"method": "ws", "blockwise": blockwise, **params}
        seg_name = os.path.join(config["seg_dataset_prefix"], name)
        if rank == 0:
            os.makedirs(lut_dir, exist_ok=True)
            lut_path = os.path.join(lut_dir, name)
            from .naming import save_npz
            aux_jobs.append(aux.submit(save_npz, lut_path + ".npz", fragment_segment_lut=np.array([seg.nodes, seg.luts[t]])))
            dump_lut_params(lut_path, recorded)
            prepare_ds(seg_name, shape=total_shape, **common)
            dump_params(seg_name, recorded)
        barrier()
        writer.submit(open_ds(seg_name, "r+"), seg_t, z0, y0)
        writer.drain()
        written.append(seg_name)
    with _trace.span("segment: stitch (connected components, LUT), relabel and write threshold by threshold"):
        has_nodes = sum(int(n) for n in seg.block_nums) > 0 or world > 1
        if has_nodes:
            seg.stitch(consume=emit)
    if not has_nodes or seg.nodes.size == 0:
        writer.drain()
        writer.close()
        for j in aux_jobs:
            j.result()
        aux.shutdown()
        return written[:1]
    with _trace.span("segment: wait for the dataset writers"):
        writer.drain()
    writer.close()
    with _trace.span("segment: wait for the database / LUT files"):
        for j in aux_jobs:
            j.result()
    aux.shutdown()
    _trace.report("segment: ")
    barrier()
    return written


def _waterz_streamed(config, affs, mask, device, thresholds, merge_function, blockwise, frag_params, roi, voxel_size, origin,
                     total_shape, block_size, ctx, layers, per_pass):
    """waterz_pipeline for a volume larger than the device: `per_pass` block layers at a time (post/watershed.py:75-153 streams
    any volume block by block through its workers; here a pass is as many layers as fit).  A pass holds one MORE layer than it
    finishes: the layer above, whose fragments the scoring of the pass's top layer reads in its context margin (a block's
    fragments depend on its own read box only, so the next pass computes that layer again, bit for bit the same, and finishes
    it); the margin BELOW comes from the previous pass's top `context` sections, kept on the device.  Fragments go to their
    dataset pass by pass, nodes and scored edges of the finished layers collect on the host, one global connected components,
    then the fragments come back from the store pass by pass for the relabelled copies.  Same datasets, LUTs and database as
    the resident form, bit for bit (tests/test_drivers_gpu.py)."""
    import torch
    from .. import _trace
    from ..blockwise import check_task_states, TaskState
    from ..volume import SlabSegmenter, stitch_components
    from .blockwise import RagStore
    from .engine import lut_relabel_multi
    from .naming import dump_lut_params
    dev = torch.device("cuda", device)
    bz = block_size[0]
    rows, cols = -(-total_shape[1] // block_size[1]), -(-total_shape[2] // block_size[2])
    nvb = int(np.prod(block_size))
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    common = dict(offset=roi[0], voxel_size=voxel_size, axis_names=affs.axis_names[1:], units=affs.units, dtype=np.uint64,
                  chunk_shape=tuple(min(b, t) for b, t in zip(block_size, total_shape)))
    prepare_ds(frags_name, shape=total_shape, **common)
    dump_params(frags_name, {"method": "ws", "blockwise": blockwise, **frag_params})
    frag_ds = open_ds(frags_name, "r+")
    writer = _LayerWriter(dev, bz)
    states = {n: TaskState(n) for n in ("WatershedFrags", "WaterzAgglom")}
    all_ids, all_pos, all_size, all_edges, all_scores = [], [], [], [], []
    carry = None
    passes = [(a, min(layers, a + per_pass)) for a in range(0, layers, per_pass)]
    for a, b in passes:
        top = 1 if b < layers else 0
        z0, z1 = a * bz, min(total_shape[0], (b + top) * bz)
        with _trace.span(f"segment (streamed): layers {a}..{b - 1} of {layers}"):
            seg = SlabSegmenter((z1 - z0, total_shape[1], total_shape[2]), block_size, ctx, layers, a, thresholds, frag_params["fragments_in_xy"],
                                frag_params["min_seed_distance"], frag_params["filter_fragments"], frag_params["remove_debris"], 256,
                                n_lanes=int(config.get("lanes", 20)), device=device, exchange_affs=False,
                                label_cap=int(config.get("label_cap", 1 << 16)), edge_cap=int(config.get("edge_cap", 1 << 17)),
                                total_rows=rows, epsilon_agglomerate=frag_params["epsilon_agglomerate"], sigma=frag_params["sigma"],
                                noise_eps=frag_params["noise_eps"], bias=frag_params["bias"], seed_eps=frag_params["seed_eps"],
                                cc_inclusive=bool(config.get("cc_inclusive", True)), queue_bins_formula=config.get("queue_bins_formula", "n_minus_1"),
                                lazy_outputs=True)
            _fill_affinities(seg, affs, origin, z0, mask, 0)
            if carry is not None:
                seg.frags[:ctx[0]].copy_(carry)    # the margin below: the previous pass's last `context` sections of fragments
                torch.cuda.current_stream(dev).synchronize()   # (the lanes read it from their own streams)
            st = seg.run_blocks_accounted()
            for k, v in st.items():
                # the layer above is counted by the pass that finishes it
                done = TaskState(k, (b - a) * rows * cols)
                done.completed_count = min(v.completed_count, done.total_block_count) if not v.failed_count else v.completed_count
                done.failed_count, done.failed_blocks, done.orphaned_count = v.failed_count, list(v.failed_blocks), v.orphaned_count
                states[k].merge(done)
            hi_id = np.uint64(b * rows * cols * nvb)    # ids of blocks in layers >= b belong to the next pass
            ids, pos, size = seg.node_table()
            keep = ids <= hi_id
            pos = np.asarray(roi[0], np.float64) + (pos + np.array([z0, 0, 0], np.float64)) * np.asarray(voxel_size, np.float64)
            all_ids.append(ids[keep]); all_pos.append(pos[keep]); all_size.append(size[keep])
            own = seg.rag_edges[:, 0] <= hi_id if len(seg.rag_edges) else np.zeros(0, bool)
            all_edges.append(seg.rag_edges[own]); all_scores.append(seg.rag_scores[own])
            nz = min(total_shape[0], b * bz) - z0
            inner = seg.interior(seg.frags)
            writer.submit(frag_ds, inner[:nz], z0, 0)
            if top:
                carry = seg.frags[nz:nz + ctx[0]].clone()
            writer.drain()      # the slab goes away with the pass
            del seg, inner
            torch.cuda.empty_cache()
    check_task_states(states)
    nodes = np.concatenate(all_ids) if all_ids else np.zeros(0, np.uint64)
    edges = np.concatenate(all_edges) if all_edges else np.zeros((0, 2), np.uint64)
    scores = np.concatenate(all_scores) if all_scores else np.zeros(0, np.float32)
    db = config.get("db") or {}
    if "db_file" in db:
        rag = RagStore()
        rag.add_nodes(nodes, np.concatenate(all_pos) if all_pos else np.zeros((0, 3)), np.concatenate(all_size) if all_size else np.zeros(0, np.int64))
        rag.add_edges(edges, scores)
        rag.to_sqlite(db["db_file"])
    written = [frags_name]
    if nodes.size == 0:
        writer.close()
        return written
    thr = thresholds if bool(config.get("cc_inclusive", True)) else [float(np.nextafter(np.float32(t), np.float32(-np.inf))) for t in thresholds]
    luts = stitch_components(nodes, edges, scores, thr)
    lut_dir = config["lut_dir"]
    seg_ds = []
    for t, threshold in enumerate(thresholds):
        params = {"merge_function": merge_function, "threshold": threshold, **frag_params}
        name = build_name(params)
        recorded = {"method": "ws", "blockwise": blockwise, **params}
        seg_name = os.path.join(config["seg_dataset_prefix"], name)
        os.makedirs(lut_dir, exist_ok=True)
        from .naming import save_npz
        save_npz(os.path.join(lut_dir, name) + ".npz", fragment_segment_lut=np.array([nodes, luts[t]]))
        dump_lut_params(os.path.join(lut_dir, name), recorded)
        prepare_ds(seg_name, shape=total_shape, **common)
        dump_params(seg_name, recorded)
        seg_ds.append(open_ds(seg_name, "r+"))
        written.append(seg_name)
    # relabel: the fragments come back from the store, as many layers at a time as a pass held
    keys = torch.from_numpy(nodes.view(np.int64)).to(dev)
    vals = torch.from_numpy(np.stack([c.view(np.int64) for c in luts])).to(dev)
    rd = open_ds(frags_name)
    step = per_pass * bz
    for z0 in range(0, total_shape[0], step):
        z1 = min(total_shape[0], z0 + step)
        with _trace.span(f"segment (streamed): relabel sections {z0}..{z1 - 1}"):
            fr = torch.from_numpy(rd[z0:z1].view(np.int64)).to(dev)
            out = lut_relabel_multi(fr, keys, vals)
            for t, ds in enumerate(seg_ds):
                writer.submit(ds, out[t], z0, 0)
            writer.drain()
            del fr, out
    writer.close()
    return written


def _waterz_worker(rank, world, config, port, results, grid):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # workers that share a GPU (more workers than cards) cannot form an RCCL group: gloo, device tensors staged through the host
    backend = config.get("backend") or ("nccl" if world <= torch.cuda.device_count() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    # the object collectives (task states, nodes and edges to rank 0, LUTs back) go through a gloo group beside an RCCL one
    obj_group = dist.new_group(backend="gloo") if backend == "nccl" else None
    try:
        results[rank] = waterz_pipeline(config, rank=rank, world=world, grid=grid, obj_group=obj_group)
    finally:
        dist.destroy_process_group()


def run_waterz_pipeline(config):
    """`num_workers` workers (post/watershed.py:56: only when blockwise), each on GPU rank % (number of GPUs), as a grid over
    the block layers and block rows of the ROI; workers that would be left without blocks are not started (the reference's
    daisy server hands blocks to any number of workers)."""
    grid = worker_grid(config) if config.get("blockwise", False) else (1, 1)
    world = grid[0] * grid[1]
    if world <= 1:
        return waterz_pipeline(config)
    import torch.multiprocessing as mp
    port = 29800 + os.getpid() % 1000
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_waterz_worker, args=(world, dict(config), port, results, grid), nprocs=world, join=True)
        return list(results[0])


def watershed_segmentation(config):
    """post/watershed.py:357-366"""
    if config.get("blockwise", False):
        if config.get("block_shape") == "roi":
            config["blockwise"] = False
        return run_waterz_pipeline(config)
    return simple_watershed(config)
