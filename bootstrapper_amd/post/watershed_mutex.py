"""`bs segment --mws` on the device: mutex-watershed segmentation.

Behavioural mirror of /root/reference/bootstrapper/post/watershed_mutex.py:8-303:
  * `simple_mutex` (:177-292): whole ROI in one piece -- fragments `<fragments_dataset>/<build_name(frag_params)>`,
    debris-free segmentation `<seg_dataset_prefix>/<build_name(seg_params)>`;
  * `volara_pipeline` (:8-174): per block (with context) mutex-watershed fragments, fragment-pair affinities over the
    neighbourhood, ONE mutex watershed of the fragment graph with `global_bias`, LUT, relabel;
  * `mutex_watershed_segmentation` (:295-303) picks between them.
The clustering runs in libbsmi (`bsmi_mws_agglom_f64`, `bsmi_mws_cluster`, `bsmi_frag_pair_affinity_u8`).  The reference hands
the blockwise stages to volara tasks (`ExtractFrags`, `AffAgglom`, `GraphMWS`, `Relabel`) -- third-party code that is
absent here, as is mwatershed: what the stages compute is restated from their call sites and parameter names (parity
unpinned); documented choices:
  * block id = z-major index of the block in the block grid; fragment ids = block id * voxels per block + 1..n;
  * fragments whose mean affinity over the first three channels is below `filter_fragments` are dropped, then
    components of fewer than `remove_debris` voxels (the clean-up of post/blockwise/watershed_frags.py:148-192, which
    the ExtractFrags parameters of the same names configure);
  * an edge of the fragment graph is written by the block that created its smaller-id fragment; its `zyx_aff` is the
    mean affinity (in [0, 1]) of all voxel pairs (p, p + offset_k) that join the two fragments, over all k;
  * GraphMWS scores an edge `global_bias[0] * zyx_aff + global_bias[1]` and clusters the graph once.
"""
import os

import numpy as np

from ..zarr_io import open_ds, prepare_ds
from .naming import build_name, dump_lut_params, dump_params


def _load_affs(affs, roi, mask_ds, dev):
    """float64 [K][...] in [0, 1] on the device (watershed_mutex.py:232-249)."""
    import torch
    data = affs[affs.roi_to_slices(*roi)]
    a = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    a = a.to(torch.float64) / 255.0 if data.dtype == np.uint8 else a.to(torch.float64)
    if mask_ds:
        mask = open_ds(mask_ds)
        a = a * torch.from_numpy((mask[mask.roi_to_slices(*roi)] > 0).astype(np.uint8)).to(dev)
    return a


def remove_small_objects(labels, min_size):
    """skimage.morphology.remove_small_objects on a label volume (watershed_mutex.py:270-274): labels whose voxel count is
    below min_size become 0.  int64 device tensor in, new tensor out."""
    import torch
    ids, inverse, counts = torch.unique(labels, return_inverse=True, return_counts=True)
    keep = (counts >= int(min_size)) | (ids == 0)
    return torch.where(keep[inverse], labels, torch.zeros_like(labels))


def simple_mutex(config, device=0):
    import torch
    from .mws import mwatershed_from_affinities

    affs = open_ds(config["affs_dataset"])
    neighborhood, bias = config.get("aff_neighborhood"), config.get("bias")
    sigma, noise_eps = config.get("sigma"), config.get("noise_eps")
    strides, randomized_strides = config.get("strides"), config.get("randomized_strides", False)
    remove_debris = config.get("remove_debris", 0)
    if neighborhood is None:
        raise ValueError("Affinities neighborrhood must be provided")
    if bias is None:
        raise ValueError("Affinities bias must be provided")
    assert len(neighborhood) == affs.shape[0], "Number of offsets must match number of affinities channels"
    assert len(neighborhood) == len(bias), "Numbes of biases must match number of affinities channels"
    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    dev = torch.device("cuda", device)
    a = _load_affs(affs, roi, config.get("mask_dataset"), dev)
    frags = mwatershed_from_affinities(a, neighborhood, bias, sigma, noise_eps, strides, randomized_strides, seed=config.get("seed"))

    frag_params = {"sigma": sigma, "noise_eps": noise_eps, "bias": bias, "strides": strides, "randomized_strides": randomized_strides}
    common = dict(offset=roi[0], voxel_size=affs.voxel_size, axis_names=affs.axis_names[1:], units=affs.units, dtype=np.uint64)
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    out = prepare_ds(frags_name, shape=tuple(frags.shape), **common)
    out[:] = frags.cpu().numpy().view(np.uint64)
    dump_params(frags_name, {"method": "mws", "blockwise": False, **frag_params})

    seg = remove_small_objects(frags, remove_debris) if remove_debris > 0 else frags
    seg_params = {**frag_params, "remove_debris": remove_debris}
    seg_name = os.path.join(config["seg_dataset_prefix"], build_name(seg_params))
    out = prepare_ds(seg_name, shape=tuple(seg.shape), **common)
    out[:] = seg.cpu().numpy().view(np.uint64)
    dump_params(seg_name, {"method": "mws", "blockwise": False, **seg_params})
    return [frags_name, seg_name]


def pair_affinities(affs_u8, offsets, frags, ids=None):
    """Fragment-pair affinities of one block on the device (volara AffAgglom): affs_u8 uint8 [K][D][H][W], frags int64
    [D][H][W] of uint64 ids -> (edges uint64 [m][2] of fragment ids, mean affinity float64 [m] in [0, 1], counts)."""
    import ctypes as C
    import torch
    from .. import _lib
    from .engine import lut_relabel
    if ids is None:
        ids = torch.unique(frags)
        ids = ids[ids != 0]
    n = int(ids.numel())
    if n == 0:
        return np.zeros((0, 2), np.uint64), np.zeros(0), np.zeros(0, np.uint64)
    dense = lut_relabel(frags, ids, torch.arange(1, n + 1, dtype=torch.int64, device=frags.device))
    a = affs_u8.contiguous()
    offs = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 3))
    cap = max(1024, 64 * n)
    npairs = C.c_uint64(0)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    while True:
        pairs = np.zeros((cap, 2), np.uint64)
        sums = np.zeros(cap, np.uint64)
        counts = np.zeros(cap, np.uint64)
        rc = _lib.lib.bsmi_frag_pair_affinity_u8(a.device.index or 0, a.data_ptr(), a.shape[0], offs.ctypes.data, dense.data_ptr(),
                                                 _lib.i64x3(frags.shape), cap, pairs.ctypes.data, sums.ctypes.data, counts.ctypes.data,
                                                 C.byref(npairs), stream)
        if rc != 0 and npairs.value > cap:
            cap = int(npairs.value)
            continue
        _lib.check(rc)
        break
    m = int(npairs.value)
    idh = ids.cpu().numpy().view(np.uint64)
    e = idh[pairs[:m].astype(np.int64) - 1]
    return e, sums[:m].astype(np.float64) / counts[:m].astype(np.float64) / 255.0, counts[:m]


def volara_pipeline(config, device=0):
    import torch
    from .blockwise import RagStore, read_with_fill, shrink_blocks
    from .engine import SegEngine, lut_relabel
    from .mws import mws_agglom, mws_cluster, shifted

    affs = open_ds(config["affs_dataset"])
    neighborhood, bias = config.get("aff_neighborhood"), config.get("bias")
    global_bias = tuple(config.get("global_bias", [1.0, -0.5]))
    filter_fragments = config.get("filter_fragments") or 0.0
    sigma, noise_eps = config.get("sigma"), config.get("noise_eps")
    strides, randomized_strides = config.get("strides"), config.get("randomized_strides", False)
    remove_debris = config.get("remove_debris", 0) or 0
    min_seed_distance = config.get("min_seed_distance")
    blockwise = config.get("blockwise", False)
    if neighborhood is None:
        raise ValueError("Affinities neighborhood must be provided")
    if bias is None:
        raise ValueError("Affinities bias must be provided")
    assert len(neighborhood) == len(bias), "Number of biases must match number of affinities channels"
    if affs.dtype != np.uint8:
        raise NotImplementedError("the blockwise device path takes uint8 affinities (what `bs predict` stores)")
    K = len(neighborhood)
    if affs.shape[0] < K:
        raise ValueError(f"{K} offsets for {affs.shape[0]} affinity channels")

    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    vs = [int(v) for v in affs.voxel_size]
    sl = affs.roi_to_slices(*roi)
    origin = tuple(s.start for s in sl[-3:])
    total = tuple(s.stop - s.start for s in sl[-3:])
    if blockwise:   # watershed_mutex.py:73-83 (voxels, as the watershed driver of this package takes them)
        block = [int(b) for b in config["block_shape"]] if config.get("block_shape") else list(affs.chunks[1:])
        ctx = [int(c) for c in config["context"]] if config.get("context") else [max(1, s // 8) for s in block]
    else:
        block, ctx = list(total), [0, 0, 0]
    block = [min(b, t) for b, t in zip(block, total)]

    frag_params = {"min_seed_distance": min_seed_distance, "sigma": sigma, "noise_eps": noise_eps, "bias": bias, "strides": strides,
                   "randomized_strides": randomized_strides, "filter_fragments": config.get("filter_fragments"), "remove_debris": remove_debris}
    seg_params = {"global_bias": list(global_bias), **frag_params}
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    lut_dir = config["lut_dir"]
    lut_name = os.path.join(lut_dir, build_name(seg_params))
    seg_name = os.path.join(config["seg_dataset_prefix"], build_name(seg_params))
    common = dict(offset=roi[0], voxel_size=affs.voxel_size, axis_names=affs.axis_names[1:], units=affs.units, dtype=np.uint64)
    frags_ds = prepare_ds(frags_name, shape=total, chunk_shape=tuple(block), **common)
    seg_ds = prepare_ds(seg_name, shape=total, chunk_shape=tuple(block), **common)
    mask = open_ds(config["mask_dataset"]) if config.get("mask_dataset") else None

    dev = torch.device("cuda", device)
    blocks = shrink_blocks(total, block)
    nvox_block = int(np.prod(block))
    read_shape = tuple(b + 2 * c for b, c in zip(block, ctx))
    eng = SegEngine(read_shape, device)
    rag = RagStore()
    frags_vol = np.zeros(total, dtype=np.uint64)

    def read_block(bi):
        wb, we = blocks[bi]
        rb = tuple(b - c + o for b, c, o in zip(wb, ctx, origin))
        re_ = tuple(e + c + o for e, c, o in zip(we, ctx, origin))
        a = read_with_fill(affs, rb, re_, lead=(affs.shape[0],))[:K]
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if mask is not None:
            t = t * torch.from_numpy((read_with_fill(mask, rb, re_) > 0).astype(np.uint8)).to(dev)
        return t

    # ---- ExtractFrags (watershed_mutex.py:124-142)
    for bi, (wb, we) in enumerate(blocks):
        a_u8 = read_block(bi)
        if int(a_u8.max()) == 0:
            continue
        gen = torch.Generator(device=dev).manual_seed(int(config.get("seed", 0)) + bi)
        x = shifted(a_u8.to(torch.float64) / 255.0, bias, sigma, noise_eps, gen)
        frags = mws_agglom(x, neighborhood, strides, randomized_strides, seed=int(config.get("seed", 0)) + bi + 1)
        first3 = a_u8[:3].contiguous() if K >= 3 else torch.cat([torch.zeros_like(a_u8[:1])] * (3 - K) + [a_u8]).contiguous()
        crop_off = tuple(c for c in ctx)
        crop_shape = tuple(e - b for b, e in zip(wb, we))
        if tuple(frags.shape) != read_shape:   # a shrunk block at the upper faces: its own engine-sized views
            sub = SegEngine(tuple(frags.shape), device)
        else:
            sub = eng
        labels, num = sub.postprocess_fragments(first3, frags.contiguous(), float(filter_fragments), int(remove_debris), crop_off, crop_shape,
                                                bi * nvox_block)
        sub.status()
        n = int(num.item())
        frags_vol[tuple(slice(b, e) for b, e in zip(wb, we))] = labels.cpu().numpy().view(np.uint64)
        if n:
            size, sums = sub.label_stats(labels, bi * nvox_block, n)
            size_h = size.cpu().numpy()
            centre = sums.cpu().numpy().astype(np.float64) / size_h[:, None]
            pos = np.asarray(roi[0], np.float64) + (np.asarray(wb, np.float64) + centre) * np.asarray(vs, np.float64)
            rag.add_nodes(np.arange(1, n + 1, dtype=np.uint64) + np.uint64(bi * nvox_block), pos, size_h)
    frags_ds[:] = frags_vol
    dump_params(frags_name, {"method": "mws", "blockwise": blockwise, **frag_params})

    # ---- AffAgglom (watershed_mutex.py:143-153): fragment-pair affinities with context; a pair belongs to the block of its smaller id
    for bi, (wb, we) in enumerate(blocks):
        a_u8 = read_block(bi)
        rb = tuple(b - c for b, c in zip(wb, ctx))
        re_ = tuple(e + c for e, c in zip(we, ctx))
        f = read_with_fill(frags_vol, rb, re_)
        if not f.any():
            continue
        ft = torch.from_numpy(np.ascontiguousarray(f).view(np.int64)).to(dev)
        e, aff, _ = pair_affinities(a_u8, neighborhood, ft)
        own = (e[:, 0] - np.uint64(1)) // np.uint64(nvox_block) == np.uint64(bi) if len(e) else np.zeros(0, bool)
        rag.add_edges(e[own], aff[own].astype(np.float32))

    # ---- GraphMWS (watershed_mutex.py:155-161): one mutex watershed of the fragment graph -> LUT
    ids, _, _ = rag.nodes()
    e, s = rag.all_edges()
    if len(ids):
        idx = np.searchsorted(ids, e.reshape(-1)).reshape(-1, 2) if len(e) else np.zeros((0, 2), np.int64)
        scores = float(global_bias[0]) * s.astype(np.float64) + float(global_bias[1])
        lab = mws_cluster(len(ids), idx.astype(np.uint64), scores)
        segment = ids[lab.astype(np.int64) - 1]   # a cluster is named by its smallest fragment id
    else:
        segment = np.zeros(0, np.uint64)
    os.makedirs(lut_dir, exist_ok=True)
    from .naming import save_npz
    save_npz(lut_name + ".npz", fragment_segment_lut=np.stack([ids, segment]) if len(ids) else np.zeros((2, 0), np.uint64))
    dump_lut_params(lut_name, {"method": "mws", "blockwise": blockwise, **seg_params})

    # ---- Relabel (watershed_mutex.py:163-172)
    keys = torch.from_numpy(ids.view(np.int64)).to(dev)
    vals = torch.from_numpy(segment.view(np.int64)).to(dev)
    for wb, we in blocks:
        sl3 = tuple(slice(b, e) for b, e in zip(wb, we))
        ft = torch.from_numpy(np.ascontiguousarray(frags_vol[sl3]).view(np.int64)).to(dev)
        seg_ds[sl3] = lut_relabel(ft, keys, vals).cpu().numpy().view(np.uint64) if len(ids) else frags_vol[sl3]
    dump_params(seg_name, {"method": "mws", "blockwise": blockwise, **seg_params})
    if "db" in config and config["db"].get("db_file"):
        rag.to_sqlite(config["db"]["db_file"])
    return [frags_name, seg_name]


def mutex_watershed_segmentation(config):
    """watershed_mutex.py:295-303."""
    if config.get("blockwise", False):
        if config.get("block_shape") == "roi":
            config["blockwise"] = False
        return volara_pipeline(config)
    return simple_mutex(config)
