"""`bs segment --ws` driver on the device.

Behavioural mirror of /root/reference/bootstrapper/post/watershed.py:206-366
(`simple_watershed`, `watershed_segmentation`): read the first three affinity channels of the ROI,
optional mask, fragments -> `<fragments_dataset>/<build_name(frag_params)>`, agglomeration at every
threshold -> `<seg_dataset_prefix>/<build_name(params)>`, parameters recorded as `bs_params`.
The arithmetic runs in libbsmi (bootstrapper_amd.post.ws / .waterz); the blockwise RAG pipeline
(`waterz_pipeline`, post/watershed.py:8-203) is not built yet and raises.
"""
import os

import numpy as np

from ..zarr_io import open_ds, prepare_ds
from .naming import build_name, dump_params


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
    if merge_function != "mean":
        raise NotImplementedError(f"merge_function {merge_function!r}: only 'mean' is implemented (the one the reference enables)")
    if any([sigma, noise_eps, bias]):
        raise NotImplementedError("affinity shifts (sigma / noise_eps / bias) are not implemented on the device")
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
    frags, _ = watershed_from_affinities(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance)
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    common = dict(offset=roi[0], voxel_size=affs.voxel_size, axis_names=affs.axis_names[1:], units=affs.units,
                  dtype=np.uint64)
    out = prepare_ds(frags_name, shape=frags.shape, **common)
    out[:] = frags.cpu().numpy().astype(np.uint64)
    dump_params(frags_name, {"method": "ws", "blockwise": False, **frag_params})

    written = [frags_name]
    for threshold, seg in zip(thresholds, agglomerate(a, thresholds, fragments=frags)):
        params = {"merge_function": merge_function, "threshold": threshold, **frag_params}
        seg_name = os.path.join(config["seg_dataset_prefix"], build_name(params))
        out = prepare_ds(seg_name, shape=seg.shape, **common)
        out[:] = seg.cpu().numpy().astype(np.uint64)
        dump_params(seg_name, {"method": "ws", "blockwise": False, **params})
        written.append(seg_name)
    return written


def watershed_segmentation(config):
    if config.get("blockwise", False):
        raise NotImplementedError(
            "blockwise watershed (fragments with context + per-block agglomeration + global RAG stitch, "
            "reference post/watershed.py:8-203) is not built yet; run with blockwise = false")
    return simple_watershed(config)
