"""`bs segment --cc` on the device: thresholded-affinity connected components.

Behavioural mirror of /root/reference/bootstrapper/post/connected_components.py:8-127 (`cc_blockwise`, `cc_affs`,
`cc_segmentation`) with the labelling of post/cc.py:7-74 in libbsmi (`bsmi_cc_affs_u8`): fragments dataset
`<fragments_dataset>/<build_name(frag_params)>`, debris-free segmentation `<seg_dataset_prefix>/<build_name(seg_params)>`,
parameters recorded as `bs_params`.
"""
import os

import numpy as np

from ..zarr_io import open_ds, prepare_ds
from .naming import build_name, dump_params


def cc_blockwise(config):
    raise NotImplementedError("Blockwise connected components not implemented yet")  # same as the reference (:8-9)


def cc_affs(config, device=0):
    import torch
    from .engine import SegEngine
    affs = open_ds(config["affs_dataset"])
    threshold = config.get("threshold", 0.5)
    sigma, noise_eps = config.get("sigma"), config.get("noise_eps")
    remove_debris = config.get("remove_debris", 0)
    if sigma is not None or noise_eps is not None:
        raise NotImplementedError("affinity shifts (sigma / noise_eps) are not implemented on the device")
    if affs.dtype != np.uint8:
        raise NotImplementedError("the device path takes uint8 affinities (what `bs predict` stores)")
    if config.get("roi_offset") is not None:
        roi = (list(config["roi_offset"]), list(config["roi_shape"]))
    else:
        roi = (list(affs.roi[0]), list(affs.roi[1]))
    data = affs[affs.roi_to_slices(*roi)][:3]
    dev = torch.device("cuda", device)
    a = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    if config.get("mask_dataset"):
        mask = open_ds(config["mask_dataset"])
        a = a * torch.from_numpy((mask[mask.roi_to_slices(*roi)] > 0).astype(np.uint8)).to(dev)
    eng = SegEngine(tuple(a.shape[1:]), device)
    frags, seg, _ = eng.cc_affs(a, threshold, remove_debris)
    eng.status()
    frag_params = {"threshold": threshold, "sigma": sigma, "noise_eps": noise_eps}
    common = dict(offset=roi[0], voxel_size=affs.voxel_size, axis_names=affs.axis_names[1:], units=affs.units, dtype=np.uint64)
    frags_name = os.path.join(config["fragments_dataset"], build_name(frag_params))
    out = prepare_ds(frags_name, shape=frags.shape, **common)
    out[:] = frags.cpu().numpy().astype(np.uint64)
    dump_params(frags_name, {"method": "cc", "blockwise": False, **frag_params})
    seg_params = {**frag_params, "remove_debris": remove_debris}
    seg_name = os.path.join(config["seg_dataset_prefix"], build_name(seg_params))
    out = prepare_ds(seg_name, shape=seg.shape, **common)
    out[:] = seg.cpu().numpy().astype(np.uint64)
    dump_params(seg_name, {"method": "cc", "blockwise": False, **seg_params})
    return [frags_name, seg_name]


def cc_segmentation(config):
    if config.get("blockwise", False):
        return cc_blockwise(config)
    return cc_affs(config)
