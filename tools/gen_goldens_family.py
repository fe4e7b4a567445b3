#!/usr/bin/env python3
"""Golden vectors for the other setups of the model family (2-D nets, LSD-only net, second-stage
nets), made by importing the REFERENCE model code (see tools/gen_goldens_unet.py for the import
recipe).  Only seeded inputs, the seeded state dict and the reference's outputs are stored.

Usage: python tools/gen_goldens_family.py   (writes tests/golden/family_*.npz)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_goldens_unet import OUT, REF, load_ref, sd_to_np  # noqa: E402


def _perturb(net):
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
            if "head" in n and n.endswith("weight"):
                p.mul_(6.0)
    net.eval()


def one(model_name, tag, num_fmaps, inc, spatial, seed, num_fmaps_out=None):
    model_mod, _ = load_ref(model_name)
    with open(os.path.join(REF, model_name, "net_config.json")) as f:
        nc = json.load(f)
    nc["num_fmaps"], nc["fmap_inc_factor"] = num_fmaps, inc
    if num_fmaps_out is not None:
        nc["num_fmaps_out"] = num_fmaps_out
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    arrs = {}
    if hasattr(model_mod, "AffsUNet"):
        net = model_mod.AffsUNet(num_fmaps=num_fmaps, num_fmaps_out=num_fmaps_out, fmap_inc_factor=inc)
        _perturb(net)
        ins = []
        for i, (name, val) in enumerate(nc["inputs"].items()):
            u8 = rng.integers(0, 256, size=(val["dims"],) + tuple(spatial), dtype=np.uint8)
            arrs[f"in{i}"] = u8
            # reference predict pipeline: gp.Normalize only (predict.py:163-164)
            ins.append(torch.from_numpy(u8.astype(np.float32) * np.float32(1.0 / 255.0))[None])
        with torch.no_grad():
            y = net(*ins)
    else:
        net = model_mod.Model(num_fmaps=num_fmaps, fmap_inc_factor=inc)
        _perturb(net)
        cin = nc["in_channels"] * nc.get("adj_slices", 1)
        shape = ((cin,) if "adj_slices" in nc else ()) + tuple(spatial)
        u8 = rng.integers(0, 256, size=shape, dtype=np.uint8)
        arrs["in0"] = u8
        x = u8.astype(np.float32) * np.float32(1.0 / 255.0)
        x = x * np.float32(2) + np.float32(-1)
        x = torch.from_numpy(x)[None] if "adj_slices" in nc else torch.from_numpy(x)[None, None]
        with torch.no_grad():
            y = net(x)
    ys = y if isinstance(y, (tuple, list)) else (y,)
    for i, t in enumerate(ys):
        arrs[f"out{i}"] = t[0].numpy()
    arrs["net_config"] = np.frombuffer(json.dumps(nc).encode(), dtype=np.uint8)
    arrs.update(sd_to_np(net.state_dict()))
    path = os.path.join(OUT, f"family_{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(path, {k: v.shape for k, v in arrs.items() if not k.startswith("w:")},
          sum(p.numel() for p in net.parameters()), "params")


if __name__ == "__main__":
    one("2d_mtlsd", "2d_mtlsd_f4i2", 4, 2, (108, 116), seed=11)
    one("2d_lsd", "2d_lsd_f3i3", 3, 3, (100, 100), seed=12)
    one("2d_affs", "2d_affs_f4i2", 4, 2, (124, 100), seed=13)
    one("3d_lsd", "3d_lsd_f4i2", 4, 2, (30, 108, 108), seed=14)
    one("3d_affs_from_2d_mtlsd", "from_2d_mtlsd_f3i2", 3, 2, (22, 100, 108), seed=15, num_fmaps_out=5)
    one("3d_affs_from_3d_lsd", "from_3d_lsd_f4i2", 4, 2, (21, 100, 100), seed=16, num_fmaps_out=6)
    one("3d_affs_from_2d_affs", "from_2d_affs_f4i3", 4, 3, (21, 100, 100), seed=17, num_fmaps_out=7)
