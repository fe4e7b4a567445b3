#!/usr/bin/env python3
"""Generate U-Net golden vectors by importing the REFERENCE model code.

Runs only in the build container (needs /root/reference).  The reference
`unet.py` imports `funlib.learn.torch.models.conv4d.Conv4d` at module scope
(reference bootstrapper/models/3d_affs/unet.py:1) but only touches it for
4-D kernels (unet.py:26), which no shipped 3-D config uses; the package is
absent here, so an empty module with `Conv4d = None` is registered before the
import.  Nothing from the reference is copied: only
seeded inputs, the seeded state_dict and the reference's outputs are stored.

Usage: python tools/gen_goldens_unet.py   (writes tests/golden/unet_*.npz)
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/bootstrapper/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub_conv4d():
    for name in ["funlib", "funlib.learn", "funlib.learn.torch",
                 "funlib.learn.torch.models", "funlib.learn.torch.models.conv4d"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["funlib.learn.torch.models.conv4d"].Conv4d = None


def load_ref(model_name):
    _stub_conv4d()
    d = os.path.join(REF, model_name)
    for m in ("model", "unet"):
        sys.modules.pop(m, None)
    sys.path.insert(0, d)
    try:
        model = importlib.import_module("model")
        unet = importlib.import_module("unet")
    finally:
        sys.path.pop(0)
    return model, unet


def sd_to_np(sd):
    return {"w:" + k: v.detach().numpy().copy() for k, v in sd.items()}


def whole_net(model_name, tag, num_fmaps, inc, in_shape, seed, outputs=None):
    model_mod, _ = load_ref(model_name)
    torch.manual_seed(seed)
    kw = dict(num_fmaps=num_fmaps, fmap_inc_factor=inc)
    net = model_mod.Model(**kw)
    # default init gives near-zero logits; scale biases up a bit so that the
    # ReLUs/sigmoids are exercised on both sides.
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
            if "head" in n and n.endswith("weight"):
                p.mul_(6.0)
    net.eval()
    rng = np.random.default_rng(seed)
    raw_u8 = rng.integers(0, 256, size=in_shape, dtype=np.uint8)
    # reference predict pipeline (models/3d_affs/predict.py:145-152):
    # Normalize (u8 -> f32 * 1/255) then IntensityScaleShift(raw, 2, -1)
    x = raw_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    x = x * np.float32(2) + np.float32(-1)
    with torch.no_grad():
        y = net(torch.from_numpy(x)[None, None])
    ys = y if isinstance(y, (tuple, list)) else (y,)
    cfg = dict(model=model_name, num_fmaps=num_fmaps, fmap_inc_factor=inc,
               in_shape=list(in_shape))
    arrs = dict(raw_u8=raw_u8, config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8))
    for i, t in enumerate(ys):
        arrs[f"out{i}"] = t[0].numpy()
    arrs.update(sd_to_np(net.state_dict()))
    path = os.path.join(OUT, f"unet_{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(path, {k: v.shape for k, v in arrs.items() if not k.startswith("w:")},
          sum(p.numel() for p in net.parameters()), "params")


def operators(seed=7):
    _, unet = load_ref("3d_affs")
    torch.manual_seed(seed)
    arrs = {}
    # ConvPass 2x(3,3,3) with residual, ReLU  (unet.py:7-76)
    cp = unet.ConvPass(5, 7, [(3, 3, 3), (3, 3, 3)], "ReLU")
    x = torch.randn(1, 5, 9, 12, 11)
    with torch.no_grad():
        arrs["convpass_x"] = x[0].numpy(); arrs["convpass_y"] = cp(x)[0].numpy()
    for k, v in cp.state_dict().items():
        arrs["convpass_w:" + k] = v.numpy().copy()
    # ConvPass with mixed kernels (1,3,3),(3,3,3) as in 3d_affs_from_* configs
    cp2 = unet.ConvPass(3, 4, [(1, 3, 3), (3, 3, 3)], "ReLU")
    x = torch.randn(1, 3, 6, 10, 9)
    with torch.no_grad():
        arrs["convpass2_x"] = x[0].numpy(); arrs["convpass2_y"] = cp2(x)[0].numpy()
    for k, v in cp2.state_dict().items():
        arrs["convpass2_w:" + k] = v.numpy().copy()
    # head: ConvPass 1x1x1 + Sigmoid (model.py:54-56)
    hd = unet.ConvPass(4, 6, [[1, 1, 1]], "Sigmoid")
    x = torch.randn(1, 4, 3, 5, 6) * 3
    with torch.no_grad():
        arrs["head_x"] = x[0].numpy(); arrs["head_y"] = hd(x)[0].numpy()
    for k, v in hd.state_dict().items():
        arrs["head_w:" + k] = v.numpy().copy()
    # Downsample (unet.py:79-106)
    ds = unet.Downsample((1, 2, 2))
    x = torch.randn(1, 3, 4, 8, 6)
    arrs["down_x"] = x[0].numpy(); arrs["down_y"] = ds(x)[0].numpy()
    # Upsample trilinear + crop_to_factor + crop + cat (unet.py:109-223)
    up = unet.Upsample((1, 2, 2), mode="trilinear", crop_factor=(1, 4, 4),
                       next_conv_kernel_sizes=[(3, 3, 3), (3, 3, 3)])
    f_left = torch.randn(1, 2, 10, 24, 22)
    g_out = torch.randn(1, 3, 8, 9, 7)
    arrs["up_f_left"] = f_left[0].numpy(); arrs["up_g_out"] = g_out[0].numpy()
    arrs["up_y"] = up(f_left, g_out)[0].numpy()
    path = os.path.join(OUT, "unet_ops.npz")
    np.savez_compressed(path, **arrs)
    print(path, {k: v.shape for k, v in arrs.items() if ":" not in k})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    operators()
    # width-reduced 3d_affs: channels 4/8/16/32; smallest valid input
    whole_net("3d_affs", "affs_f4i2", 4, 2, (30, 108, 108), seed=1)
    # odd channel counts (3/9/27/81): exercises channel padding
    whole_net("3d_affs", "affs_f3i3", 3, 3, (30, 108, 116), seed=2)
    # two-headed 3d_mtlsd, returns (lsds, affs)
    whole_net("3d_mtlsd", "mtlsd_f4i2", 4, 2, (31, 108, 108), seed=3)
