"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the
reference 3-D U-Net forward pass on PyTorch CPU operators.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The shipped path (bootstrapper_amd/) never does.

Pinned against golden vectors produced by importing the reference model code
(tools/gen_goldens_unet.py -> tests/golden/unet_*.npz); see
tests/test_oracle_unet.py.

Reference being restated (paths relative to /root/reference/bootstrapper):
  models/3d_affs/unet.py:7-76     ConvPass   (conv stack + cropped 1x1x1 residual)
  models/3d_affs/unet.py:79-106   Downsample (MaxPool3d, divisibility check)
  models/3d_affs/unet.py:109-223  Upsample   (trilinear, crop_to_factor, crop, cat)
  models/3d_affs/unet.py:440-478  UNet.rec_forward / forward
  models/3d_affs/model.py:28-64   Model (unet + sigmoid 1x1x1 head)
  models/3d_mtlsd/model.py:28-68  Model (unet + lsds_head + affs_head)
  models/3d_affs/predict.py:145-152  u8 -> [-1,1] normalisation, x255 -> u8 store
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _act(name, t):
    if name is None:
        return t
    if name == "ReLU":
        return torch.relu(t)
    if name == "Sigmoid":
        return torch.sigmoid(t)
    raise ValueError(f"activation {name} not restated")


def _center_crop(t, spatial):
    """unet.py:55-61 / 203-213: centre crop of the trailing 3 dims, offset = (a-b)//2."""
    sl = [slice(None)] * (t.dim() - 3)
    for have, want in zip(t.shape[-3:], spatial):
        o = (have - want) // 2
        sl.append(slice(o, o + want))
    return t[tuple(sl)]


def conv_pass(x, sd, prefix, kernel_sizes, activation):
    """unet.py:63-76.  sd keys: {prefix}.conv_pass.{2*i}.weight/bias, {prefix}.residual.0.*"""
    out = x
    n = len(kernel_sizes)
    for i in range(n):
        out = F.conv3d(out, sd[f"{prefix}.conv_pass.{2 * i}.weight"],
                       sd[f"{prefix}.conv_pass.{2 * i}.bias"])
        if activation is not None and i < n - 1:
            out = _act(activation, out)
    res = F.conv3d(x, sd[f"{prefix}.residual.0.weight"], sd[f"{prefix}.residual.0.bias"])
    ret = out + _center_crop(res, out.shape[-3:])
    return _act(activation, ret)


def downsample(x, factor):
    """unet.py:96-106."""
    for d in range(1, 4):
        if x.shape[-d] % factor[-d] != 0:
            raise RuntimeError(
                "Can not downsample shape %s with factor %s, mismatch in spatial dimension %d"
                % (tuple(x.shape), tuple(factor), 3 - d))
    return F.max_pool3d(x, tuple(factor), stride=tuple(factor))


def crop_to_factor_shape(spatial, factor, kernel_sizes):
    """unet.py:147-201: largest s' = n*f + c <= s with c = conv crop of the next pass."""
    conv_crop = tuple(sum(ks[d] - 1 for ks in kernel_sizes) for d in range(3))
    ns = (int(math.floor(float(s - c) / f)) for s, c, f in zip(spatial, conv_crop, factor))
    target = tuple(n * f + c for n, c, f in zip(ns, conv_crop, factor))
    if target != tuple(spatial):
        if not all(t > c for t, c in zip(target, conv_crop)):
            raise AssertionError("Feature map too small for translation equivariance")
    return target


def upsample_cat(f_left, g_out, factor, crop_factor, next_kernel_sizes):
    """unet.py:215-223 with mode='trilinear' (constant_upsample=True, model.py:50)."""
    g_up = F.interpolate(g_out, scale_factor=tuple(float(f) for f in factor), mode="trilinear")
    target = crop_to_factor_shape(g_up.shape[-3:], crop_factor, next_kernel_sizes)
    g_c = _center_crop(g_up, target)
    f_c = _center_crop(f_left, g_c.shape[-3:])
    return torch.cat([f_c, g_c], dim=1)


def crop_factors(downsample_factors):
    """unet.py:353-362."""
    out, prod = [], None
    for f in downsample_factors[::-1]:
        prod = list(f) if prod is None else [a * b for a, b in zip(f, prod)]
        out.append(prod)
    return out[::-1]


def unet_forward(cfg, sd, x):
    """unet.py:440-478 (num_heads == 1)."""
    dfs = cfg["downsample_factors"]
    nl = len(dfs) + 1
    ksd = cfg.get("kernel_size_down") or [[[3, 3, 3], [3, 3, 3]]] * nl
    ksu = cfg.get("kernel_size_up") or [[[3, 3, 3], [3, 3, 3]]] * (nl - 1)
    cfs = crop_factors(dfs)

    def rec(level, f_in):
        i = nl - level - 1
        f_left = conv_pass(f_in, sd, f"unet.l_conv.{i}", ksd[i], "ReLU")
        if level == 0:
            return f_left
        g_in = downsample(f_left, dfs[i])
        g_out = rec(level - 1, g_in)
        f_right = upsample_cat(f_left, g_out, dfs[i], cfs[i], ksu[i])
        return conv_pass(f_right, sd, f"unet.r_conv.0.{i}", ksu[i], "ReLU")

    return rec(nl - 1, x)


def head_names(model_name):
    """Order in which the reference Model.forward returns its heads."""
    return {"3d_affs": ["affs_head"], "3d_lsd": ["lsds_head"],
            "3d_mtlsd": ["lsds_head", "affs_head"]}[model_name]


def model_forward(cfg, sd, x, heads):
    """model.py:58-64.  x: float32 (1, Cin, D, H, W).  Returns list of (Cout, d, h, w)."""
    sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
    with torch.no_grad():
        z = unet_forward(cfg, sd, x)
        return [conv_pass(z, sd, h, [[1, 1, 1]], "Sigmoid")[0] for h in heads]


def normalize_raw(raw_u8):
    """predict.py:147-149: Normalize (u8 * 1/255 as float32) then IntensityScaleShift(2, -1)."""
    x = raw_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    return x * np.float32(2) + np.float32(-1)


def to_u8(pred):
    """predict.py:153-154: IntensityScaleShift(255, 0) then ZarrWrite's cast to the uint8
    dataset (numpy astype: truncation toward zero)."""
    return (pred * np.float32(255)).astype(np.uint8)


def predict_block(cfg, sd, raw_u8, heads):
    """raw_u8: (D,H,W) or (Cin,D,H,W) uint8 -> list of float32 head outputs."""
    x = normalize_raw(raw_u8)
    if x.ndim == 3:
        x = x[None]
    outs = model_forward(cfg, sd, torch.from_numpy(x)[None], heads)
    return [o.numpy() for o in outs]


def default_cfg(num_fmaps=12, inc=5, in_channels=1):
    return dict(in_channels=in_channels, num_fmaps=num_fmaps, fmap_inc_factor=inc,
                downsample_factors=[[1, 2, 2]] * 3,
                kernel_size_down=[[[3, 3, 3], [3, 3, 3]]] * 4,
                kernel_size_up=[[[3, 3, 3], [3, 3, 3]]] * 3)


# ---- the other setups of the model family ------------------------------------------------------
# 2-D setups (models/2d_mtlsd/unet.py: Conv2d, MaxPool2d, mode="bilinear" at :400) are restated as the 3-D
# operators above over a (1, h, w) volume: unit-depth kernels and factors, and trilinear interpolation with
# scale 1 along z has weight 1 on the section itself, i.e. it is the bilinear one.  Second-stage setups
# (models/3d_affs_from_2d_mtlsd/model.py:28-66) use num_fmaps_out for the last right-side ConvPass, which is
# only visible in the tensor shapes of the state dict, and concatenate their inputs along the channels.

FAMILY_HEADS = {"3d_affs": "affs_head", "3d_lsds": "lsds_head", "2d_affs": "aff_head", "2d_lsds": "lsd_head"}


def lift_cfg(net_config):
    """net_config.json dict of any setup -> cfg for unet_forward (3-D kernel sizes and factors)."""
    def lift(k):
        k = [int(v) for v in k]
        return [1] + k if len(k) == 2 else k
    dfs = [lift(f) for f in net_config["downsample_factors"]]
    return dict(downsample_factors=dfs,
                kernel_size_down=[[lift(k) for k in ks] for ks in net_config["kernel_size_down"]],
                kernel_size_up=[[lift(k) for k in ks] for ks in net_config["kernel_size_up"]])


def lift_sd(sd):
    """Conv2d weights (O, I, kh, kw) -> (O, I, 1, kh, kw)."""
    out = {}
    for k, v in sd.items():
        v = torch.from_numpy(v) if isinstance(v, np.ndarray) else v
        out[k] = v[:, :, None] if v.dim() == 4 else v
    return out


def family_forward(net_config, sd, x):
    """x: float32 (1, C, D, H, W) normalised input ((1, C, 1, H, W) for a 2-D setup).
    Returns the head outputs (dims, d, h, w) in the order of net_config["outputs"]."""
    heads = [FAMILY_HEADS[k] for k in net_config["outputs"]]
    return model_forward(lift_cfg(net_config), lift_sd(sd), x, heads)


def normalize_unit(u8):
    """gp.Normalize alone (models/3d_affs_from_2d_mtlsd/predict.py:163-164)."""
    return u8.astype(np.float32) * np.float32(1.0 / 255.0)
