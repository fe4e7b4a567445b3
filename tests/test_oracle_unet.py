"""Pin oracle/unet_ref.py against golden vectors generated from the reference
model code (tools/gen_goldens_unet.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref as R

TOL = 1e-5  # oracle and reference both run torch CPU fp32; differences are summation order only


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    return d, sd


@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3", "mtlsd_f4i2"])
def test_whole_net_matches_reference(golden_dir, tag):
    d, sd = _load(golden_dir, f"unet_{tag}.npz")
    meta = json.loads(bytes(d["config"]).decode())
    cfg = R.default_cfg(meta["num_fmaps"], meta["fmap_inc_factor"])
    outs = R.predict_block(cfg, sd, d["raw_u8"], R.head_names(meta["model"]))
    for i, o in enumerate(outs):
        ref = d[f"out{i}"]
        assert o.shape == ref.shape
        assert np.abs(o - ref).max() < TOL


def test_operators_match_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "unet_ops.npz"))

    def sd(prefix):
        return {"p." + k[len(prefix) + 3:]: torch.from_numpy(d[k]) for k in d.files
                if k.startswith(prefix + "_w:")}

    with torch.no_grad():
        y = R.conv_pass(torch.from_numpy(d["convpass_x"])[None], sd("convpass"), "p",
                        [(3, 3, 3), (3, 3, 3)], "ReLU")[0].numpy()
        assert np.abs(y - d["convpass_y"]).max() < TOL
        y = R.conv_pass(torch.from_numpy(d["convpass2_x"])[None], sd("convpass2"), "p",
                        [(1, 3, 3), (3, 3, 3)], "ReLU")[0].numpy()
        assert np.abs(y - d["convpass2_y"]).max() < TOL
        y = R.conv_pass(torch.from_numpy(d["head_x"])[None], sd("head"), "p",
                        [[1, 1, 1]], "Sigmoid")[0].numpy()
        assert np.abs(y - d["head_y"]).max() < TOL
        y = R.downsample(torch.from_numpy(d["down_x"])[None], (1, 2, 2))[0].numpy()
        assert np.array_equal(y, d["down_y"])
        y = R.upsample_cat(torch.from_numpy(d["up_f_left"])[None],
                           torch.from_numpy(d["up_g_out"])[None], (1, 2, 2), (1, 4, 4),
                           [(3, 3, 3), (3, 3, 3)])[0].numpy()
        assert y.shape == d["up_y"].shape
        assert np.abs(y - d["up_y"]).max() < TOL


def test_downsample_rejects_indivisible():
    with pytest.raises(RuntimeError):
        R.downsample(torch.zeros(1, 1, 4, 7, 6), (1, 2, 2))


FAMILY = ["2d_mtlsd_f4i2", "2d_lsd_f3i3", "2d_affs_f4i2", "3d_lsd_f4i2", "from_2d_mtlsd_f3i2", "from_3d_lsd_f4i2",
          "from_2d_affs_f4i3"]


def family_case(golden_dir, tag):
    """(net_config, state dict, [u8 inputs], normalised float input (1,C,D,H,W), [reference outputs])"""
    d, sd = _load(golden_dir, f"family_{tag}.npz")
    nc = json.loads(bytes(d["net_config"]).decode())
    ins = [d[k] for k in sorted(k for k in d.files if k.startswith("in") and k[2:].isdigit())]
    if "in_channels" in nc:
        x = R.normalize_raw(ins[0])
        x = x[:, None] if "adj_slices" in nc else x[None]      # 2-D: (C, 1, H, W); 3-D raw: (1, D, H, W)
    else:
        x = np.concatenate([R.normalize_unit(a) for a in ins], axis=0)
    outs = [d[k] for k in sorted(k for k in d.files if k.startswith("out"))]
    return nc, sd, ins, x[None], outs


@pytest.mark.parametrize("tag", FAMILY)
def test_family_matches_reference(golden_dir, tag):
    """2-D setups restated as unit-depth 3-D operators, second-stage setups with num_fmaps_out and several inputs."""
    nc, sd, _, x, refs = family_case(golden_dir, tag)
    outs = R.family_forward(nc, sd, torch.from_numpy(x))
    assert len(outs) == len(refs)
    for o, ref in zip(outs, refs):
        o = o.numpy()
        if ref.ndim == 3:          # 2-D setup: reference output (dims, h, w)
            assert o.shape[1] == 1
            o = o[:, 0]
        assert o.shape == ref.shape
        assert np.abs(o - ref).max() < TOL
