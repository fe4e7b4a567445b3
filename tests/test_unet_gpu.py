"""Parity of the HIP U-Net path (through the C ABI) with the oracle and with the golden
vectors generated from the reference model code.  Needs an MI355X."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# north_star: affinities within 1e-4 of the PyTorch CPU path (f32 MFMA mode)
TOL_F32 = 1e-4
# bf16 operands / f32 accumulate: 8-bit mantissa activations through 14+ stacked convs.  Measured on the golden nets of
# this module (the prints below): at most 1.7e-3 (4.5e-3 on the full-size net, tests/test_fullsize_gpu.py); the gate is the
# measured value with a margin of 3.
TOL_BF16 = 5e-3
# split bf16 (hi + lo operands, hi*hi + lo*hi + hi*lo on the bf16 MFMA, f32 accumulate): same gate as f32
TOL_BF16X3 = 1e-4


def _load(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, f"unet_{tag}.npz"))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    meta = json.loads(bytes(d["config"]).decode())
    return d, sd, meta


def _net_config(meta):
    outs = {"3d_affs": {"3d_affs": {"dims": 6}},
            "3d_mtlsd": {"3d_lsds": {"dims": 10}, "3d_affs": {"dims": 6}}}[meta["model"]]
    return {"in_channels": 1, "num_fmaps": meta["num_fmaps"], "fmap_inc_factor": meta["fmap_inc_factor"],
            "downsample_factors": [[1, 2, 2]] * 3, "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
            "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3, "outputs": outs}


@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3", "mtlsd_f4i2"])
@pytest.mark.parametrize("prec,tol", [("f32", TOL_F32), ("bf16x3", TOL_BF16X3), ("bf16", TOL_BF16)])
def test_forward_matches_reference_goldens(golden_dir, tag, prec, tol):
    from bootstrapper_amd.unet import Model
    from oracle import unet_ref as R
    d, sd, meta = _load(golden_dir, tag)
    m = Model(_net_config(meta), precision=prec).load_state_dict(sd)
    x = torch.from_numpy(R.normalize_raw(d["raw_u8"]))[None, None].cuda()
    y = m(x)
    ys = y if isinstance(y, tuple) else (y,)
    for i, t in enumerate(ys):
        got = t[0].cpu().numpy()
        ref = d[f"out{i}"]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        print(f"{tag} {prec} head{i}: max abs err {err:.3e}")
        assert err < tol


@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3"])
def test_u8_pipeline_matches_oracle(golden_dir, tag):
    """u8 in -> u8 out (predict worker arithmetic).  f32 mode; a 1e-4 float difference can
    flip the truncation at an integer boundary, so allow +-1 and require >99% exact."""
    from bootstrapper_amd.unet import Model
    from oracle import unet_ref as R
    d, sd, meta = _load(golden_dir, tag)
    m = Model(_net_config(meta), precision="f32").load_state_dict(sd)
    u8, f32 = m.predict_u8(torch.from_numpy(d["raw_u8"]).cuda(), want_f32=True)
    ref_f = R.predict_block(R.default_cfg(meta["num_fmaps"], meta["fmap_inc_factor"]), sd, d["raw_u8"],
                            R.head_names(meta["model"]))[0]
    ref_u8 = R.to_u8(ref_f)
    got = u8[0].cpu().numpy()
    assert np.abs(f32[0].cpu().numpy() - ref_f).max() < TOL_F32
    diff = np.abs(got.astype(np.int32) - ref_u8.astype(np.int32))
    assert diff.max() <= 1
    assert (diff == 0).mean() > 0.99


def test_random_shapes_against_oracle():
    """Seeded random weights, non-golden shapes (ragged: W != H, odd depth), mixed kernel
    sizes (1,3,3)/(3,3,3) as the 3d_affs_from_* configs use; oracle = torch CPU fp32."""
    from bootstrapper_amd.unet import Model
    from oracle import unet_ref as R
    rng = np.random.default_rng(11)
    cfgs = [
        dict(num_fmaps=5, inc=3, ksd=[[[3, 3, 3], [3, 3, 3]]] * 3, ksu=[[[3, 3, 3], [3, 3, 3]]] * 2,
             dfs=[[1, 2, 2], [1, 2, 2]], shape=(23, 68, 76)),
        dict(num_fmaps=6, inc=2, ksd=[[[1, 3, 3], [3, 3, 3]]] * 3, ksu=[[[1, 3, 3], [3, 3, 3]]] * 2,
             dfs=[[1, 2, 2], [2, 2, 2]], shape=(16, 60, 52)),
        dict(num_fmaps=4, inc=2, ksd=[[[3, 3, 3]]] * 2, ksu=[[[3, 3, 3]]], dfs=[[1, 3, 3]], shape=(9, 35, 38)),
    ]
    for c in cfgs:
        nl = len(c["dfs"]) + 1
        fm = [c["num_fmaps"] * c["inc"] ** l for l in range(nl)]
        sd = {}

        def add_pass(prefix, cin, cout, ks):
            ci = cin
            for i, k in enumerate(ks):
                fan = ci * k[0] * k[1] * k[2]
                sd[f"{prefix}.conv_pass.{2 * i}.weight"] = (rng.standard_normal((cout, ci, *k)) * (1.5 / np.sqrt(fan))).astype(np.float32)
                sd[f"{prefix}.conv_pass.{2 * i}.bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
                ci = cout
            sd[f"{prefix}.residual.0.weight"] = (rng.standard_normal((cout, cin, 1, 1, 1)) / np.sqrt(cin)).astype(np.float32)
            sd[f"{prefix}.residual.0.bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)

        for l in range(nl):
            add_pass(f"unet.l_conv.{l}", 1 if l == 0 else fm[l - 1], fm[l], c["ksd"][l])
        for l in range(nl - 1):
            add_pass(f"unet.r_conv.0.{l}", fm[l] + fm[l + 1], fm[l], c["ksu"][l])
        add_pass("affs_head", fm[0], 6, [[1, 1, 1]])
        cfg = dict(in_channels=1, num_fmaps=c["num_fmaps"], fmap_inc_factor=c["inc"],
                   downsample_factors=c["dfs"], kernel_size_down=c["ksd"], kernel_size_up=c["ksu"])
        raw = rng.integers(0, 256, size=c["shape"], dtype=np.uint8)
        ref = R.predict_block(cfg, sd, raw, ["affs_head"])[0]
        nc = dict(cfg, outputs={"3d_affs": {"dims": 6}})
        m = Model(nc, precision="f32").load_state_dict(sd)
        assert m.output_shape(c["shape"]) == ref.shape[1:]
        _, f32 = m.predict_u8(torch.from_numpy(raw).cuda(), want_f32=True)
        err = np.abs(f32[0].cpu().numpy() - ref).max()
        print(c["shape"], "f32 err", err)
        assert err < TOL_F32
        m.set_precision("bf16")
        _, b16 = m.predict_u8(torch.from_numpy(raw).cuda(), want_f32=True)
        err = np.abs(b16[0].cpu().numpy() - ref).max()
        print(c["shape"], "bf16 err", err)
        # these nets use a 1.5x gain per layer (large logits): indexing check only
        assert err < 0.1


def test_extract_block_reflect_matches_numpy_pad():
    from bootstrapper_amd.unet import extract_block_reflect
    rng = np.random.default_rng(5)
    vol = rng.integers(0, 256, size=(20, 37, 41), dtype=np.uint8)
    pad = ((14, 30), (46, 50), (46, 46))
    ref_full = np.pad(vol, pad, mode="reflect")
    for off, shape in [((-14, -46, -46), (30, 60, 64)), ((0, 0, 0), (20, 37, 41)), ((8, 20, 30), (42, 67, 57))]:
        got = extract_block_reflect(torch.from_numpy(vol).cuda(), off, shape).cpu().numpy()
        sl = tuple(slice(o + p[0], o + p[0] + s) for o, p, s in zip(off, pad, shape))
        assert np.array_equal(got, ref_full[sl])


def test_forward_before_weights_fails_loudly():
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd._lib import BsmiError
    m = Model({"in_channels": 1, "num_fmaps": 4, "fmap_inc_factor": 2, "downsample_factors": [[1, 2, 2]],
               "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 2, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]],
               "outputs": {"3d_affs": {"dims": 6}}})
    with pytest.raises(BsmiError):
        m(torch.zeros(1, 1, 20, 40, 40, device="cuda"))


FAMILY = ["2d_mtlsd_f4i2", "2d_lsd_f3i3", "2d_affs_f4i2", "3d_lsd_f4i2", "from_2d_mtlsd_f3i2", "from_3d_lsd_f4i2",
          "from_2d_affs_f4i3"]


@pytest.mark.parametrize("tag", FAMILY)
@pytest.mark.parametrize("prec,tol", [("f32", TOL_F32), ("bf16x3", TOL_BF16X3), ("bf16", TOL_BF16)])
def test_model_family_matches_reference_goldens(golden_dir, tag, prec, tol):
    """The other setups of the family: 2-D nets (Conv2d state dicts, (1,C,H,W) inputs), the LSD-only net and the
    second-stage nets (several inputs, num_fmaps_out, (1,3,3) kernels), against outputs of the reference models."""
    from bootstrapper_amd.unet import Model
    from test_oracle_unet import family_case
    nc, sd, ins, x, refs = family_case(golden_dir, tag)
    m = Model(nc, precision=prec).load_state_dict(sd)
    xt = torch.from_numpy(x).cuda()
    if m.two_d:
        y = m(xt[:, :, 0])                                   # (1, C, H, W), like the reference Model.forward
    elif len(ins) > 1:
        splits = np.cumsum([a.shape[0] for a in ins])[:-1]
        y = m(*[torch.from_numpy(p).cuda() for p in np.split(x, splits, axis=1)])   # forward(input_lsds, input_affs)
    else:
        y = m(xt)
    ys = y if isinstance(y, tuple) else (y,)
    assert len(ys) == len(refs)
    for i, (t, ref) in enumerate(zip(ys, refs)):
        got = t[0].cpu().numpy()
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        print(f"{tag} {prec} head{i}: max abs err {err:.3e}")
        assert err < tol
    if prec == "f32":
        # u8 in -> u8 out with the setup's own normalisation (raw: u8/255*2-1, predictions: u8/255)
        raw = torch.from_numpy(np.concatenate(ins, axis=0) if len(ins) > 1 else ins[0]).cuda()
        if m.two_d:
            raw = raw[:, None]                                # (C, 1, H, W): one section
        u8 = m.predict_u8(raw)
        for t, ref in zip(u8, refs):
            ref = ref if ref.ndim == 4 else ref[:, None]
            want = (ref * np.float32(255)).astype(np.uint8)
            diff = np.abs(t.cpu().numpy().astype(np.int32) - want.astype(np.int32))
            assert diff.max() <= 1 and (diff == 0).mean() > 0.99


def test_2d_setup_predicts_a_stack_of_sections(golden_dir):
    """A 2-D setup's kernels have unit depth, so a (C, D, H, W) stack is D independent sections in one pass:
    identical to running the sections one by one."""
    from bootstrapper_amd.unet import Model
    from test_oracle_unet import family_case
    nc, sd, ins, _, _ = family_case(golden_dir, "2d_mtlsd_f4i2")
    m = Model(nc, precision="f32").load_state_dict(sd)
    rng = np.random.default_rng(0)
    vol = torch.from_numpy(rng.integers(0, 256, (7,) + ins[0].shape[1:], dtype=np.uint8)).cuda()   # 7 sections
    stack = torch.stack([vol[0:5], vol[1:6], vol[2:7]])                                               # (3, 5, H, W)
    both = m.predict_u8(stack)
    for z in range(5):
        one = m.predict_u8(stack[:, z:z + 1])
        for a, b in zip(both, one):
            assert torch.equal(a[:, z:z + 1], b)
