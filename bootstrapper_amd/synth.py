"""Seeded synthetic weights and volumes for benchmarks and full-size property tests
(there is no network for checkpoints or CREMI data)."""
import numpy as np
import torch

from .unet import HEAD_OF_OUTPUT


def synthetic_state_dict(net_config, seed=0, head_gain=2.0):
    """Random weights with the reference's state-dict keys and shapes.

    torch's default Conv3d init shrinks the activation variance by ~3x per layer, which
    after 16 stacked layers makes every affinity the same constant; variance-preserving
    scales are used instead so that the predicted affinities are structured and the
    watershed downstream sees a non-degenerate field."""
    rng = np.random.default_rng(seed)
    nf, inc = net_config["num_fmaps"], net_config["fmap_inc_factor"]
    dfs = net_config["downsample_factors"]
    nl = len(dfs) + 1
    ksd = net_config.get("kernel_size_down") or [[[3, 3, 3], [3, 3, 3]]] * nl
    ksu = net_config.get("kernel_size_up") or [[[3, 3, 3], [3, 3, 3]]] * (nl - 1)
    fm = [nf * inc ** l for l in range(nl)]
    sd = {}

    def add_pass(prefix, cin, cout, ks, gain=1.0):
        ci = cin
        for i, k in enumerate(ks):
            fan = ci * int(np.prod(k))
            sd[f"{prefix}.conv_pass.{2 * i}.weight"] = (
                rng.standard_normal((cout, ci, *k), dtype=np.float32) * np.float32(gain * np.sqrt(1.0 / fan)))
            sd[f"{prefix}.conv_pass.{2 * i}.bias"] = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05)
            ci = cout
        sd[f"{prefix}.residual.0.weight"] = (
            rng.standard_normal((cout, cin, 1, 1, 1), dtype=np.float32) * np.float32(gain * np.sqrt(1.0 / cin)))
        sd[f"{prefix}.residual.0.bias"] = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05)

    for l in range(nl):
        add_pass(f"unet.l_conv.{l}", net_config["in_channels"] if l == 0 else fm[l - 1], fm[l], ksd[l])
    for l in range(nl - 1):
        add_pass(f"unet.r_conv.0.{l}", fm[l] + fm[l + 1], fm[l], ksu[l])
    for name, val in net_config["outputs"].items():
        add_pass(HEAD_OF_OUTPUT[name], nf, int(val["dims"]), [[1, 1, 1]], gain=head_gain)
    return sd


def synthetic_volume(shape, seed=0, device="cuda", corr=(2, 8, 8)):
    """Blobby uint8 volume: low-resolution seeded noise, trilinearly upsampled by `corr`,
    plus a little white noise, min-max scaled to [0, 255]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo = [max(2, (s + c - 1) // c + 1) for s, c in zip(shape, corr)]
    coarse = torch.rand(lo, generator=g, dtype=torch.float32).to(device)
    fine = torch.nn.functional.interpolate(coarse[None, None], size=tuple(int(l * c) for l, c in zip(lo, corr)),
                                           mode="trilinear", align_corners=False)[0, 0]
    fine = fine[: shape[0], : shape[1], : shape[2]]
    g2 = torch.Generator(device=device).manual_seed(seed + 1)
    # white noise generated slice by slice to bound memory
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    lo_v, hi_v = float(fine.min()), float(fine.max())
    for z in range(shape[0]):
        n = torch.rand(shape[1:], generator=g2, dtype=torch.float32, device=device)
        v = (fine[z] - lo_v) / max(hi_v - lo_v, 1e-6) * 0.9 + n * 0.1
        out[z] = (v.clamp(0, 1) * 255).to(torch.uint8)
    return out
