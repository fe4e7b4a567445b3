"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of one training step.

Only tests/ may import this module.  Pinned against tests/golden/train_*.npz, produced by
tools/gen_goldens_train.py running the reference model, loss and optimizer (tests/test_oracle_train.py).

Reference being restated (paths relative to /root/reference/bootstrapper):
  models/3d_affs/model.py:67-92    WeightedMSELoss: mean of w*(p-t)^2 over the voxels with w > 0, or over all
                                   voxels if the weighted error is zero everywhere
  models/3d_mtlsd/model.py         the same per head, summed (lsds + affs)
  models/3d_affs/train.py:152-159  training_step: loss(model(raw), gt, weights); Adam(lr=0.5e-4), torch defaults
                                   betas (0.9, 0.999), eps 1e-8, no weight decay
"""
import numpy as np
import torch

from . import unet_ref as R


def weighted_mse(pred, target, weights):
    scale = weights * (pred - target) ** 2
    if len(torch.nonzero(scale)) != 0:
        return torch.mean(torch.masked_select(scale, torch.gt(weights, 0)))
    return torch.mean(scale)


def forward_train(cfg, sd, x, heads):
    """x: float32 tensor (1, Cin, D, H, W); sd: dict of tensors (requires_grad as wanted) -> list of head outputs."""
    z = R.unet_forward(cfg, sd, x)
    return [R.conv_pass(z, sd, h, [[1, 1, 1]], "Sigmoid") for h in heads]


def loss_and_grads(cfg, sd_np, x, targets, weights, heads):
    """-> (loss float, {name: grad ndarray}, [pred ndarray])"""
    sd = {k: torch.from_numpy(np.array(v, dtype=np.float32)).requires_grad_(True) for k, v in sd_np.items()}
    preds = forward_train(cfg, sd, torch.from_numpy(x)[None, None] if x.ndim == 3 else torch.from_numpy(x)[None], heads)
    loss = sum(weighted_mse(p, torch.from_numpy(t), torch.from_numpy(w)) for p, t, w in zip(preds, targets, weights))
    loss.backward()
    return float(loss.item()), {k: v.grad.numpy() for k, v in sd.items()}, [p.detach().numpy() for p in preds]


def adam_step(params, grads, state, lr=0.5e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad, no weight decay) on dicts of float32 ndarrays; state = {"t", "m", "v"}."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = {}
    for k, p in params.items():
        g = grads[k].astype(np.float32)
        m = state.setdefault("m", {}).get(k, np.zeros_like(p))
        v = state.setdefault("v", {}).get(k, np.zeros_like(p))
        m = np.float32(beta1) * m + np.float32(1 - beta1) * g
        v = np.float32(beta2) * v + np.float32(1 - beta2) * g * g
        state["m"][k], state["v"][k] = m, v
        bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
        step_size = lr / bc1
        denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
        out[k] = (p - np.float32(step_size) * (m / denom)).astype(np.float32)
    return out
