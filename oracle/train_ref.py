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


# ---- affinity training targets (models/3d_affs/train.py:127-139) ---------------------------------------------
def grow_boundary(labels, unlabelled, steps, only_xy=True, background=0):
    """gp.GrowBoundary as gp/custom_grow_boundary.py:71-110 states it, with a fixed step count: per label, erode
    (labels == label) | (mask == 0) `steps` times (scipy binary_erosion, border_value=1); voxels in no eroded mask
    become background.  Returns a new array."""
    from scipy.ndimage import binary_erosion
    gt = np.array(labels)
    if only_xy:
        for z in range(gt.shape[0]):
            gt[z] = grow_boundary(gt[z], None if unlabelled is None else unlabelled[z], steps, False, background)
        return gt
    foreground = np.zeros(gt.shape, dtype=bool)
    masked = None if unlabelled is None else np.equal(unlabelled, 0)
    for label in np.unique(gt):
        if label == background:
            continue
        label_mask = gt == label
        if masked is not None:
            label_mask = np.logical_or(label_mask, masked)
        if steps > 0:
            label_mask = binary_erosion(label_mask, iterations=steps, border_value=1)
        foreground |= label_mask
    gt[~foreground] = background
    return gt


def affinities_from_labels(labels, neighborhood):
    """gunpowder AddAffinities (seg_to_affgraph): aff[e][p] = labels[p] == labels[p + nhood[e]] and labels[p] > 0;
    positions whose neighbour falls outside the block get affinity 0 and mask 0."""
    shape = labels.shape
    affs = np.zeros((len(neighborhood),) + shape, np.float32)
    mask = np.zeros_like(affs)
    for e, off in enumerate(neighborhood):
        src, dst = [], []
        for o, n in zip(off, shape):
            lo, hi = max(0, -o), min(n, n - o)
            dst.append(slice(lo, hi))
            src.append(slice(lo + o, hi + o))
        a, b = labels[tuple(dst)], labels[tuple(src)]
        affs[(e,) + tuple(dst)] = ((a == b) & (a > 0)).astype(np.float32)
        mask[(e,) + tuple(dst)] = 1.0
    return affs, mask


def balance_labels(affs, mask, clip=(0.05, 0.95)):
    """gunpowder BalanceLabels, two classes over the whole sample: w = mask / (2 * clipped class fraction)."""
    total = max(float(mask.sum()), 1.0)
    frac = min(max(float((affs * mask).sum()) / total, clip[0]), clip[1])
    return (mask * np.where(affs > 0, np.float32(1.0 / (2.0 * frac)), np.float32(1.0 / (2.0 * (1.0 - frac))))).astype(np.float32)


def affinity_targets(labels, unlabelled, neighborhood, grow_steps, only_xy=True):
    grown = grow_boundary(labels, unlabelled, grow_steps, only_xy)
    affs, mask = affinities_from_labels(grown, neighborhood)
    if unlabelled is not None:
        mask = mask * (unlabelled > 0)[None]
    return grown, affs, balance_labels(affs, mask)
