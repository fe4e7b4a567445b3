#!/usr/bin/env python3
"""Generate training-step golden vectors by running the REFERENCE model, loss and optimizer.

Build container only (needs /root/reference).  For a width-reduced net: seeded input, targets and weights,
the reference `Model` (models/3d_affs/model.py:28-64, models/3d_mtlsd/model.py) with its `WeightedMSELoss`
(model.py:67-92) and `torch.optim.Adam(lr=0.5e-4)` (models/3d_affs/train.py:158-159) for two steps.  Stored:
the initial state_dict, the batch, the loss of both steps, every parameter gradient of step 0, and the
parameters after each step.  Nothing of the reference is copied: inputs and outputs only.

Usage: python tools/gen_goldens_train.py   (writes tests/golden/train_*.npz)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_goldens_unet import OUT, load_ref  # noqa: E402


def train_case(model_name, tag, num_fmaps, inc, in_shape, seed, lr):
    model_mod, _ = load_ref(model_name)
    torch.manual_seed(seed)
    net = model_mod.Model(num_fmaps=num_fmaps, fmap_inc_factor=inc)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
            if "head" in n and n.endswith("weight"):
                p.mul_(6.0)
    net.train()
    rng = np.random.default_rng(seed)
    x = (rng.random(in_shape, dtype=np.float32) * 2 - 1).astype(np.float32)
    xt = torch.from_numpy(x)[None, None]
    with torch.no_grad():
        y0 = net(xt)
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    batch = {}
    for i, y in enumerate(ys):
        batch[f"gt{i}"] = (rng.random(tuple(y.shape)) > 0.5).astype(np.float32) if i == len(ys) - 1 else rng.random(tuple(y.shape)).astype(np.float32)
        w = rng.random(tuple(y.shape)).astype(np.float32) * 2
        w[rng.random(tuple(y.shape)) < 0.3] = 0          # unlabelled voxels: the masked mean of the loss
        batch[f"w{i}"] = w
    loss_fn = model_mod.WeightedMSELoss()
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    arrs = dict(x=x, config=np.frombuffer(json.dumps(dict(model=model_name, num_fmaps=num_fmaps, fmap_inc_factor=inc,
                                                            in_shape=list(in_shape), lr=lr)).encode(), dtype=np.uint8))
    arrs.update(batch)
    for k, v in net.state_dict().items():
        arrs["w0:" + k] = v.detach().numpy().copy()
    for step in range(2):
        opt.zero_grad()
        pred = net(xt)
        preds = pred if isinstance(pred, (tuple, list)) else (pred,)
        args = []
        for i, p in enumerate(preds):
            args += [p, torch.from_numpy(batch[f"gt{i}"]), torch.from_numpy(batch[f"w{i}"])]
        if len(preds) == 2:  # mtlsd: (lsds_prediction, lsds_target, lsds_weights, affs_prediction, affs_target, affs_weights)
            loss = loss_fn(*args)
        else:
            loss = loss_fn(*args)
        loss.backward()
        arrs[f"loss{step}"] = np.float32(loss.item())
        if step == 0:
            for n, p in net.named_parameters():
                arrs["g0:" + n] = p.grad.detach().numpy().copy()
        opt.step()
        for k, v in net.state_dict().items():
            arrs[f"w{step + 1}:" + k] = v.detach().numpy().copy()
    path = os.path.join(OUT, f"train_{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(path, "loss", arrs["loss0"], arrs["loss1"], sum(p.numel() for p in net.parameters()), "params")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    train_case("3d_affs", "affs_f4i2", 4, 2, (30, 108, 108), seed=11, lr=0.5e-4)
    train_case("3d_affs", "affs_f3i3_lr1e-2", 3, 3, (30, 108, 116), seed=12, lr=1e-2)
    train_case("3d_mtlsd", "mtlsd_f4i2", 4, 2, (31, 108, 108), seed=13, lr=0.5e-4)
