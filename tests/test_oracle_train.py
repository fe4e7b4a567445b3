"""The training-step oracle against goldens produced by the reference model, loss and optimizer."""
import json
import os

import numpy as np
import pytest

from oracle import train_ref as T
from oracle import unet_ref as R


@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3_lr1e-2", "mtlsd_f4i2"])
def test_training_step_oracle_vs_reference(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, f"train_{tag}.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    cfg = R.default_cfg(meta["num_fmaps"], meta["fmap_inc_factor"])
    heads = R.head_names(meta["model"])
    params = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    targets = [d[f"gt{i}"] for i in range(len(heads))]
    weights = [d[f"w{i}"] for i in range(len(heads))]
    state = {}
    for step in range(2):
        loss, grads, _ = T.loss_and_grads(cfg, params, d["x"], targets, weights, heads)
        assert abs(loss - float(d[f"loss{step}"])) < 2e-6 * max(1.0, abs(loss))
        if step == 0:
            for k, g in grads.items():
                ref = d["g0:" + k]
                assert np.abs(g - ref).max() <= 1e-6 * max(1e-3, np.abs(ref).max()), k
        params = T.adam_step(params, grads, state, lr=meta["lr"])
        for k, p in params.items():
            ref = d[f"w{step + 1}:" + k]
            assert np.abs(p - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) + meta["lr"] * 2e-3, (step, k)


def test_weighted_mse_all_zero_branch():
    import torch
    p = torch.tensor([0.2, 0.7]); t = p.clone(); w = torch.tensor([0.0, 3.0])
    assert float(T.weighted_mse(p, t, w)) == 0.0          # scale is zero everywhere: plain mean
    t2 = torch.tensor([0.0, 0.7])
    assert float(T.weighted_mse(p, t2, w)) == 0.0         # the only error sits on an unweighted voxel
    w2 = torch.tensor([2.0, 0.0])
    assert abs(float(T.weighted_mse(p, t2, w2)) - 2.0 * 0.04) < 1e-7
