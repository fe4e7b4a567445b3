"""The device training step (csrc/train.hip through the C ABI) against the goldens the reference model, loss and
optimizer produced (tests/golden/train_*.npz): loss, every parameter gradient, parameters after two Adam steps."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net_config(meta):
    outs = {"3d_affs": {"3d_affs": {"dims": 6}}, "3d_mtlsd": {"3d_lsds": {"dims": 10}, "3d_affs": {"dims": 6}}}[meta["model"]]
    return {"in_channels": 1, "num_fmaps": meta["num_fmaps"], "fmap_inc_factor": meta["fmap_inc_factor"],
            "downsample_factors": [[1, 2, 2]] * 3, "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
            "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3, "outputs": outs}


@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3_lr1e-2", "mtlsd_f4i2"])
def test_training_step_vs_reference_goldens(golden_dir, tag):
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    d = np.load(os.path.join(golden_dir, f"train_{tag}.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    m = Model(_net_config(meta), precision="f32").load_state_dict(sd)
    tr = Trainer(m, meta["in_shape"], lr=meta["lr"])
    nh = len(m.heads)
    raw = torch.from_numpy(d["x"]).cuda()
    targets = [torch.from_numpy(d[f"gt{i}"][0]).cuda() for i in range(nh)]
    weights = [torch.from_numpy(d[f"w{i}"][0]).cuda() for i in range(nh)]
    worst = []
    for step in range(2):
        loss = tr.forward_backward(raw, targets, weights)
        assert abs(loss - float(d[f"loss{step}"])) < 1e-5 * max(1.0, abs(loss)), (step, loss, float(d[f"loss{step}"]))
        if step == 0:
            for k in sd:
                if "g0:" + k not in d.files:
                    continue
                ref = d["g0:" + k].ravel()
                got = tr.read(k, "grad")
                err = np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())
                worst.append((err, k))
                assert err < 1e-3, (k, err, np.abs(ref).max())
        tr.optimizer_step()
        for k in sd:
            ref = d[f"w{step + 1}:" + k].ravel()
            got = tr.read(k, "param")
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) + meta["lr"] * 5e-2, (step, k)
    print("largest relative gradient error:", max(worst))
    tr.close()


def test_data_parallel_two_ranks(tmp_path, golden_dir):
    """Two ranks (processes) with different samples: the flat gradient buffer is summed over the ranks, Adam sees the
    mean, and both ranks end with the same parameters = Adam(mean gradient) of the oracle's optimizer."""
    import subprocess
    import sys
    from oracle import train_ref as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", os.path.join(root, "tests", "ddp_train_worker.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    d = np.load(os.path.join(golden_dir, "train_affs_f4i2.npz"))
    keys = [k[3:] for k in d.files if k.startswith("w0:")]
    mean = {}
    for k in keys:
        assert np.abs(a["l:" + k] - b["l:" + k]).max() > 0 or "bias" in k       # the ranks really saw different samples
        s = a["l:" + k] + b["l:" + k]
        assert np.allclose(a["s:" + k], s, rtol=1e-6, atol=1e-9) and np.array_equal(a["s:" + k], b["s:" + k])
        assert np.array_equal(a["p:" + k], b["p:" + k])
        mean[k] = s / 2
    ref = T.adam_step({k: d["w0:" + k].ravel() for k in keys}, mean, {}, lr=1e-3)
    for k in keys:
        assert np.abs(a["p:" + k] - ref[k]).max() <= 1e-6 + 1e-3 * 2e-3, k
