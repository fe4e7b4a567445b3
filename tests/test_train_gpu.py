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


@pytest.mark.parametrize("arithmetic", ["f32", "split-bf16", "f32+deterministic", "split-bf16+deterministic"])
@pytest.mark.parametrize("tag", ["affs_f4i2", "affs_f3i3_lr1e-2", "mtlsd_f4i2"])
def test_training_step_vs_reference_goldens(golden_dir, tag, arithmetic):
    """arithmetic "f32": exact f32 MFMA, the reference's own arithmetic.  "split-bf16" (the default of Trainer): the
    convolutions multiply bf16 hi + lo pairs (2^-17 per product): same loss to 1e-5, gradients to 1e-2 of the largest
    entry.  Measured per component (tools/probe_train_err.py): weight gradients 3e-6, input gradients 1e-5, forward 2e-5 on two
    of these nets and 4.6e-3 on the third -- one activation of a 16-channel, few-voxel layer within 1e-5 of zero changes sides
    of the ReLU, and with it the bias gradient by that voxel's share.  f32: 1e-4, measured 3e-6; the parameters after the Adam steps may differ by two steps (2 lr): the first Adam steps move every
    entry by about lr in the direction of its sign, whatever its magnitude, so an entry within the gradient error of zero may
    go the other way (f32: a twentieth of a step)."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    deterministic = arithmetic.endswith("+deterministic")  # the ordered reductions (bsmi_unet_train_set_deterministic): same goldens
    arithmetic = arithmetic.split("+")[0]
    grad_tol, step_tol = (1e-4, 5e-2) if arithmetic == "f32" else (1e-2, 2.1)
    d = np.load(os.path.join(golden_dir, f"train_{tag}.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    m = Model(_net_config(meta), precision="f32").load_state_dict(sd)
    tr = Trainer(m, meta["in_shape"], lr=meta["lr"], arithmetic=arithmetic, deterministic=deterministic)
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
                assert err < grad_tol, (k, err, np.abs(ref).max())
        tr.optimizer_step()
        for k in sd:
            ref = d[f"w{step + 1}:" + k].ravel()
            got = tr.read(k, "param")
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) + meta["lr"] * step_tol, (step, k)
    print("largest relative gradient error:", max(worst))
    tr.close()


@pytest.mark.parametrize("arithmetic", ["split-bf16", "f32"])
def test_deterministic_step_is_bit_reproducible(arithmetic):
    """Trainer(deterministic=True): two runs of two steps (forward, backward, Adam) of the full net on the same inputs give the
    same BITS -- loss, every gradient, every parameter -- like the reference's CPU path.  The default mode
    orders its float atomics by chance: it agrees with the deterministic one to 1e-4 of the largest gradient, not bitwise."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from bootstrapper_amd.synth import synthetic_state_dict
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    sd = synthetic_state_dict(NC, 0)
    shape = (32, 196, 196)
    rng = np.random.default_rng(1)
    x = torch.from_numpy((rng.random(shape, dtype=np.float32) * 2 - 1).astype(np.float32)).cuda()
    gt = w = None
    runs = []
    for deterministic in (True, True, False):
        m = Model(NC, precision="f32").load_state_dict(sd)
        tr = Trainer(m, shape, lr=1e-3, arithmetic=arithmetic, deterministic=deterministic)
        if gt is None:
            out = (6,) + tuple(tr.out_shape)
            gt = torch.from_numpy((rng.random(out) > 0.5).astype(np.float32)).cuda()
            wn = rng.random(out).astype(np.float32)
            wn[rng.random(out) < 0.2] = 0
            w = torch.from_numpy(wn).cuda()
        rec = []
        for step in range(2):
            loss = tr.forward_backward(x, [gt], [w])
            rec.append((np.float64(loss), tr.grads.cpu().numpy().copy()))
            tr.optimizer_step()
            torch.cuda.synchronize()
            rec.append((None, tr.params.cpu().numpy().copy()))
        runs.append(rec)
        tr.close()
        del m
    a, b, c = runs
    for i, ((la, ta), (lb, tb)) in enumerate(zip(a, b)):
        assert la is None or la.tobytes() == lb.tobytes(), (i, la, lb)
        assert ta.tobytes() == tb.tobytes(), f"record {i}: {int((ta != tb).sum())} of {ta.size} values differ between two deterministic runs"
    # the default mode against the deterministic one: the same step up to the order of the additions
    (la, ga), (lc, gc) = a[0], c[0]
    assert abs(la - lc) < 1e-6 * max(1.0, abs(la))
    assert np.abs(ga - gc).max() < 1e-4 * np.abs(ga).max()


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


def test_bs_train_driver(tmp_path):
    """`bs train`: random crops from a labelled Zarr volume, three iterations, a checkpoint in the reference's layout
    that the predict-side Model loads, and resuming from it."""
    from bootstrapper_amd.train import run_training, latest_checkpoint
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.zarr_io import prepare_ds
    rng = np.random.default_rng(4)
    store = str(tmp_path / "vol.zarr")
    raw = rng.integers(0, 256, size=(40, 130, 130), dtype=np.uint8)
    labels = np.zeros((40, 130, 130), dtype=np.uint64)
    for i, (z, y, x) in enumerate(rng.integers(0, 100, size=(40, 3))):
        labels[z % 30:z % 30 + 10, y:y + 30, x:x + 30] = i + 1
    for name, arr in (("raw", raw), ("labels", labels)):
        ds = prepare_ds(f"{store}/{name}", arr.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(20, 64, 64), dtype=arr.dtype)
        ds[:] = arr
    setup = tmp_path / "setup_01"
    setup.mkdir()
    nc = {"in_channels": 1, "num_fmaps": 4, "fmap_inc_factor": 2, "downsample_factors": [[1, 2, 2]] * 3,
          "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3,
          "input_shape": [30, 108, 108], "output_shape": [2, 16, 16],
          "outputs": {"3d_affs": {"dims": 3, "neighborhood": [[-1, 0, 0], [0, -1, 0], [0, 0, -1]], "grow_boundary": 1}}}
    (setup / "net_config.json").write_text(json.dumps(nc))
    cfg = tmp_path / "train.toml"
    cfg.write_text(f'setup_dir = "{setup}"\nvoxel_size = [40, 4, 4]\nmax_iterations = 3\nsave_checkpoints_every = 3\nsave_snapshots_every = 1000\n'
                   f'[[samples]]\nraw = "{store}/raw"\nlabels = "{store}/labels"\n')
    logs = []
    assert run_training(str(cfg), log=logs.append) == 3
    ckpt, step = latest_checkpoint(str(setup))
    assert step == 3 and os.path.basename(ckpt) == "model_checkpoint_3.ckpt"
    assert any("train_loss" in l for l in logs)
    import glob
    from bootstrapper_amd.tb_events import read_scalars
    events = glob.glob(str(setup / "log" / "version_0" / "events.out.tfevents.*"))   # TensorBoardLogger(setup_dir, name="log"), training.py:130
    assert len(events) == 1 and read_scalars(events[0])[0] == "brain.Event:2"
    m = Model(nc, precision="f32").load_checkpoint(ckpt)            # the predict worker's loader (predict.py:98-108)
    y = m(torch.zeros(1, 1, 30, 108, 108, device="cuda"))
    assert tuple(y.shape) == (1, 3, 2, 16, 16) and bool(torch.isfinite(y).all())
    cfg.write_text(cfg.read_text().replace("max_iterations = 3", "max_iterations = 5").replace("save_checkpoints_every = 3", "save_checkpoints_every = 5"))
    assert run_training(str(cfg), log=logs.append) == 5 and latest_checkpoint(str(setup))[1] == 5
    assert glob.glob(str(setup / "log" / "version_1" / "events.out.tfevents.*"))     # the resumed run logs under its own version
    assert any("resuming from" in l for l in logs)


def test_resume_restores_optimizer_state_and_steps_repeat(golden_dir):
    """Two steps, a checkpoint (weights + Adam moments + step count in the Lightning layout), a fresh trainer resumed from it,
    two more steps: the parameters equal four uninterrupted steps.  The weight gradients are accumulated with f32 atomics,
    so two runs of the same step agree to rounding, not bit for bit: the check states that tolerance."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer, save_checkpoint, load_optimizer_state
    d = np.load(os.path.join(golden_dir, "train_affs_f4i2.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    nc = {"in_channels": 1, "num_fmaps": meta["num_fmaps"], "fmap_inc_factor": meta["fmap_inc_factor"],
          "downsample_factors": [[1, 2, 2]] * 3, "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
          "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3, "outputs": {"3d_affs": {"dims": 6}}}
    batch = {"raw": torch.from_numpy(d["x"]).cuda(), "gt_affs": torch.from_numpy(d["gt0"][0]).cuda(),
             "affs_weights": torch.from_numpy(d["w0"][0]).cuda()}

    def run(n_steps, model=None, resume=None):
        m = model or Model(nc, precision="f32").load_state_dict(sd)
        tr = Trainer(m, meta["in_shape"], lr=1e-3)
        if resume:
            assert load_optimizer_state(tr, resume)
        losses = [tr.training_step(batch) for _ in range(n_steps)]
        return m, tr, losses

    _, tr4, losses4 = run(4)
    p4 = {k: tr4.read(k) for k in sd}
    assert tr4.step_count() == 4 and losses4[3] < losses4[0]
    tr4.close()
    _, tr4b, _ = run(4)
    d_rr = max(np.abs(tr4b.read(k) - p4[k]).max() for k in sd)   # run-to-run: atomics reorder the weight-gradient sums
    tr4b.close()
    m2, tr2, _ = run(2)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "model_checkpoint_2.ckpt")
        save_checkpoint(tr2, path, 2)
        tr2.close()
        ck = torch.load(path, map_location="cpu", weights_only=True)
        assert ck["global_step"] == 2 and set(ck["optimizer_states"][0]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
        assert len(ck["optimizer_states"][0]["param_groups"][0]["params"]) == len(sd)
        m3 = Model(nc, precision="f32").load_checkpoint(path)
        _, tr3, _ = run(2, model=m3, resume=path)
        assert tr3.step_count() == 4
        d_resume = max(np.abs(tr3.read(k) - p4[k]).max() for k in sd)
        # without the optimizer state the first resumed steps are bias-corrected sign steps: visibly different
        m4 = Model(nc, precision="f32").load_checkpoint(path)
        _, tr5, _ = run(2, model=m4)
        d_nostate = max(np.abs(tr5.read(k) - p4[k]).max() for k in sd)
        tr3.close(); tr5.close()
        print(f"run-to-run {d_rr:.3e}, resumed {d_resume:.3e}, resumed without optimizer state {d_nostate:.3e}")
        assert d_rr < 1e-4 and d_resume < 1e-4 and d_nostate > 10 * max(d_rr, d_resume, 1e-5)


def test_predict_after_training_uses_the_trained_weights(golden_dir):
    """Trainer.close() hands the trained parameters back to the Model: a predict in the split-bf16 (or bf16) mode on the same
    Model afterwards re-packs its weight images from them -- equal to a fresh Model loaded with the trained parameters."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    d = np.load(os.path.join(golden_dir, "train_affs_f4i2.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    nc = _net_config(meta)
    batch = {"raw": torch.from_numpy(d["x"]).cuda(), "gt_affs": torch.from_numpy(d["gt0"][0]).cuda(),
             "affs_weights": torch.from_numpy(d["w0"][0]).cuda()}
    m = Model(nc).load_state_dict(sd)                       # default precision (bf16x3): its images exist before training
    x = torch.from_numpy(d["x"]).cuda()[None, None]
    before = m(x).clone()
    tr = Trainer(m, meta["in_shape"], lr=1e-2)
    for _ in range(3):
        tr.training_step(batch)
    trained = {k: tr.read(k).reshape(sd[k].shape) for k in sd}
    tr.close()
    for prec in ("bf16x3", "bf16", "f32"):
        got = m.set_precision(prec)(x)
        want = Model(nc, precision=prec).load_state_dict(trained)(x)
        assert torch.equal(got, want), prec
    assert float((m.set_precision("bf16x3")(x) - before).abs().max()) > 1e-4    # and they are not the old weights


def test_ranks_draw_different_samples(tmp_path):
    """`bs train` seeds its sample stream with 42 + rank: two data-parallel ranks must not train on the same crops."""
    from bootstrapper_amd.train import make_sample_source
    from bootstrapper_amd.zarr_io import prepare_ds
    rng = np.random.default_rng(4)
    store = str(tmp_path / "vol.zarr")
    raw = rng.integers(0, 256, size=(40, 130, 130), dtype=np.uint8)
    labels = rng.integers(1, 9, size=(40, 130, 130)).astype(np.uint64)
    for name, arr in (("raw", raw), ("labels", labels)):
        ds = prepare_ds(f"{store}/{name}", arr.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(20, 64, 64), dtype=arr.dtype)
        ds[:] = arr
    nc = {"input_shape": [30, 108, 108], "output_shape": [2, 16, 16],
          "outputs": {"3d_affs": {"dims": 3, "neighborhood": [[-1, 0, 0], [0, -1, 0], [0, 0, -1]], "grow_boundary": 0}}}
    cfg = {"samples": [{"raw": f"{store}/raw", "labels": f"{store}/labels"}]}
    a, b, a2 = (next(make_sample_source(cfg, nc, 0, r)) for r in (0, 1, 0))
    assert torch.equal(a["raw"], a2["raw"]) and not torch.equal(a["raw"], b["raw"])
    # the producer thread of `bs train` (training.py:107-114's loader workers): the same batches in the same order, ready ahead
    from bootstrapper_amd.train import PrefetchSource
    inline = make_sample_source(cfg, nc, 0, 0)
    ahead = PrefetchSource(make_sample_source(cfg, nc, 0, 0), depth=3)
    try:
        for _ in range(7):
            x, y = next(inline), next(ahead)
            assert set(x) == set(y) and all(torch.equal(x[k], y[k]) for k in x)
    finally:
        ahead.close()
    assert not ahead.thread.is_alive()

    class Broken:
        def __iter__(self):
            return self

        def __next__(self):
            raise ValueError("no samples")
    with pytest.raises(ValueError, match="no samples"):     # a failure in the producer reaches the training thread
        next(PrefetchSource(Broken()))


def test_full_net_training_step_vs_cpu_oracle():
    """The full 3d_affs net (94.7 M parameters, 1500/1800-channel layers: multi-tile weight gradients, persistent
    split-K input gradients) on the reference's training block, against the CPU oracle's autograd (about 20 s)."""
    from oracle import train_ref as T
    from oracle import unet_ref as R
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from bootstrapper_amd.synth import synthetic_state_dict
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    sd = synthetic_state_dict(NC, 0)
    shape = (32, 196, 196)
    rng = np.random.default_rng(0)
    x = (rng.random(shape, dtype=np.float32) * 2 - 1).astype(np.float32)
    ref_loss = ref_grads = None
    # exact f32 first (2e-3: the oracle's own autograd differs from it by 2.4e-4), then the default split-bf16 arithmetic
    for arithmetic, tol in (("f32", 2e-3), ("split-bf16", 5e-3)):
        m = Model(NC, precision="f32").load_state_dict(sd)
        tr = Trainer(m, shape, arithmetic=arithmetic)
        out = (6,) + tuple(tr.out_shape)
        if ref_grads is None:
            gt = (rng.random(out) > 0.5).astype(np.float32)
            w = rng.random(out).astype(np.float32)
            w[rng.random(out) < 0.2] = 0
        loss = tr.forward_backward(torch.from_numpy(x).cuda(), [torch.from_numpy(gt).cuda()], [torch.from_numpy(w).cuda()])
        if ref_grads is None:
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            ref_loss, ref_grads, _ = T.loss_and_grads(R.default_cfg(12, 5), sd, x, [gt[None]], [w[None]], ["affs_head"])
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (arithmetic, loss, ref_loss)
        worst = (0.0, "")
        for k, ref in ref_grads.items():
            got = tr.read(k, "grad").reshape(ref.shape)
            err = float(np.abs(got - ref).max() / max(1e-12, np.abs(ref).max()))
            worst = max(worst, (err, k))
            assert err < tol, (arithmetic, k, err)
        print(f"full net, {arithmetic}: loss", loss, "largest relative gradient error", worst)
        tr.close()
        del m


@pytest.mark.parametrize("tag", ["2d_mtlsd_f4i2", "from_2d_mtlsd_f3i2", "3d_lsd_f4i2"])
def test_training_step_model_family_vs_oracle(golden_dir, tag):
    """The training step on the other setups of the family (Conv2d state dicts, num_fmaps_out, (1,3,3) kernels,
    12 input channels): loss and every gradient against the CPU oracle's autograd over the forward that
    tests/test_oracle_unet.py pins to the reference models."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from oracle import train_ref as T
    from oracle import unet_ref as R
    from test_oracle_unet import family_case
    nc, sd, _, x, refs = family_case(golden_dir, tag)
    m = Model(nc, precision="f32").load_state_dict(sd)
    tr = Trainer(m, x.shape[2:], lr=1e-3)
    rng = np.random.default_rng(2)
    shapes = [r.shape if r.ndim == 4 else (r.shape[0], 1) + r.shape[1:] for r in refs]
    targets = [rng.random(s, dtype=np.float32) for s in shapes]
    weights = [(rng.random(s, dtype=np.float32) * (rng.random(s) > 0.3)).astype(np.float32) for s in shapes]
    loss = tr.forward_backward(torch.from_numpy(x[0]).cuda(), [torch.from_numpy(t).cuda() for t in targets],
                               [torch.from_numpy(w).cuda() for w in weights])
    heads = [R.FAMILY_HEADS[k] for k in nc["outputs"]]
    lsd = {k: v.numpy() for k, v in R.lift_sd(sd).items()}
    ref_loss, ref_grads, _ = T.loss_and_grads(R.lift_cfg(nc), lsd, x[0], [t[None] for t in targets], [w[None] for w in weights], heads)
    assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    worst = 0.0
    for k, g in ref_grads.items():
        got = tr.read(k, "grad")
        err = np.abs(got - g.ravel()).max() / max(1e-6, np.abs(g).max())
        worst = max(worst, err)
        assert err < 1e-3, (k, err)
    print(f"{tag}: loss {loss:.6f}, largest relative gradient error {worst:.2e}")
    tr.optimizer_step()
    assert tr.param_shapes()[next(iter(sd))] == tuple(sd[next(iter(sd))].shape)     # Conv2d shapes survive for checkpoints
    tr.close()


@pytest.mark.parametrize("steps,only_xy,with_mask", [(1, True, False), (2, True, True), (1, False, True), (0, True, False), (3, False, False)])
def test_affinity_targets_vs_oracle(steps, only_xy, with_mask):
    """GrowBoundary -> AddAffinities -> BalanceLabels in one device call, against the numpy / scipy restatement
    (binary_erosion per label, as the reference's gp/custom_grow_boundary.py does)."""
    from bootstrapper_amd.train import affinity_targets
    from oracle import train_ref as TR
    rng = np.random.default_rng(steps * 7 + only_xy + 2 * with_mask)
    # blocky labels with background gaps: Voronoi cells of random seeds, some cells set to 0
    D, H, W = 6, 40, 37
    seeds = rng.integers(0, [D, H, W], size=(25, 3))
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    dist = ((zz[..., None] - seeds[:, 0]) * 4) ** 2 + (yy[..., None] - seeds[:, 1]) ** 2 + (xx[..., None] - seeds[:, 2]) ** 2
    labels = (dist.argmin(-1) + 1).astype(np.int64) * 1000003
    labels[np.isin(labels // 1000003, [3, 7, 11])] = 0
    unl = (labels > 0).astype(np.uint8)
    if with_mask:
        unl[:, 5:12, 20:30] = 0          # an unknown region that still carries labels
        unl[2:4, 25:, :6] = 0
    nhood = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -9, 0], [0, 0, -9]]
    grown, affs, weights = TR.affinity_targets(labels, unl, nhood, steps, only_xy)
    lab_t = torch.from_numpy(labels.copy()).cuda()
    a, w = affinity_targets(lab_t, torch.from_numpy(unl).cuda(), nhood, steps, only_xy)
    got_grown = lab_t.cpu().numpy()
    assert np.array_equal(got_grown, grown)
    assert np.array_equal(a.cpu().numpy(), affs)
    assert np.allclose(w.cpu().numpy(), weights, rtol=1e-6, atol=0)
    assert (grown != labels).any() == (steps > 0)


@pytest.mark.parametrize("df,sigma,vs", [(1, (6.0, 6.0, 6.0), (3.0, 2.0, 2.0)), (2, (80.0, 80.0, 80.0), (40.0, 4.0, 4.0)), (2, (10.0, 12.0, 9.0), (2.0, 2.0, 3.0))])
def test_lsd_targets_vs_numpy_restatement(df, sigma, vs):
    """bsmi_train_lsd_targets against the numpy / scipy restatement of lsd's LsdExtractor (oracle/lsd_ref.py; the lsd
    package itself is absent: parity unpinned).  Floating point: the kernel sums the window directly in float64 with
    float32 weights, scipy filters axis by axis -- agreement to 1e-4 of the [0, 1] range."""
    from bootstrapper_amd.train import lsd_targets
    from oracle.lsd_ref import lsd_targets as ref_lsd
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(7)
    shape = (16, 72, 64)
    blobs = gaussian_filter(rng.random(shape), (1, 4, 4))
    labels = (np.digitize(blobs, np.quantile(blobs, [0.2, 0.4, 0.6, 0.8])) + 1).astype(np.int64)
    labels[blobs < np.quantile(blobs, 0.1)] = 0                       # background
    labels[:, :, 40:] += 7                                            # more objects, a straight boundary
    unl = (rng.random(shape) > 0.1).astype(np.uint8)
    off, roi = (4, 16, 12), (8, 40, 36)
    lsds, w = lsd_targets(torch.from_numpy(labels).cuda(), off, roi, sigma, vs, df, torch.from_numpy(unl).cuda())
    ref, wref = ref_lsd(labels, off, roi, sigma, vs, df, unl)
    got = lsds.cpu().numpy()
    assert got.shape == (10,) + roi and np.array_equal(w.cpu().numpy(), wref)
    err = np.abs(got - ref).max(axis=(1, 2, 3))
    print("max abs error per channel", err)
    assert err.max() < 1e-4
    fg = labels[tuple(slice(o, o + s) for o, s in zip(off, roi))] != 0
    assert np.all(got[:, ~fg] == 0) and got[9][fg].min() > 0 and 0.2 < got[0][fg].mean() < 0.8
    assert got[3:6][:, fg].std() > 0.01                               # the variances vary over the objects


def test_bs_train_mtlsd_driver(tmp_path):
    """`bs train` for the two-headed 3d_mtlsd setup (BASELINE config 3 in small): LSD and affinity targets built on the
    device from a labelled Zarr volume, the sum of both losses trained for three iterations."""
    from bootstrapper_amd.train import run_training, latest_checkpoint, make_sample_source
    from bootstrapper_amd.zarr_io import prepare_ds
    rng = np.random.default_rng(4)
    store = str(tmp_path / "vol.zarr")
    raw = rng.integers(0, 256, size=(40, 130, 130), dtype=np.uint8)
    labels = np.zeros((40, 130, 130), dtype=np.uint64)
    for i, (z, y, x) in enumerate(rng.integers(0, 100, size=(40, 3))):
        labels[z % 30:z % 30 + 10, y:y + 30, x:x + 30] = i + 1
    for name, arr in (("raw", raw), ("labels", labels)):
        ds = prepare_ds(f"{store}/{name}", arr.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(20, 64, 64), dtype=arr.dtype)
        ds[:] = arr
    setup = tmp_path / "setup_02"
    setup.mkdir()
    nc = {"in_channels": 1, "num_fmaps": 4, "fmap_inc_factor": 2, "downsample_factors": [[1, 2, 2]] * 3,
          "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3,
          "input_shape": [30, 108, 108], "output_shape": [2, 16, 16],
          "outputs": {"3d_lsds": {"dims": 10, "sigma": 80, "downsample": 2},
                      "3d_affs": {"dims": 3, "neighborhood": [[-1, 0, 0], [0, -1, 0], [0, 0, -1]], "grow_boundary": 1}}}
    (setup / "net_config.json").write_text(json.dumps(nc))
    cfg = tmp_path / "train.toml"
    cfg.write_text(f'setup_dir = "{setup}"\nvoxel_size = [40, 4, 4]\nmax_iterations = 3\nsave_checkpoints_every = 3\nsave_snapshots_every = 1000\n'
                   f'[[samples]]\nraw = "{store}/raw"\nlabels = "{store}/labels"\n')
    b = next(make_sample_source({"samples": [{"raw": f"{store}/raw", "labels": f"{store}/labels"}], "voxel_size": [40, 4, 4]}, nc))
    assert tuple(b["gt_lsds"].shape) == (10, 2, 16, 16) and tuple(b["gt_affs"].shape) == (3, 2, 16, 16)
    assert float(b["gt_lsds"].max()) <= 1.0 and float(b["lsds_weights"].sum()) > 0
    logs = []
    cfg.write_text(cfg.read_text().replace("save_snapshots_every = 1000", "save_snapshots_every = 2"))
    assert run_training(str(cfg), log=logs.append) == 3
    assert latest_checkpoint(str(setup))[1] == 3 and any("train_loss" in l for l in logs)
    # snapshots at step 1 and every 2 steps (training.py:46-93): batch + predictions, [-1, 1] floats as uint8, centred offsets
    from bootstrapper_amd.zarr_io import open_ds
    for step in (1, 2):
        snap = str(setup / "snapshots" / f"batch_{step}_rank_0.zarr")
        raw_s, lsds_s, pa = open_ds(snap + "/raw"), open_ds(snap + "/gt_lsds"), open_ds(snap + "/pred_affs")
        assert raw_s.dtype == np.uint8 and raw_s.shape == (30, 108, 108) and raw_s.offset == (0, 0, 0) and raw_s.voxel_size == (40, 4, 4)
        assert lsds_s.shape == (10, 2, 16, 16) and lsds_s.offset == (14 * 40, 46 * 4, 46 * 4) and pa.shape == (3, 2, 16, 16)
        assert pa.dtype == np.float32 and 0 < float(pa[:].mean()) < 1
    assert not (setup / "snapshots" / "batch_3_rank_0.zarr").exists()


_WGRAD_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from bootstrapper_amd.unet import Model
from bootstrapper_amd.training import Trainer
from bootstrapper_amd.synth import synthetic_state_dict
cfg = {"in_channels": 1, "num_fmaps": 12, "fmap_inc_factor": 5, "downsample_factors": [[1, 2, 2], [1, 2, 2]],
       "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 3, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 2, "outputs": {"3d_affs": {"dims": 6}}}
shape = (22, 116, 116)
sd = synthetic_state_dict(cfg, 3)
m = Model(cfg, precision="f32").load_state_dict(sd)
tr = Trainer(m, shape, lr=1e-4)
g = torch.Generator().manual_seed(5)
raw = torch.rand((1, 1) + shape, generator=g).cuda()
out = tr.out_shape
tg = [torch.rand((6,) + tuple(out), generator=g).cuda()]
wt = [torch.rand((6,) + tuple(out), generator=g).cuda()]
tr.forward_backward(raw, tg, wt)
np.savez(sys.argv[2], **{k: tr.read(k, "grad") for k in sd if k.endswith("weight")})
tr.close()
"""


def test_split_bf16_weight_gradients_match_f32_form(tmp_path):
    """wgrad_x3_kernel (packed operands, split-bf16 MFMA, tap-major accumulation) against the f32 MFMA kernels it replaced
    (BSMI_WGRAD_X3=0, read once per process: two child processes), on a net with 12 / 60 / 300 channels, 3-group and
    ragged lines: every weight gradient to 2e-5 of its largest entry."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_WGRAD_CHILD)
    outs = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"g{mode}.npz")
        env = dict(os.environ, BSMI_WGRAD_X3=mode)
        subprocess.run([sys.executable, str(script), root, path], check=True, env=env, timeout=300)
        outs[mode] = np.load(path)
    assert len(outs["1"].files) >= 10
    worst = 0.0
    for k in outs["1"].files:
        a, b = outs["1"][k], outs["0"][k]
        scale = max(np.abs(b).max(), 1e-12)
        err = np.abs(a - b).max() / scale
        worst = max(worst, err)
        assert err < 2e-5, (k, err, scale)
    print("largest relative difference split-bf16 vs f32 weight gradients:", worst)


def test_f32_inference_between_training_steps_sees_current_weights():
    """The f32 weight images of the launches that train in their split-bf16 form are refreshed lazily (an optimizer step
    leaves them stale; an f32 forward on the handle brings them up to date): the prediction of the training handle after
    two steps equals that of a fresh model loaded with the trained parameters."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from bootstrapper_amd.synth import synthetic_state_dict
    cfg = {"in_channels": 1, "num_fmaps": 12, "fmap_inc_factor": 5, "downsample_factors": [[1, 2, 2], [1, 2, 2]],
           "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 3, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 2, "outputs": {"3d_affs": {"dims": 6}}}
    shape = (22, 116, 116)
    sd = synthetic_state_dict(cfg, 3)
    m = Model(cfg, precision="f32").load_state_dict(sd)
    tr = Trainer(m, shape, lr=1e-3)
    g = torch.Generator().manual_seed(5)
    raw = torch.rand((1, 1) + shape, generator=g).cuda()
    out = tuple(tr.out_shape)
    batch = {"raw": raw, "gt_affs": torch.rand((6,) + out, generator=g).cuda(), "affs_weights": torch.rand((6,) + out, generator=g).cuda()}
    before = m.forward(raw)[0].clone()
    for _ in range(2):
        tr.training_step(batch)
    after = m.forward(raw)[0].clone()
    trained = {k: tr.read(k, "param").reshape(np.asarray(v).shape) for k, v in sd.items()}
    fresh = Model(cfg, precision="f32").load_state_dict(trained)
    want = fresh.forward(raw)[0]
    assert float((after - before).abs().max()) > 1e-4          # the steps did move the prediction
    assert float((after - want).abs().max()) < 1e-6, float((after - want).abs().max())
    tr.training_step(batch)                                    # images stale again when training ends
    trained = {k: tr.read(k, "param").reshape(np.asarray(v).shape) for k, v in sd.items()}
    tr.close()
    want = Model(cfg, precision="f32").load_state_dict(trained).forward(raw)[0]
    assert float((m.forward(raw)[0] - want).abs().max()) < 1e-6
