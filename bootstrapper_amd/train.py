"""`bs train` on the device: config checks, a minimal sample source and the training loop.

Reference being mirrored (paths relative to /root/reference/bootstrapper):
  train.py:13-120                 setup_train: sample path checks (same error texts), setup_dir / iteration settings
  models/3d_affs/train.py:160-199 train(setup_dir, voxel_size, max_iterations, samples, save_checkpoints_every, ...)
  training.py:96-137              fit(): seed 42, checkpoints `model_checkpoint_<step>`, resume from the latest one

What is NOT restated: the gunpowder augmentation chain of models/3d_affs/train.py:88-117 (SimpleAugment,
DeformAugment, ShiftAugment, noise / intensity / gamma / impulse / smooth / defect augmentations) and the snapshot
callback -- third-party pipeline code outside the hot path.  `SampleSource` does the deterministic part only: random
location with the >= 5 % labelled-voxel rejection, Normalize + IntensityScaleShift(2, -1), then GrowBoundary,
AddAffinities on the configured neighbourhood and BalanceLabels in one device call (`affinity_targets`).  The arithmetic of the step itself is libbsmi (csrc/train.hip).
"""
import ctypes as C
import glob
import json
import os
import re

import numpy as np
import torch

try:
    import tomllib as _toml
except ImportError:  # python < 3.11
    import tomli as _toml

from . import _lib
from .zarr_io import open_ds


def setup_train(config_file):
    """train.py:13-120: load the TOML, check the sample datasets, return the config."""
    with open(config_file, "rb") as f:
        config = _toml.load(f)
    samples = config.get("samples", [])
    if "samples" in config and not samples:
        raise ValueError(f"No training samples provided in {config_file}")
    for sample in samples:
        raw, labels, mask = sample["raw"], sample["labels"], sample.get("mask")
        if not os.path.exists(raw):
            raise ValueError(f"Raw dataset path {raw} does not exist")
        if ".zarray" not in os.listdir(raw):
            raise ValueError(f"Raw dataset path {raw} does not contain a zarr array")
        if not os.path.exists(labels):
            raise ValueError(f"Labels dataset path {labels} does not exist")
        if ".zarray" not in os.listdir(labels) and not glob.glob(os.path.join(labels, "**", ".zarray"), recursive=True):
            raise ValueError(f"Labels dataset prefix {labels} does not contain any array")
        if mask is not None and not os.path.exists(mask):
            raise ValueError(f"Mask dataset path {mask} does not exist")
    return config


def affinity_targets(labels, unlabelled, neighborhood, grow_steps=0, only_xy=True, clip=(0.05, 0.95)):
    """models/3d_affs/train.py:127-139 on the device (bsmi_train_affinity_targets): GrowBoundary(labels,
    mask=unlabelled, steps, only_xy) -> AddAffinities(neighborhood) -> BalanceLabels.
    labels: int64 CUDA (D, H, W), overwritten with the grown-boundary labels; unlabelled: uint8 CUDA (D, H, W) or None.
    Returns (gt_affs, affs_weights), float32 (n, D, H, W)."""
    if labels.dtype != torch.int64 or not labels.is_cuda or not labels.is_contiguous():
        raise ValueError("labels must be a contiguous int64 CUDA tensor")
    if unlabelled is not None and (unlabelled.dtype != torch.uint8 or unlabelled.shape != labels.shape or not unlabelled.is_contiguous()):
        raise ValueError("unlabelled must be a contiguous uint8 tensor of the labels' shape")
    n = len(neighborhood)
    nb = (C.c_int32 * (3 * n))(*[int(v) for off in neighborhood for v in off])
    affs = torch.empty((n,) + tuple(labels.shape), dtype=torch.float32, device=labels.device)
    weights = torch.empty_like(affs)
    _lib.check(_lib.lib.bsmi_train_affinity_targets(
        labels.device.index, C.c_void_p(labels.data_ptr()), C.c_void_p(unlabelled.data_ptr()) if unlabelled is not None else None,
        _lib.i64x3(labels.shape), nb, n, int(grow_steps), 1 if only_xy else 0, float(clip[0]), float(clip[1]),
        C.c_void_p(affs.data_ptr()), C.c_void_p(weights.data_ptr()), C.c_void_p(torch.cuda.current_stream(labels.device).cuda_stream)))
    return affs, weights


def lsd_targets(labels, roi_offset, roi_shape, sigma, voxel_size, downsample=1, unlabelled=None):
    """3-D local shape descriptors of one sample on the device (reference models/3d_mtlsd/train.py:134-141).
    labels: int64 CUDA (D, H, W) holding the crop with the window context; -> (gt_lsds, lsds_weights) float32 (10, d, h, w)."""
    if labels.dtype != torch.int64 or not labels.is_cuda or labels.dim() != 3 or not labels.is_contiguous():
        raise ValueError("labels must be a contiguous int64 CUDA tensor")
    if unlabelled is not None and (unlabelled.dtype != torch.uint8 or unlabelled.shape != labels.shape or not unlabelled.is_contiguous()):
        raise ValueError("unlabelled must be a contiguous uint8 tensor of the labels' shape")
    sig = [float(sigma)] * 3 if isinstance(sigma, (int, float)) else [float(v) for v in sigma]
    out = torch.empty((10,) + tuple(int(v) for v in roi_shape), dtype=torch.float32, device=labels.device)
    weights = torch.empty_like(out)
    _lib.check(_lib.lib.bsmi_train_lsd_targets(
        labels.device.index, C.c_void_p(labels.data_ptr()), C.c_void_p(unlabelled.data_ptr()) if unlabelled is not None else None,
        _lib.i64x3(labels.shape), _lib.i64x3(roi_offset), _lib.i64x3(roi_shape), (C.c_float * 3)(*sig),
        (C.c_float * 3)(*[float(v) for v in voxel_size]), int(downsample), C.c_void_p(out.data_ptr()), C.c_void_p(weights.data_ptr()),
        C.c_void_p(torch.cuda.current_stream(labels.device).cuda_stream)))
    return out, weights


class SampleSource:
    """Infinite iterator of reference-style batches from (raw, labels[, mask]) Zarr volumes."""

    def __init__(self, samples, input_shape, output_shape, neighborhood, device=0, seed=42, head="affs", grow_boundary=0,
                 lsd_sigma=None, lsd_downsample=1, voxel_size=(1, 1, 1)):
        self.samples = [(open_ds(s["raw"]), open_ds(s["labels"]), open_ds(s["mask"]) if s.get("mask") else None) for s in samples]
        self.inp, self.out = tuple(input_shape), tuple(output_shape)
        self.nhood = [list(map(int, o)) for o in neighborhood]
        self.rng = np.random.default_rng(seed)
        self.dev = torch.device("cuda", device)
        self.grow = int(grow_boundary)
        # head: "affs" (3d_affs), "lsds" (3d_lsd) or "mtlsd" (3d_mtlsd: both, models/3d_mtlsd/train.py:134-153)
        if head not in ("affs", "lsds", "mtlsd"):
            raise ValueError(f"unknown head {head!r}")
        self.head = head
        self.lsd_sigma, self.lsd_df = lsd_sigma, int(lsd_downsample)
        self.voxel_size = tuple(float(v) for v in voxel_size)
        if head != "affs" and lsd_sigma is None:
            raise ValueError("the LSD head needs net_config outputs.3d_lsds.sigma")

    def __iter__(self):
        return self

    def __next__(self):
        ctx = [(i - o) // 2 for i, o in zip(self.inp, self.out)]
        for _ in range(1000):
            raw_ds, lab_ds, mask_ds = self.samples[self.rng.integers(len(self.samples))]
            shape = lab_ds.shape[-3:]
            if any(s < o for s, o in zip(shape, self.out)):
                raise ValueError("labels volume smaller than the network's output block")
            off = [int(self.rng.integers(0, s - o + 1)) for s, o in zip(shape, self.out)]
            sl = tuple(slice(a, a + o) for a, o in zip(off, self.out))
            labels = lab_ds[sl].astype(np.int64)
            unl = (mask_ds[sl] > 0) if mask_ds is not None else (labels > 0)
            if unl.mean() < 0.05:  # gp.Reject(mask=unlabelled, min_masked=0.05)
                continue
            # raw with context; outside the volume: zeros (gp.Pad(raw, None))
            raw = np.zeros(self.inp, dtype=np.uint8)
            rshape = raw_ds.shape[-3:]
            src, dst = [], []
            for a, c, n, i in zip(off, ctx, rshape, self.inp):
                lo, hi = max(a - c, 0), min(a - c + i, n)
                src.append(slice(lo, hi))
                dst.append(slice(lo - (a - c), hi - (a - c)))
            raw[tuple(dst)] = raw_ds[tuple(src)]
            x = torch.from_numpy(raw).to(self.dev).float() * (1.0 / 255.0) * 2.0 - 1.0
            lab = torch.from_numpy(labels).to(self.dev)
            unl_dev = torch.from_numpy(unl.astype(np.uint8)).to(self.dev)
            batch = {"raw": x}
            if self.head != "affs":
                # descriptors see the labels 3 sigma beyond the output block (AddLocalShapeDescriptor grows its request by
                # that context; beyond the volume the padded labels are 0), snapped to the sub-sampling grid
                df = self.lsd_df
                sig = [float(self.lsd_sigma)] * 3 if isinstance(self.lsd_sigma, (int, float)) else list(self.lsd_sigma)
                cv = [int(-(-3.0 * s // v)) for s, v in zip(sig, self.voxel_size)]
                cv = [-(-c // df) * df for c in cv]
                big = np.zeros([o + 2 * c for o, c in zip(self.out, cv)], dtype=np.int64)
                bun = np.zeros(big.shape, dtype=np.uint8)
                src, dst = [], []
                for a, c, n, o in zip(off, cv, shape, self.out):
                    lo, hi = max(a - c, 0), min(a + o + c, n)
                    src.append(slice(lo, hi))
                    dst.append(slice(lo - (a - c), hi - (a - c)))
                big[tuple(dst)] = lab_ds[tuple(src)].astype(np.int64)
                bun[tuple(dst)] = (mask_ds[tuple(src)] > 0) if mask_ds is not None else (big[tuple(dst)] > 0)
                lsds, lw = lsd_targets(torch.from_numpy(big).to(self.dev), cv, self.out, sig, self.voxel_size, df,
                                       torch.from_numpy(bun).to(self.dev))
                batch.update(gt_lsds=lsds, lsds_weights=lw)
            if self.head != "lsds":
                affs, weights = affinity_targets(lab, unl_dev, self.nhood, self.grow, only_xy=True)
                batch.update(gt_affs=affs, affs_weights=weights)
            return batch
        raise RuntimeError("no training location with at least 5 % labelled voxels found")


class PrefetchSource:
    """Batches produced ahead of the training thread (reference training.py:107-114: `DataLoader(dataset, num_workers=8,
    persistent_workers=True, pin_memory=True)`): a producer thread walks `source` -- Zarr chunk decoding runs in libbsmi's
    native threads without the GIL, the target kernels on a side stream -- and keeps up to `depth` finished batches queued,
    so a 20 ms training step never waits for crops and targets.  One producer: the batches come in the source's own order
    (same seed, same sequence as the source used directly)."""

    def __init__(self, source, depth=4, device=0):
        import queue
        import threading
        self.source = source
        self.dev = torch.device("cuda", device)
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self.stream = torch.cuda.Stream(self.dev)
        self._stop = False
        self.thread = threading.Thread(target=self._run, name="bsmi-samples", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            with torch.cuda.stream(self.stream):
                for batch in self.source:
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                    while not self._stop:
                        try:
                            self.q.put((batch, ev), timeout=0.1)
                            break
                        except Exception:  # noqa: BLE001 - queue.Full: the trainer is busy, try again
                            continue
                    if self._stop:
                        return
        except BaseException as exc:  # noqa: BLE001 - handed to the training thread
            self.q.put(exc)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        batch, ev = item
        ev.synchronize()   # host-side: a stream parked behind a device-side wait slows the kernels that run meanwhile
        cur = torch.cuda.current_stream(self.dev)
        for t in batch.values():
            t.record_stream(cur)   # allocated on the producer's stream, used on the trainer's
        return batch

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except Exception:  # noqa: BLE001 - queue.Empty
            pass
        self.thread.join(timeout=5)


def make_sample_source(config, net_config, device=0, rank=0):
    """The built-in sample stream of `bs train` for this rank: seed 42 + rank, so that data-parallel ranks see different
    samples (with one seed for all, the averaged gradient would be the single-rank gradient computed N times)."""
    outs = net_config["outputs"]
    out3d, lsd3d = outs.get("3d_affs"), outs.get("3d_lsds")
    if set(outs) - {"3d_affs", "3d_lsds"} or not outs:
        raise NotImplementedError(f"the built-in sample source feeds the 3-D setups (3d_affs, 3d_lsd, 3d_mtlsd), not {sorted(outs)}")
    head = "mtlsd" if out3d and lsd3d else ("affs" if out3d else "lsds")
    nhood = out3d["neighborhood"][: int(out3d["dims"])] if out3d else [[-1, 0, 0], [0, -1, 0], [0, 0, -1]]
    return SampleSource(config["samples"], net_config["input_shape"], net_config["output_shape"], nhood, device=device,
                        seed=42 + int(rank), head=head, grow_boundary=int(out3d.get("grow_boundary", 0)) if out3d else 0,
                        lsd_sigma=lsd3d.get("sigma") if lsd3d else None, lsd_downsample=int(lsd3d.get("downsample", 1)) if lsd3d else 1,
                        voxel_size=config.get("voxel_size", (1, 1, 1)))


def default_init(net_config, seed=42):
    """torch's default Conv3d initialisation (kaiming_uniform(a=sqrt(5)), bias uniform(+-1/sqrt(fan_in))) for every
    parameter of the reference Model, keyed like its state_dict."""
    from .unet import HEAD_OF_OUTPUT
    g = torch.Generator().manual_seed(seed)
    nf, inc = int(net_config["num_fmaps"]), int(net_config["fmap_inc_factor"])
    dfs = net_config["downsample_factors"]
    nl = len(dfs) + 1
    ksd = net_config.get("kernel_size_down") or [[[3, 3, 3], [3, 3, 3]]] * nl
    ksu = net_config.get("kernel_size_up") or [[[3, 3, 3], [3, 3, 3]]] * (nl - 1)
    sd = {}

    def conv(key, cout, cin, k):
        fan_in = cin * int(np.prod(k))
        bound = 1.0 / np.sqrt(fan_in)
        sd[key + ".weight"] = ((torch.rand((cout, cin) + tuple(k), generator=g) * 2 - 1) * bound).numpy()
        sd[key + ".bias"] = ((torch.rand(cout, generator=g) * 2 - 1) * bound).numpy()

    def conv_pass(prefix, cin, cout, kernels):
        c = cin
        for i, k in enumerate(kernels):
            conv(f"{prefix}.conv_pass.{2 * i}", cout, c, k)
            c = cout
        conv(f"{prefix}.residual.0", cout, cin, (1, 1, 1))

    for lvl in range(nl):
        conv_pass(f"unet.l_conv.{lvl}", int(net_config["in_channels"]) if lvl == 0 else nf * inc ** (lvl - 1), nf * inc ** lvl, ksd[lvl])
    for lvl in range(nl - 1):
        conv_pass(f"unet.r_conv.0.{lvl}", nf * inc ** lvl + nf * inc ** (lvl + 1), nf * inc ** lvl, ksu[lvl])
    for name, val in net_config["outputs"].items():
        conv_pass(HEAD_OF_OUTPUT[name], nf, int(val["dims"]), [(1, 1, 1)])
    return sd


def latest_checkpoint(setup_dir):
    ckpts = glob.glob(os.path.join(setup_dir, "model_checkpoint_*.ckpt"))
    if not ckpts:
        return None, 0
    step = lambda p: int(re.findall(r"(\d+)", os.path.basename(p))[-1])  # noqa: E731
    best = max(ckpts, key=step)
    return best, step(best)


def run_training(config_file, device=0, batches=None, log=print):
    """`bs train <config>`: returns the number of iterations run.  Under torch.distributed every rank draws its own
    samples (seed 42 + rank: Lightning's seed_everything(42, workers=True) gives each rank's loader workers their own
    stream, training.py:128) and the gradients are averaged over the ranks."""
    from .unet import Model
    from .training import Trainer, fit, load_optimizer_state
    dist = torch.distributed
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    config = setup_train(config_file)
    setup_dir = config["setup_dir"]
    with open(os.path.join(setup_dir, "net_config.json")) as f:
        net_config = json.load(f)
    max_iterations = int(config["max_iterations"])
    model = Model(net_config, device=device, precision="f32")
    ckpt, done = latest_checkpoint(setup_dir)
    if ckpt:
        model.load_checkpoint(ckpt)
        log(f"resuming from {ckpt}")
    else:
        model.load_state_dict(default_init(net_config, seed=42))
    # `arithmetic`, `deterministic` (additions to the reference's train config): "split-bf16" (default) or "f32"; ordered
    # reductions so that two runs from the same state give the same bits (default false), see training.Trainer
    trainer = Trainer(model, net_config["input_shape"], arithmetic=config.get("arithmetic", "split-bf16"),
                      deterministic=bool(config.get("deterministic", False)))
    if ckpt and load_optimizer_state(trainer, ckpt):
        log(f"optimizer state restored (step {trainer.step_count()})")
    if batches is None:
        log("note: the reference's gunpowder augmentations are not part of this engine; samples are random crops")
        batches = make_sample_source(config, net_config, device, rank)
        depth = int(config.get("prefetch", 4))   # an addition to the reference's train config: batches kept ready (0: inline)
        if depth > 0:
            batches = PrefetchSource(batches, depth, device)
    try:
        n = fit(trainer, batches, max_iterations, int(config.get("save_checkpoints_every", 0)), setup_dir, log=log, start_iteration=done,
                save_snapshots_every=int(config.get("save_snapshots_every", 0)), voxel_size=config.get("voxel_size"))
    finally:
        if isinstance(batches, PrefetchSource):
            batches.close()
    trainer.close()
    return n
