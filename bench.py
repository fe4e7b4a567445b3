#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s of predict + segment on a synthetic 1024^3 uint8 volume
processed in 128^3 output blocks (BASELINE.json metric), one process per GPU.

A "step" is one 128^3 output block taken through the whole hot path with its input already
resident in HBM.  The timed region takes a box of `--steps` blocks per GPU (the ranks' boxes
stacked along z) through bootstrapper_amd.volume.VolumePipeline:
  reflect-padded (156,220,220) read -> 3-D U-Net -> uint8 affinities                    (predict)
  per block, on the 160^3 read box (context 16): seeded-watershed fragments, crop,
  26-connected relabel with global ids, node statistics; RAG edge scoring              (segment)
  global thresholded connected components at [0.2, 0.35, 0.5] -> LUT -> relabel        (stitch)
so what is timed ends in ONE consistent segmentation of the box per threshold.  Default
precision: split bf16 (bf16x3), the mode that meets the 1e-4 parity gate of the fp32 reference.
Exchange steps (N > 1): context margins of affinities / fragments between z-neighbours, scored
edges to rank 0, LUT broadcast; everything else is rank-local (weak scaling).

  python bench.py --gpus 1 --steps 64 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (implicit-GEMM
conv kernels, HIP-event timed inside the timed region), `predict_only` / `segment_only` (the two
halves of the timed region), `cpu_baseline` (the repo's CPU restatement timed on this node's host
cores; N=1 only) and `modes` (speed and max error of every precision mode on the same blocks).
"""
import argparse
import json
import os
import sys
import time

# one hardware queue per HIP stream of the pipeline (predict lanes + segmentation lanes): with
# the runtime default of 4, streams share queues and a long sequential segmentation kernel
# stalls the predict stream queued behind it.  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NET_CONFIG = {  # reference models/3d_affs/net_config.json
    "in_channels": 1, "num_fmaps": 12, "fmap_inc_factor": 5,
    "downsample_factors": [[1, 2, 2], [1, 2, 2], [1, 2, 2]],
    "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
    "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3,
    "outputs": {"3d_affs": {"dtype": "uint8", "dims": 6}},
}
OUT_BLOCK = (128, 128, 128)
CONTEXT = (14, 46, 46)          # (input - output) / 2 of the 3-D nets (reference predict.py:127-131)
SEG_CONTEXT = (16, 16, 16)      # block_size / 8 (reference post/watershed.py:79-83)
THRESHOLDS = [0.2, 0.35, 0.5]   # reference segment.py:17
FILTER_FRAGMENTS, REMOVE_DEBRIS = 0.1, 64   # reference segment.py:16 (`bs segment --ws` defaults)
BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, dense bf16 MFMA
# what the conv kernels are priced against, per precision mode.  The split mode spends three bf16 MFMAs per product of the
# algorithmic count, so its ceiling is a third of the dense bf16 peak.
MFMA_PEAK_TFLOPS = {"bf16": BF16_DENSE_PEAK_TFLOPS, "bf16x3": BF16_DENSE_PEAK_TFLOPS / 3.0, "f32": 157.3}
MFMA_PEAK_NOTE = {"bf16": "dense bf16 MFMA peak", "f32": "dense f32 MFMA peak",
                  "bf16x3": "dense bf16 MFMA peak / 3: each f32-accurate product is hi*hi + lo*hi + hi*lo on the bf16 MFMA"}
HBM_PEAK_BYTES = 8.0e12
# HBM-side bytes per conv launch cannot be counted inside this process: they come from the committed
# rocprofv3 PMC passes of the same kernels on the same block (tools/pmc_traffic.py; FETCH_SIZE and WRITE_SIZE
# in separate passes, KiB -> bytes, FETCH_SIZE doubled for gfx950 wide reads as the guide prescribes).
TRAFFIC_PROFILES = {"bf16x3": os.path.join(ROOT, "profiles", "r04_conv_traffic_pmc_bf16x3.json"),
                    "bf16": os.path.join(ROOT, "profiles", "r01_g_conv_traffic_pmc.json")}


def pmc_traffic(precision):
    path = TRAFFIC_PROFILES.get(precision)
    try:
        with open(path) as f:
            return float(json.load(f)["traffic_bytes_per_conv_launch"]), os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


SQ_PROFILES = {"bf16x3": os.path.join(ROOT, "profiles", "r04_conv_sq_pmc_bf16x3.json"),
               "bf16": os.path.join(ROOT, "profiles", "r01_g_conv_sq_pmc.json")}


TRAIN_PEAK_F32 = 3.0 / (2.0 / 157.3 + 3.0 / 2500.0)  # TFLOP/s: `--train-arithmetic f32` (forward, input gradients on the f32 pipe)
X3_PEAK = 2500.0 / 3.0


def pmc_mfma_busy(precision):
    """MFMA-busy fraction of the persistent implicit-GEMM launches (the dominant kernels), from the committed PMC
    passes (tools/run_pmc_passes.sh + tools/pmc_sq.py): SIMD cycles with the matrix pipe busy / SIMD cycles."""
    try:
        with open(SQ_PROFILES[precision]) as f:
            ks = json.load(f)["kernels"]
        rows = [r for k, r in ks.items() if "conv_igemm_sk_kernel" in k]
        tot = sum(r["chip_cycles_per_launch"] * r["launches"] for r in rows)
        return sum(r["mfma_busy"] * r["chip_cycles_per_launch"] * r["launches"] for r in rows) / tot
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota and by
    the per-GPU share of the box (16 host cores per GPU on the benchmark pool)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("BSMI_BENCH_CORES", "16"))))


def job_blocks_for(steps, world=1, max_edge=8):
    """(layers, blocks in y, blocks in x) with layers * y * x == steps: the box of blocks a rank takes; the ranks stack their
    boxes along z.  The whole job stays inside the volume where the factors allow (`max_edge` = 1024 / 128 blocks per axis,
    so at most max_edge // world layers per rank: 8 ranks x 64 blocks = one layer of 8 x 8 blocks each = the 1024^3 volume);
    among those the smallest cross-section y * x (a block can be segmented once the next layer of blocks is predicted)."""
    max_layers = max(1, max_edge // max(1, world))
    best = None
    for gx in range(1, steps + 1):
        if steps % gx:
            continue
        for gy in range(gx, steps // gx + 1):
            if (steps // gx) % gy:
                continue
            gz = steps // gx // gy
            fits = gz <= max_layers and gy <= max_edge
            key = (not fits, gy * gx, gy - gx)
            if best is None or key < best[0]:
                best = (key, (gz, gy, gx))
    return best[1]


def cpu_baseline(raw_blocks, affs_u8_host, seg_blocks):
    """CPU restatement (oracle/) timed on this node's host cores on a bounded sample of the same workload.

    predict: the torch-CPU fp32 network (oracle/unet_ref.py) on real (156,220,220) -> 128^3 blocks of the job, all cores.
    segment: the blockwise pipeline composed from the C restatement (oracle/blockwise_ref.py: fragments with context 16,
    clean-up, RAG scoring, connected components, relabel) on `seg_blocks` 128^3 blocks of the affinities the GPU
    predicted, blocks of a stage side by side on all cores."""
    from oracle import unet_ref as R
    from oracle.blockwise_ref import cpu_blockwise
    from bootstrapper_amd.synth import synthetic_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(NET_CONFIG, 0)
    cfg = R.default_cfg(12, 5)
    nvox = int(np.prod(OUT_BLOCK))
    R.predict_block(cfg, sd, np.ascontiguousarray(raw_blocks[0][:40, :116, :116]), ["affs_head"])  # thread pool and allocator warm-up
    outs = []
    t0 = time.perf_counter()
    for raw in raw_blocks:
        outs.append(R.predict_block(cfg, sd, raw, ["affs_head"])[0])
    t_pred = time.perf_counter() - t0
    pred_vox_s = len(raw_blocks) * nvox / t_pred

    t0 = time.perf_counter()
    frags, nodes, E, Sc, segs = cpu_blockwise(affs_u8_host, OUT_BLOCK, SEG_CONTEXT, 10, FILTER_FRAGMENTS, REMOVE_DEBRIS, THRESHOLDS, 256,
                                              workers=cores)
    t_seg = time.perf_counter() - t0
    seg_vox_s = affs_u8_host[0].size / t_seg
    both = 1.0 / (1.0 / pred_vox_s + 1.0 / seg_vox_s)
    return {
        "value": both / 1e6, "unit": "Mvoxels/s", "cores": cores, "cores_affinity": len(os.sched_getaffinity(0)), "kind": "port",
        "sample": (f"predict: torch-CPU fp32 restatement, {len(raw_blocks)} blocks (156,220,220)->128^3 of the job in {t_pred:.1f} s "
                   f"-> {pred_vox_s / 1e3:.2f} kvox/s; segment: blockwise pipeline of the C restatement (fragments with context 16, "
                   f"RAG scoring, connected components, relabel) on {seg_blocks} blocks of GPU-predicted affinities, {cores} cores, "
                   f"{len(nodes)} fragments, {len(E)} edges, {t_seg:.1f} s -> {seg_vox_s / 1e6:.2f} Mvox/s"),
        "predict_kvox_s": pred_vox_s / 1e3, "segment_Mvox_s": seg_vox_s / 1e6,
        "cores_note": "cores = threads used (torch intra-op threads of the predict restatement, pool threads of the segment one): the affinity mask, "
                      "capped by the cgroup CPU quota and by BSMI_BENCH_CORES (default 16 = one GPU's share of the node's host cores); cores_affinity = the uncapped mask",
    }, outs, frags, (E, Sc)


def same_partition(a, b):
    """Two label volumes describe the same fragments after an id remap: background on background, and the (a, b) id pairs
    that occur form a bijection."""
    a, b = a.ravel(), b.ravel()
    if not np.array_equal(a == 0, b == 0):
        return False
    pairs = np.unique(np.stack([a, b]), axis=1)
    return len(np.unique(pairs[0])) == pairs.shape[1] == len(np.unique(pairs[1]))


def check_fragments(gpu_frags, cpu_frags, sub, job):
    """The CPU restatement's fragments of the sampled sub-box against the GPU pipeline's, block by block, on the blocks whose
    read box (write box + context) is the same in both runs: every block but those on a face of the sub-box that lies inside
    the job.  -> (blocks compared, blocks whose fragments differ after an id remap)"""
    compared = bad = 0
    for z in range(sub[0]):
        for y in range(sub[1]):
            for x in range(sub[2]):
                b = (z, y, x)
                if any(b[d] + 1 == sub[d] and sub[d] != job[d] for d in range(3)):
                    continue
                sl = tuple(slice(b[d] * OUT_BLOCK[d], (b[d] + 1) * OUT_BLOCK[d]) for d in range(3))
                compared += 1
                bad += not same_partition(gpu_frags[sl], cpu_frags[sl])
    return compared, bad


def regrid_ids(ids, sub, job):
    """fragment ids of a run over the sub-box `sub` (blocks) -> the ids the same fragments carry in a run over the box `job` that
    starts at the same corner: id = block index (z-major in the run's own grid) * voxels per block + label."""
    nv = np.uint64(int(np.prod(OUT_BLOCK)))
    ids = np.asarray(ids, np.uint64)
    b, lab = (ids - np.uint64(1)) // nv, (ids - np.uint64(1)) % nv
    z, r = b // np.uint64(sub[1] * sub[2]), b % np.uint64(sub[1] * sub[2])
    y, x = r // np.uint64(sub[2]), r % np.uint64(sub[2])
    out = ((z * np.uint64(job[1]) + y) * np.uint64(job[2]) + x) * nv + lab + np.uint64(1)
    return np.where(ids > 0, out, np.uint64(0))


def check_edges(gpu_edges, gpu_scores, cpu_edges, cpu_scores, sub, job):
    """The scored edges (what decides the segmentations) of the GPU pipeline against the CPU restatement's, on the blocks whose
    scoring read the same fragments in both runs: a block's read box holds its 26 neighbours' fragments, so the block and all its
    neighbours inside the job must have had the same read box -- every block at least two away from a face of the sub-box that
    lies inside the job.  Bit for bit: same edges, same float32 scores (NaN = never merged).  -> (blocks compared, edges compared,
    blocks that differ)"""
    nv = int(np.prod(OUT_BLOCK))
    ce = regrid_ids(cpu_edges, sub, job).reshape(-1, 2)
    g_owner = (gpu_edges[:, 0] - np.uint64(1)) // np.uint64(nv)
    c_owner = (ce[:, 0] - np.uint64(1)) // np.uint64(nv)
    blocks = n_edges = bad = 0
    for z in range(sub[0]):
        for y in range(sub[1]):
            for x in range(sub[2]):
                b = (z, y, x)
                if any(b[d] + 2 >= sub[d] and sub[d] != job[d] for d in range(3)):
                    continue
                bid = (z * job[1] + y) * job[2] + x
                ge, gs = gpu_edges[g_owner == bid], gpu_scores[g_owner == bid]
                ee, es = ce[c_owner == bid], cpu_scores[c_owner == bid]
                go, eo = np.lexsort((ge[:, 1], ge[:, 0])), np.lexsort((ee[:, 1], ee[:, 0]))
                same = len(ge) == len(ee) and np.array_equal(ge[go], ee[eo]) and np.array_equal(gs[go].view(np.uint32), es[eo].view(np.uint32))
                blocks += 1
                n_edges += len(ee)
                bad += not same
    return blocks, n_edges, bad


def drivers_leg(raw_box, sd, precision, model=None, vol=None, origin=(0, 0, 0)):
    """`bs predict` + `bs segment --ws` (blockwise) as a user runs them, on an on-disk Zarr store holding the same box of
    blocks: checkpoint load, chunk decode / encode (Blosc lz4, the zarr default), file reads and writes included.  Outside
    the timed region; `raw_box`: uint8 host array of the job's output extent (the drivers reflect-pad at its faces)."""
    import json as _json
    import shutil
    import tempfile
    from bootstrapper_amd.predict import run_prediction
    from bootstrapper_amd.segment import run_segmentation
    from bootstrapper_amd.zarr_io import prepare_ds
    tmp = tempfile.mkdtemp(prefix="bsmi_bench_", dir=os.environ.get("BSMI_BENCH_TMP"))
    try:
        setup = os.path.join(tmp, "3d_affs")
        os.makedirs(setup)
        nc = dict(NET_CONFIG, input_shape=[o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT)], output_shape=list(OUT_BLOCK),
                  shape_increase=[0, 0, 0], inputs={"raw": {"dims": 1}}, outputs={"3d_affs": {"dtype": "uint8", "dims": 6}})
        with open(os.path.join(setup, "net_config.json"), "w") as f:
            _json.dump(nc, f)
        ckpt = os.path.join(setup, "model_checkpoint_1")
        torch.save({"model_state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, ckpt + ".ckpt")
        store = os.path.join(tmp, "vol.zarr")
        ds = prepare_ds(store + "/raw", raw_box.shape, offset=(0, 0, 0), voxel_size=(1, 1, 1), chunk_shape=OUT_BLOCK, dtype=np.uint8,
                        axis_names=["z", "y", "x"], units=["nm"] * 3)
        ds[:] = raw_box
        pred_toml = os.path.join(tmp, "pred.toml")
        with open(pred_toml, "w") as f:
            f.write(f'["01-3d_affs"]\nsetup_dir = "{setup}"\ninput_datasets = ["{store}/raw"]\ncheckpoint = "{ckpt}"\n'
                    f'output_datasets_prefix = "{store}/predictions"\nchain_str = ""\nnum_workers = 1\nnum_gpus = 1\n')
        seg_toml = os.path.join(tmp, "seg.toml")
        with open(seg_toml, "w") as f:
            f.write(f'affs_dataset = "{store}/predictions/1/3d_affs"\nfragments_dataset = "{store}/fragments"\n'
                    f'seg_dataset_prefix = "{store}/segmentations"\nblockwise = true\nblock_shape = {list(OUT_BLOCK)}\n'
                    f'context = {list(SEG_CONTEXT)}\n[db]\ndb_file = "{tmp}/rag.db"\n[ws_params]\nthresholds = {THRESHOLDS}\n'
                    f'min_seed_distance = 10\nfilter_fragments = {FILTER_FRAGMENTS}\nremove_debris = {REMOVE_DEBRIS}\n')
        t0 = time.perf_counter()
        run_prediction(pred_toml, "01", precision=precision)
        t_pred = time.perf_counter() - t0
        # what `bs predict` stored against the engine called directly on the same blocks (the command predicts while its
        # write-behind threads copy, encode and write the blocks before: nothing of that may show in the data)
        def check_predictions(model):
            vol = torch.from_numpy(raw_box).to(torch.device("cuda", model.device))   # the store's extent: the drivers reflect-pad at ITS faces
            origin = (0, 0, 0)
            from bootstrapper_amd.unet import extract_block_reflect
            from bootstrapper_amd.zarr_io import open_ds
            pds = open_ds(f"{store}/predictions/1/3d_affs")
            nb = [s // b for s, b in zip(raw_box.shape, OUT_BLOCK)]
            picks = sorted({(0, 0, 0), (nb[0] - 1, nb[1] - 1, nb[2] - 1), (nb[0] // 2, nb[1] // 2, nb[2] // 2), (0, nb[1] - 1, nb[2] // 2),
                            (nb[0] - 1, 0, nb[2] - 1), (nb[0] // 2, 0, 0)})
            in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
            differ = 0
            for b in picks:
                off = [origin[d] + b[d] * OUT_BLOCK[d] - CONTEXT[d] for d in range(3)]
                want = model.predict_u8(extract_block_reflect(vol, off, in_block))[0].cpu().numpy()
                got = pds[(slice(None),) + tuple(slice(b[d] * OUT_BLOCK[d], (b[d] + 1) * OUT_BLOCK[d]) for d in range(3))]
                differ += not np.array_equal(got, want)
            if differ:
                raise SystemExit(f"parity (drivers): {differ} of {len(picks)} sampled blocks of the prediction dataset differ from the engine's own prediction")
            return {"blocks_compared": len(picks), "blocks_differing": differ,
                    "what": "the prediction dataset `bs predict` wrote against the engine called on the same blocks: all six channels, bit for bit"}
        pred_check = check_predictions(model) if model is not None and model != "late" else None
        t0 = time.perf_counter()
        written = run_segmentation(seg_toml, "ws")
        t_seg = time.perf_counter() - t0
        # ... and the `filter` step of a bootstrap round (BASELINE config 4: predict -> segment -> filter): `bs refine` size filter
        # on the last threshold's segmentation (refine.py:131-257: object table, rule, relabelled copy of the dataset)
        import contextlib
        import io
        from bootstrapper_amd.refine import size_filter
        t0 = time.perf_counter()
        filter_error = None
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                filtered = size_filter(written[-1], min_size=500)
        except Exception as exc:  # noqa: BLE001 - reported in the line; the two commands the leg is about have run
            filtered, filter_error = None, f"{type(exc).__name__}: {exc}"
        t_filter = time.perf_counter() - t0
        if model == "late":  # a process of its own for the commands: its engine is built only now
            from bootstrapper_amd.unet import Model
            pred_check = check_predictions(Model(NET_CONFIG, device=torch.cuda.current_device(), precision=precision).load_state_dict(sd))

        def du(path):
            return sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(path) for f in fs)
        nvox = raw_box.size
        return {"what": "`bs predict` then `bs segment --ws` (blockwise, one worker) on an on-disk Zarr store of the same box of blocks: "
                        "checkpoint load and weight packing, Blosc-lz4 chunk decode / encode and file I/O included (outside the timed region); "
                        "then `bs refine` size filter on the last segmentation (filter_seconds, not in Mvoxels_per_s)",
                "blocks": int(nvox // int(np.prod(OUT_BLOCK))), "predict_seconds": t_pred, "segment_seconds": t_seg, "filter_seconds": t_filter,
                "Mvoxels_per_s": nvox / (t_pred + t_seg) / 1e6, "predict_Mvoxels_per_s": nvox / t_pred / 1e6,
                "segment_Mvoxels_per_s": nvox / t_seg / 1e6, "round_Mvoxels_per_s": nvox / (t_pred + t_seg + t_filter) / 1e6,
                "datasets_written": len(written) + 1 + (1 if filtered else 0), "filter_error": filter_error, "prediction_check": pred_check,
                "store_bytes": du(store), "tmp_dir": os.path.dirname(tmp) or tmp}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


TRAIN_SHAPE = (32, 196, 196)


def train_run(dev, local_rank, rank, world, arithmetic, steps, warmup, deterministic=False):
    """`steps` training steps of the full 3d_affs net on the reference's training block -> (seconds, last loss, first-step gradients, forward FLOPs)"""
    import torch.distributed as dist
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from bootstrapper_amd.synth import synthetic_state_dict
    shape = TRAIN_SHAPE

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
    model = Model(NET_CONFIG, device=local_rank, precision="f32").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
    tr = Trainer(model, shape, arithmetic=arithmetic, deterministic=deterministic)
    g = torch.Generator(device=dev).manual_seed(rank)
    out = (6,) + tuple(tr.out_shape)
    batch = {"raw": torch.rand(shape, generator=g, device=dev) * 2 - 1,
             "gt_affs": (torch.rand(out, generator=g, device=dev) > 0.5).float(),
             "affs_weights": torch.rand(out, generator=g, device=dev)}
    tr.forward_backward(batch["raw"], [batch["gt_affs"]], [batch["affs_weights"]])
    grads = tr.grads.clone()  # of the first step, before any update: what the two arithmetics are compared on
    for _ in range(warmup):
        tr.training_step(batch)
    barrier()
    t0 = time.perf_counter()
    loss = 0.0
    for _ in range(steps):
        loss = tr.training_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fwd = model.flops(shape)
    tr.close()
    return dt, loss, grads, fwd


def train_leg(dev, local_rank):
    """The training step beside the headline (default run, one GPU): 10 steps in the default split-bf16 arithmetic, 4 in exact
    f32, and how far the first-step gradients of the two are apart."""
    NS = 20  # steps timed after 5 untimed ones (the first steps grow scratch buffers and ramp the clocks: 10 after 2 read 1.2 ms high)
    dt, loss, grads, fwd = train_run(dev, local_rank, 0, 1, "split-bf16", NS, 5)
    dt32, loss32, grads32, _ = train_run(dev, local_rank, 0, 1, "f32", 4, 1)
    # Trainer(deterministic=True): ordered folds instead of float atomics; twice, to report that the two runs end on the same bits
    dtd, lossd, gradsd, _ = train_run(dev, local_rank, 0, 1, "split-bf16", NS, 5, deterministic=True)
    _, lossd2, gradsd2, _ = train_run(dev, local_rank, 0, 1, "split-bf16", NS, 5, deterministic=True)
    step_flops = 3.0 * fwd
    return {"what": "3d_affs U-Net (94.7M params) forward + WeightedMSELoss + backward + Adam, block (32,196,196) -> (6,4,104,104), batch 1; "
                    "`bench.py --mode train` is the full line",
            "split-bf16": {"ms_per_step": dt / NS * 1e3, "samples_per_s": NS / dt, "tflops": step_flops * NS / dt / 1e12,
                           "frac_of_split_peak": step_flops * NS / dt / 1e12 / X3_PEAK, "last_loss": loss},
            "f32": {"ms_per_step": dt32 / 4 * 1e3, "samples_per_s": 4 / dt32, "tflops": step_flops * 4 / dt32 / 1e12, "last_loss": loss32},
            "split-bf16 deterministic": {"ms_per_step": dtd / NS * 1e3, "last_loss": lossd,
                                         "two_runs_bit_equal": bool(torch.equal(gradsd, gradsd2) and lossd == lossd2),
                                         "max_gradient_difference_to_default_rel": float((gradsd - grads).abs().max() / grads.abs().max())},
            "max_gradient_difference_rel": float((grads - grads32).abs().max() / grads32.abs().max())}


def train_main(args):
    """Secondary benchmark (`--mode train`): samples/s of the fp32 training step of the full 3d_affs net on the
    reference's training block (32,196,196) -> (4,104,104), batch 1 per GPU, gradients averaged over the ranks."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    from bootstrapper_amd.synth import synthetic_state_dict
    def run(arithmetic, steps, warmup):
        return train_run(dev, local_rank, rank, world, arithmetic, steps, warmup, deterministic=args.train_deterministic)

    dt, loss, grads, fwd = run(args.train_arithmetic, args.steps, args.warmup)
    step_flops = 3.0 * fwd  # forward + input gradients + weight gradients
    achieved = step_flops * args.steps / dt / 1e12
    split = args.train_arithmetic == "split-bf16"
    # split-bf16: every convolution of the step spends three bf16 MFMAs per product (2500 / 3 TFLOP/s); f32: forward and
    # input gradients on the f32 matrix pipe (157.3), the weight gradients as split-bf16 all the same
    peak = X3_PEAK if split else TRAIN_PEAK_F32
    other = None
    if rank == 0 and world == 1 and not args.no_modes:  # the other arithmetic beside it: speed, and how far the two sets of gradients are apart
        o_arith = "f32" if split else "split-bf16"
        o_dt, o_loss, o_grads, _ = run(o_arith, max(2, args.steps // 2), 1)
        err = float((grads - o_grads).abs().max() / o_grads.abs().max())
        other = {"arithmetic": o_arith, "ms_per_step": o_dt / max(2, args.steps // 2) * 1e3, "last_loss": o_loss,
                 "max_gradient_difference_rel": err, "note": "first-step gradients of the two arithmetics, largest entry difference / largest entry"}
    out_json = {"metric": "training samples/s, 3d_affs U-Net (fp32 tensors, loss, gradients, Adam; convolutions in %s arithmetic), "
                          "(32,196,196) blocks, batch 1 per GPU" % args.train_arithmetic, "value": world * args.steps / dt,
                "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3" if split else "f32", "data": "synthetic",
                "config": {"workload": "3d_affs U-Net (94.7M params) forward + WeightedMSELoss + backward + Adam, block (32,196,196) -> "
                                       "(6,4,104,104), gradient groups all-reduced over RCCL during the backward pass for N > 1",
                           "arithmetic": args.train_arithmetic, "last_loss": loss},
                "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                             "kernel": "whole step: conv_igemm (forward, input gradients) + wgrad_x3_kernel (weight gradients)",
                             "peak_note": "split-bf16: 2500 / 3 TFLOP/s (three bf16 MFMAs per product); f32: 3 / (2 / 157.3 + 3 / 2500)",
                             "algorithmic_tflop_per_step": step_flops / 1e12},
                "other_arithmetic": other}
    if rank == 0:
        print(json.dumps(out_json), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def whole_volume_leg(model, vol, args, dev, local_rank, rank, world, obj_group, use_dist, seg_kw, barrier, flops_block, peak, check):
    """BASELINE.json's metric on its own configuration: predict + segment of the WHOLE synthetic volume (1024^3 = 8 x 8 x 8 blocks
    of 128^3), resident in HBM, the block layers dealt to the ranks (8 / N layers of 8 x 8 blocks each: at N = 1 one GPU takes all
    512 blocks, at N = 8 every rank one layer -- the same volume at every N, i.e. the strong-scaling reading of the metric beside the
    weak-scaling `value`).  Timed like the headline: barrier + synchronize on both sides, max over ranks."""
    import torch.distributed as dist
    from bootstrapper_amd.volume import VolumePipeline
    edge = args.volume // OUT_BLOCK[0]
    job = (edge // world, edge, edge)
    nblocks = job[0] * job[1] * job[2]
    pipe = VolumePipeline(model, OUT_BLOCK, CONTEXT, job, SEG_CONTEXT, THRESHOLDS, n_lanes=args.seg_lanes, device=local_rank,
                          rank=rank, world=world, overlap=args.overlap, obj_group=obj_group, **seg_kw)
    pipe.seg.overlap_lanes = args.overlap_lanes
    pipe.seg.prime()
    barrier()
    t0 = time.perf_counter()
    pipe.run(vol)
    barrier()
    dt = time.perf_counter() - t0
    t_pred = pipe.t_predict
    mine = {"rank": rank, "blocks": nblocks, "predict_seconds": t_pred, **{k + "_seconds": v for k, v in pipe.seg.timers.items()},
            "fragments": int(sum(int(n) for n in pipe.seg.block_nums)), "scored_edges": int(len(pipe.seg.rag_scores))}
    if use_dist:
        t = torch.tensor([dt, t_pred], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_pred = (float(v) for v in t.tolist())
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=obj_group)
    else:
        parts = [mine]
    nvox = args.volume ** 3
    res = {"what": f"the whole synthetic {args.volume}^3 volume = {edge}x{edge}x{edge} blocks of 128^3 through the same pipeline (predict every block "
                   f"into the resident slabs, fragments with context, RAG scoring, global connected components, LUT, relabel), {edge // world} block "
                   f"layer(s) of {edge}x{edge} blocks per GPU on {world} GPU(s); outside the weak-scaling timed region, timed the same way",
           "blocks": nblocks * world, "n_gpus": world, "seconds": dt, "Mvoxels_per_s": nvox / dt / 1e6, "ms_per_block_per_gpu": dt / nblocks * 1e3,
           "predict_seconds": t_pred, "segment_seconds": dt - t_pred,
           "predict_mfma_frac": flops_block * nblocks / t_pred / 1e12 / peak,
           "fragments": int(len(pipe.seg.nodes)), "scored_edges": int(sum(p["scored_edges"] for p in parts)),
           "segments": [int(len(np.unique(c))) for c in pipe.seg.luts], "per_rank": parts}
    if check:
        # spot check against the CPU restatement: the 3 x 3 x 3 blocks at the volume's corner through the C pipeline; fragments
        # of the 8 blocks and scored edges of the one block whose inputs are the same in both runs
        from oracle.blockwise_ref import cpu_blockwise
        sub = tuple(min(3, j) for j in job)
        a = pipe.seg.interior(pipe.seg.affs)[:, :sub[0] * 128, :sub[1] * 128, :sub[2] * 128].contiguous().cpu().numpy()
        t1 = time.perf_counter()
        cf, _, ce, cs, _ = cpu_blockwise(a, OUT_BLOCK, SEG_CONTEXT, 10, FILTER_FRAGMENTS, REMOVE_DEBRIS, THRESHOLDS, 256, workers=host_cores())
        t_cpu = time.perf_counter() - t1
        gf = pipe.seg.interior(pipe.seg.frags)[:sub[0] * 128, :sub[1] * 128, :sub[2] * 128].cpu().numpy().view(np.uint64)
        compared, bad = check_fragments(gf, cf, sub, job)
        exact = bool(np.array_equal(gf[:128 * (sub[0] - 1), :128 * (sub[1] - 1), :128 * (sub[2] - 1)],
                                    regrid_ids(cf, sub, job)[:128 * (sub[0] - 1), :128 * (sub[1] - 1), :128 * (sub[2] - 1)]))
        eb, ne, ebad = check_edges(pipe.seg.rag_edges, pipe.seg.rag_scores, ce, cs, sub, job)
        res["parity_check"] = {"what": f"CPU restatement on the {sub[0]}x{sub[1]}x{sub[2]} blocks at the volume's corner ({t_cpu:.1f} s): fragments equal after an id "
                                       "remap on the blocks with the same read box, their ids equal outright, scored edges equal bit for bit where both runs scored the same fragments",
                               "blocks_compared": compared, "blocks_differing": bad, "ids_equal": exact,
                               "edge_blocks_compared": eb, "edges_compared": ne, "edge_blocks_differing": ebad}
        if bad or ebad or not exact:
            raise SystemExit(f"parity (whole volume): fragments of {bad}/{compared} blocks, edges of {ebad}/{eb} blocks differ from the CPU restatement's (ids equal: {exact})")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="predict", choices=["predict", "train", "train-leg", "drivers-leg"],
                    help="predict = the headline predict + segment benchmark; train = training-step samples/s (secondary)")
    ap.add_argument("--train-deterministic", action="store_true", help="--mode train: Trainer(deterministic=True), ordered reductions")
    ap.add_argument("--train-arithmetic", default="split-bf16", choices=["split-bf16", "f32"],
                    help="--mode train: split-bf16 (default; convolutions as bf16 hi + lo products) or f32 (exact f32 MFMA)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64,
                    help="128^3 blocks per GPU in the timed region: a box of blocks of the 1024^3 volume, the ranks' boxes stacked along z")
    ap.add_argument("--warmup", type=int, default=2, help="blocks taken through the whole pipeline before the timed region")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "f32"],
                    help="bf16x3 (default): split bf16, within the 1e-4 parity gate; bf16: throughput mode (4e-3); f32: exact f32 MFMA")
    ap.add_argument("--volume", type=int, default=1024, help="edge of the synthetic cubic volume")
    ap.add_argument("--seg-lanes", type=int, default=20, help="blocks of a segmentation stage in flight side by side")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl (= RCCL, the real thing) or gloo (rehearsal on a box with fewer GPUs)")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank, but with the process group of an N > 1 run: RCCL initialised, the gloo side group for object collectives, "
                         "the barrier and the max-over-ranks reduction executed (a rehearsal of the distributed plumbing on a one-GPU box)")
    ap.add_argument("--overlap", action="store_true",
                    help="start a block's segmentation as soon as the blocks it reads are predicted (default: stage by stage; +3 %% end to end, conv launches 8 %% slower)")
    ap.add_argument("--overlap-lanes", type=int, default=3, help="--overlap: lanes that take tasks while blocks are still being predicted")
    ap.add_argument("--no-segment", action="store_true", help="predict only (diagnostic; not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pred-lanes", type=int, default=2,
                    help="engines whose forward passes overlap, block k on engine k mod K (measured on 64 blocks: 1 lane 96.4, 2 lanes 103.6, 3 lanes 96.7 Mvoxels/s)")
    ap.add_argument("--no-modes", action="store_true", help="skip the per-precision predict lines (speed and error of f32 / bf16x3 / bf16)")
    ap.add_argument("--no-train", action="store_true", help="skip the `train` leg (ms per training step in both arithmetics)")
    ap.add_argument("--no-drivers", action="store_true", help="skip the `drivers` leg (bs predict + bs segment on an on-disk Zarr store of the whole volume)")
    ap.add_argument("--drivers-box", action="store_true", help="`drivers` leg on the box of --steps blocks instead of the whole volume")
    ap.add_argument("--no-whole-volume", action="store_true", help="skip the `whole_volume` leg (the whole 1024^3 volume, resident, over all ranks)")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="per-launch HIP-event timing (the roofline figures) on every Nth block of the timed region")
    ap.add_argument("--cpu-predict-blocks", type=int, default=2)
    ap.add_argument("--cpu-segment-blocks", type=int, default=16)
    args = ap.parse_args()
    if args.mode == "drivers-leg":  # the default run's `drivers` object on the whole volume, printed by a child process of that run
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
        from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        box = synthetic_volume((args.volume,) * 3, seed=0, device=torch.device("cuda", lr)).cpu().numpy()
        torch.cuda.empty_cache()
        print(json.dumps(drivers_leg(box, synthetic_state_dict(NET_CONFIG, 0), args.precision, "late")), flush=True)
        return
    if args.mode == "train-leg":  # the default run's `train` object, printed by a child process of that run
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        print(json.dumps(train_leg(torch.device("cuda", lr), lr)), flush=True)
        return
    if args.mode == "train":
        if "--steps" not in sys.argv:
            args.steps = 20
        if "--warmup" not in sys.argv:
            args.warmup = 5  # the first steps grow scratch buffers and ramp the clocks
        return train_main(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # The two side legs of the default one-GPU run -- the drivers on the whole volume as an on-disk store, the training step -- run
    # FIRST, each in a process of its own, while this process has not touched the GPU yet: that is how a user runs those commands
    # (one process, the card to itself).  Measured in round 4: inside this process after its resident pipelines `bs segment` spent 1.3 s
    # allocating its slab (0.03 s alone) and the training step's second stream gained nothing (21.6 against 18.5 ms); in a child
    # beside this process's 23 idle hardware queues `bs segment` took 3.9 s instead of 2.0.
    side_legs = {}
    if rank == 0 and world == 1 and not args.force_dist and not os.environ.get("BSMI_BENCH_LEGS_INPROC"):
        import shutil
        import subprocess
        tmp_root = os.environ.get("BSMI_BENCH_TMP") or __import__("tempfile").gettempdir()

        def child(mode, timeout):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", mode, "--precision", args.precision, "--volume", str(args.volume)],
                                   capture_output=True, text=True, timeout=timeout, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not lines:
                    raise RuntimeError(f"exit code {r.returncode}: {r.stderr[-800:]}")
                d = json.loads(lines[-1])
                d["process"] = "a child of the bench run, before the run touched the GPU"
                return d
            except Exception as exc:  # noqa: BLE001 - a secondary leg: its failure is reported in the line, the headline stands
                return {"error": f"{type(exc).__name__}: {exc}"}
        whole_fits = shutil.disk_usage(tmp_root).free > 14 * args.volume ** 3 and not args.drivers_box
        if not args.no_drivers and not args.no_segment and whole_fits:
            side_legs["drivers"] = child("drivers-leg", 1500)
        if not args.no_train:
            side_legs["train"] = child("train-leg", 900)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
    if args.backend == "gloo":  # rehearsal on a box with fewer GPUs than ranks: the ranks share the GPUs there are
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    obj_group = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29555")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            # pickled objects (edge lists to rank 0, LUTs back) travel through a gloo group beside the RCCL one
            obj_group = dist.new_group(backend="gloo")
        else:
            dist.init_process_group("gloo")
        # the collectives of the volume pipeline once before anything is timed: object gather / broadcast on the side group,
        # a device all-reduce on the default one
        parts = [None] * world
        dist.all_gather_object(parts, {"rank": rank, "device": local_rank}, group=obj_group)
        hello = [parts if rank == 0 else None]
        dist.broadcast_object_list(hello, src=0, group=obj_group)
        assert [p["rank"] for p in hello[0]] == list(range(world))
        one = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        assert int(one.item()) == world

    from bootstrapper_amd.unet import Model, extract_block_reflect
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from bootstrapper_amd.volume import VolumePipeline

    sd = synthetic_state_dict(NET_CONFIG, 0)
    model = Model(NET_CONFIG, device=local_rank, precision=args.precision).load_state_dict(sd)
    # predict lanes: engines with the same weights whose forward passes overlap (VolumePipeline); `model` is lane 0
    engines = [model] + [Model(NET_CONFIG, device=local_rank, precision=args.precision).load_state_dict(sd) for _ in range(max(1, args.pred_lanes) - 1)]
    in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
    flops_block = model.flops(in_block)
    nvox_block = int(np.prod(OUT_BLOCK))

    vol_shape = (args.volume,) * 3
    vol = synthetic_volume(vol_shape, seed=0, device=dev)  # the input store: every rank reads its blocks (+ halo) from it
    job = job_blocks_for(args.steps, world)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up: `warmup` blocks through every stage (kernel images, workspaces of the block shapes, process-group channels)
    seg_kw = dict(min_seed_distance=10, filter_fragments=FILTER_FRAGMENTS, remove_debris=REMOVE_DEBRIS)
    # (the timed pipeline's slabs are allocated first, so that nothing but the barrier lies between the warm-up and the timed region)
    pipe = VolumePipeline(engines, OUT_BLOCK, CONTEXT, job, SEG_CONTEXT, THRESHOLDS, n_lanes=args.seg_lanes, device=local_rank,
                          rank=rank, world=world, segment=not args.no_segment, overlap=args.overlap, obj_group=obj_group, **seg_kw)
    warm = VolumePipeline(engines, OUT_BLOCK, CONTEXT, (max(len(engines), args.warmup), 1, 1), SEG_CONTEXT, THRESHOLDS, n_lanes=args.seg_lanes,
                          device=local_rank, rank=rank, world=world, segment=not args.no_segment, obj_group=obj_group, **seg_kw)
    if not args.no_segment:
        pipe.seg.overlap_lanes = args.overlap_lanes
    warm.run(vol)
    del warm
    if not args.no_segment:
        pipe.seg.prime()   # the slab-sized reductions / relabel kernels once, on the empty slab
    for m in engines:
        m.profile(max(1, args.profile_every))
        m.profile_totals(reset=True)
        m.profile_executed(reset=True)
    barrier()
    t0 = time.perf_counter()
    segs = pipe.run(vol)
    barrier()
    dt = time.perf_counter() - t0
    totals, executed_flops = None, 0.0
    for m in engines:  # the profiled launches of every predict lane together
        t = m.profile_totals(reset=True)
        totals = t if totals is None else {k: tuple(a + b for a, b in zip(totals[k], t[k])) for k in t}
        executed_flops += m.profile_executed(reset=True)
        m.profile(False)
    t_pred, t_seg = pipe.t_predict, 0.0
    single_lane = None
    if len(engines) > 1:
        # the same launches with the card to themselves (outside the timed region): eight blocks on lane 0 alone, every launch timed
        from bootstrapper_amd.unet import extract_block_reflect
        in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
        model.profile(1)
        model.profile_totals(reset=True)
        model.profile_executed(reset=True)
        with torch.cuda.stream(pipe.pred_stream):
            for i in range(8):
                model.predict_u8(extract_block_reflect(vol, [pipe.origin[0] - CONTEXT[0], pipe.origin[1] - CONTEXT[1], pipe.origin[2] - CONTEXT[2] + 128 * (i % 2)], in_block))
        torch.cuda.synchronize(dev)
        t1l = model.profile_totals(reset=True)
        ex1 = model.profile_executed(reset=True)
        model.profile(False)
        ms1, fl1, n1 = t1l["conv"]
        if ms1 > 0:
            dense = BF16_DENSE_PEAK_TFLOPS if args.precision != "f32" else MFMA_PEAK_TFLOPS["f32"]
            single_lane = {"achieved": fl1 / (ms1 * 1e-3) / 1e12, "frac": fl1 / (ms1 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
                           "avg_launch_ms": ms1 / max(n1, 1), "launches": int(n1), "executed_mfma_frac": ex1 / (ms1 * 1e-3) / 1e12 / dense,
                           "what": "eight blocks on one lane alone after the timed region: the per-launch figures without another forward pass beside them"}
    if not args.no_segment:
        # the segmentation half on its own (outside the timed region: there it overlaps the predict stream): the same
        # slab of affinities again, from fragments to the relabelled volumes; must reproduce the timed run's result
        first = segs.clone()
        barrier()
        t1 = time.perf_counter()
        again = pipe.seg.run()
        barrier()
        t_seg = time.perf_counter() - t1
        if not torch.equal(first, again):
            raise SystemExit("the segmentation-only pass does not reproduce the segmentation of the timed run")
        del first, again

    if use_dist:
        t = torch.tensor([dt, t_pred, t_seg], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_pred, t_seg = (float(v) for v in t.tolist())

    conv_ms, conv_flops, conv_launches = totals["conv"]
    peak = MFMA_PEAK_TFLOPS[args.precision]
    traffic, traffic_src = pmc_traffic(args.precision)
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # K predict lanes: a launch shares the card with the launches of the other lanes, so its own duration says what ONE lane gets.
    # How many forward passes were in flight on average = (all engines' profiled launch time, scaled to every block) / predict time
    unet_ms = sum(v[0] for v in totals.values())
    per_lane = [len(range(lane, args.steps, len(engines))) for lane in range(len(engines))]   # blocks of each lane
    profiled_blocks = max(1, sum(-(-n // max(1, args.profile_every)) for n in per_lane))       # every lane profiles every Nth of ITS passes
    in_flight = (unet_ms / profiled_blocks * args.steps) / (t_pred * 1e3) if t_pred > 0 else 1.0
    nvox = world * args.steps * nvox_block
    value = nvox / dt / 1e6
    out = {
        "metric": "Mvoxels/s predict+segment, 1024^3 vol in 128^3 blocks" if not args.no_segment else "Mvoxels/s predict only (diagnostic)",
        "value": value, "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"box of {job[0] * world}x{job[1]}x{job[2]} 128^3 output blocks of a synthetic {args.volume}^3 uint8 volume "
                               "(156,220,220 reads, reflect padded): 3d_affs U-Net (94.7M params, seeded random weights) -> uint8 affinities -> "
                               "per block: xy seeded watershed on the 160^3 read box (context 16), fragment filter 0.1 / debris 64, crop, 26-connected relabel, node statistics, "
                               "RAG edge scoring (mean affinity, 256-bin queue) -> global connected components at [0.2,0.35,0.5] -> LUT -> "
                               "relabel: one consistent segmentation per threshold",
                   "blocks_per_gpu": args.steps, "job_blocks_per_gpu": list(job),
                   "parallelism": f"slabs of block layers over {world} GPU(s); face exchange of affinities and fragments, edges to rank 0, LUT broadcast",
                   "seg_lanes": len(pipe.seg.lanes), "overlap": bool(args.overlap)},
        "predict_only": {"Mvoxels_per_s": nvox / t_pred / 1e6, "seconds": t_pred, "mfma_frac": flops_block * args.steps / t_pred / 1e12 / peak,
                         "note": "start of the timed region to the last predicted block" + ("; the segmentation lanes already run meanwhile" if args.overlap else "")},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "peak_note": MFMA_PEAK_NOTE[args.precision],
                     "traffic": traffic, "traffic_unit": "bytes per launch (memory side of L2, Infinity-Cache hits included)",
                     "traffic_source": traffic_src,
                     "kernel": "every convolution stage of the U-Net: bsmi::conv_igemm_kernel / conv_igemm_sk_kernel launches, conv_h16_kernel for four of the six stages with at most 64 output channels and, for the seven stages in Winograd F(4x4,3x3) form, wino4_in_kernel + the batched conv_igemm_sk_kernel launch (36 batches) + wino4_out_kernel (their time is inside the stage's; FLOPs are the direct convolution's: see executed_mfma_frac)",
                     "launches": int(conv_launches), "avg_launch_ms": conv_ms / max(conv_launches, 1),
                     "timed": f"HIP events around every launch of every {max(1, args.profile_every)}. block of the timed region",
                     # the multiplies the matrix pipe was actually given (tile padding, all 16 / 36 batches of a Winograd stage, three bf16
                     # products per product of the split mode) against the DENSE peak of the pipe: a Winograd stage computes its layer
                     # with 2.25x (F(2x2)) or 4x (F(4x4)) fewer multiplies than `achieved` counts, so `frac` can pass what the pipe could
                     # do on the direct form; this figure cannot
                     "frac_note": ("frac = algorithmic TFLOP/s of the DIRECT convolution / (dense bf16 MFMA peak / 3), the ceiling of the split-bf16 direct form; "
                                   "the Winograd F(4x4) stages compute their layers with a quarter of the multiplies, so frac may pass 1: the pipe's own "
                                   "utilisation is executed_mfma_frac, and frac_of_dense_peak prices the same algorithmic rate against the hardware's 2500 TFLOP/s"
                                   if args.precision == "bf16x3" else "frac = algorithmic TFLOP/s of the direct convolution / dense MFMA peak of the dtype"),
                     "frac_of_dense_peak": achieved / (BF16_DENSE_PEAK_TFLOPS if args.precision != "f32" else MFMA_PEAK_TFLOPS["f32"]),
                     "executed_tflops": executed_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                     "executed_mfma_frac": (executed_flops / (conv_ms * 1e-3) / 1e12 / (BF16_DENSE_PEAK_TFLOPS if args.precision != "f32" else MFMA_PEAK_TFLOPS["f32"])) if conv_ms > 0 else 0.0,
                     "mfma_busy_pmc": pmc_mfma_busy(args.precision),
                     "algorithmic_tflop_per_block": flops_block / 1e12,
                     "single_lane": single_lane,
                     "predict_lanes": len(engines), "passes_in_flight": in_flight,
                     "chip_achieved": achieved * in_flight, "chip_frac": achieved * in_flight / peak,
                     "chip_note": ("with K > 1 predict lanes a launch shares the card with launches of the other lanes' forward passes: achieved / frac / "
                                   "avg_launch_ms / executed_mfma_frac are per launch (what rocprofv3 sees), chip_* = per launch x passes_in_flight "
                                   "(passes_in_flight = all lanes' launch time / predict wall time)"),
                     "other_unet_ms_per_block": sum(totals[k][0] for k in ("input", "pool", "upsample", "head")) / max(args.steps, 1)},
    }
    if not args.no_segment:
        # two figures: the stage inside the timed region (what the pipeline pays, what the CPU ratio is taken from) and the
        # same stage repeated afterwards; they agree once the slab-sized torch paths have been used before the timed region
        # (SlabSegmenter.prime).  With --overlap the stage is not separable and only the repeated pass is given.
        t_in = max(dt - t_pred, 1e-9) if not args.overlap else t_seg
        seg_vox_s = nvox / t_in
        out["segment_only"] = {"note": "seconds: end of the predict stage to the end of the timed region (fragments -> relabelled volumes); "
                                       "repeat_pass: the same stage again on the same affinities after the timed region",
                               "Mvoxels_per_s": seg_vox_s / 1e6, "seconds": t_in, "bytes_per_voxel": 38,
                               "hbm_frac": 38.0 * seg_vox_s / world / HBM_PEAK_BYTES,
                               "repeat_pass": {"seconds": t_seg, "Mvoxels_per_s": nvox / t_seg / 1e6},
                               "fragments": int(len(pipe.seg.nodes)), "segments": [int(len(np.unique(c))) for c in pipe.seg.luts]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nb = max(1, min(args.cpu_predict_blocks, args.steps))
        raws = [extract_block_reflect(vol, [o + lo - c for o, lo, c in zip(pipe.origin, b, CONTEXT)], in_block)
                for b, _ in pipe.seg.boxes[:nb]]
        # the CPU sample: whole block layers of the job where a layer fits the budget (the sample then shares every face but
        # the last layer's with the job, and its fragments can be checked against the GPU's block by block)
        if job[1] * job[2] <= args.cpu_segment_blocks:
            sj = (min(job[0], max(1, args.cpu_segment_blocks // (job[1] * job[2]))), job[1], job[2])
        else:
            sj = job_blocks_for(max(1, min(args.cpu_segment_blocks, args.steps)))
            sj = tuple(min(a, b) for a, b in zip(sj, job))
        affs = pipe.seg.interior(pipe.seg.affs)[:, :sj[0] * 128, :sj[1] * 128, :sj[2] * 128].contiguous().cpu().numpy()
        out["cpu_baseline"], cpu_outs, cpu_frags, cpu_edges = cpu_baseline([r.cpu().numpy() for r in raws], affs, sj[0] * sj[1] * sj[2])
        if not args.no_segment:
            out["segment_only"]["gpu_over_cpu"] = out["segment_only"]["Mvoxels_per_s"] / out["cpu_baseline"]["segment_Mvox_s"]
            out["segment_only"]["gpu_over_cpu_note"] = (f"against {out['cpu_baseline']['cores']} host cores, this GPU's share of the node; at equal share "
                                                        "the ratio does not grow with the GPU count")
            # the CPU restatement is also the checker: its fragments on the sampled blocks against what the GPU pipeline produced
            gpu_frags = pipe.seg.interior(pipe.seg.frags)[:sj[0] * 128, :sj[1] * 128, :sj[2] * 128].cpu().numpy().view(np.uint64)
            compared, bad = check_fragments(gpu_frags, cpu_frags, sj, job)
            out["cpu_baseline"]["parity_check"] = {"blocks_compared": compared, "blocks_differing": bad,
                                                   "what": "fragments of the GPU pipeline vs the CPU restatement on the same affinities, equal after an id remap"}
            if bad:
                raise SystemExit(f"parity: the fragments of {bad} of {compared} sampled blocks differ from the CPU restatement's")
            # ... and its scored edges (the host merge loops' output, which the fragments alone would not catch) where both runs
            # scored the same fragments
            eb, ne, ebad = check_edges(pipe.seg.rag_edges, pipe.seg.rag_scores, cpu_edges[0], cpu_edges[1], sj, job)
            out["cpu_baseline"]["parity_check"].update({"edge_blocks_compared": eb, "edges_compared": ne, "edge_blocks_differing": ebad,
                                                        "edges_what": "scored RAG edges of those blocks whose 26 neighbours had the same read box in both runs: same edges, float32 scores bit for bit"})
            if ebad:
                raise SystemExit(f"parity: the scored edges of {ebad} of {eb} sampled blocks differ from the CPU restatement's")
        out["predict_only"]["gpu_over_cpu"] = out["predict_only"]["Mvoxels_per_s"] * 1e3 / out["cpu_baseline"]["predict_kvox_s"]
        if not args.no_modes:
            # speed / accuracy of every precision mode on the same blocks, errors against the CPU fp32 restatement
            modes = {}
            for prec in ("f32", "bf16x3", "bf16"):
                model.set_precision(prec)
                err = 0.0
                for raw, ref in zip(raws, cpu_outs):
                    _, f32 = model.predict_u8(raw, want_f32=True)
                    err = max(err, float((f32[0].cpu() - torch.from_numpy(ref)).abs().max()))
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                reps = 4
                for i in range(reps):
                    model.predict_u8(raws[i % len(raws)])
                torch.cuda.synchronize(dev)
                ms = (time.perf_counter() - t1) / reps * 1e3
                modes[prec] = {"ms_per_block": ms, "Mvoxels_per_s": nvox_block / ms / 1e3, "tflops": flops_block / ms / 1e9,
                               "max_abs_err_vs_cpu_fp32": err, "within_1e-4": err < 1e-4}
            model.set_precision(args.precision)
            out["modes"] = modes
    # ---- the metric's own configuration: the WHOLE volume (1024^3 = 8 x 8 x 8 blocks), resident, over all ranks -----------------
    per_rank = {"rank": rank, "predict_seconds": pipe.t_predict, **{k + "_seconds": v for k, v in pipe.seg.timers.items()}} if not args.no_segment else None
    if use_dist and per_rank is not None:
        parts = [None] * world
        dist.all_gather_object(parts, per_rank, group=obj_group)
        out["per_rank"] = parts
    elif per_rank is not None:
        out["per_rank"] = [per_rank]
    pipe_origin = pipe.origin
    edge = args.volume // OUT_BLOCK[0]
    if not args.no_segment and not args.no_whole_volume and edge % world == 0 and (args.steps, job) != (edge ** 3 // world, (edge // world, edge, edge)):
        del pipe, segs
        pipe = segs = None
        torch.cuda.empty_cache()
        try:
            out["whole_volume"] = whole_volume_leg(engines, vol, args, dev, local_rank, rank, world, obj_group, use_dist, seg_kw, barrier,
                                                   flops_block, peak, check=(rank == 0 and world == 1 and not args.no_cpu_baseline))
            out["config"]["whole_volume"] = out["whole_volume"]["what"]
        except Exception as exc:  # noqa: BLE001 - the headline above stands; with several ranks a failure of one rank ends the job
            if world > 1:
                raise
            out["whole_volume"] = {"error": f"{type(exc).__name__}: {exc}"}
    elif not args.no_segment and not args.no_whole_volume and edge % world == 0:
        out["whole_volume"] = {"what": "the timed region above IS the whole volume", "Mvoxels_per_s": value, "seconds": dt}
    if "drivers" in side_legs:
        out["drivers"] = side_legs["drivers"]
        out["drivers"].setdefault("Mvoxels_per_s", 0.0)
        out["drivers"]["whole_volume"] = True
        resident = out.get("whole_volume", {}).get("Mvoxels_per_s")
        out["drivers"]["resident_Mvoxels_per_s"] = resident
        out["drivers"]["frac_of_resident"] = out["drivers"]["Mvoxels_per_s"] / resident if resident else None
    elif rank == 0 and world == 1 and not args.no_drivers and not args.no_segment:
        # in this process: the box of `--steps` blocks when the volume's datasets would not fit the temporary directory (or
        # BSMI_BENCH_LEGS_INPROC / --force-dist)
        import shutil
        tmp_root = os.environ.get("BSMI_BENCH_TMP") or __import__("tempfile").gettempdir()
        whole = shutil.disk_usage(tmp_root).free > 14 * args.volume ** 3 and not args.drivers_box
        if whole:
            box = vol.cpu().numpy()
        else:
            ext = tuple(j * b for j, b in zip(job, OUT_BLOCK))
            box = vol[tuple(slice(o, o + e) for o, e in zip(pipe_origin, ext))].cpu().numpy()
        pipe = segs = None
        torch.cuda.empty_cache()
        try:
            out["drivers"] = drivers_leg(box, sd, args.precision, model, vol, (0, 0, 0) if whole else pipe_origin)
        except Exception as exc:  # noqa: BLE001
            import traceback
            out["drivers"] = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:], "Mvoxels_per_s": 0.0}
        out["drivers"]["whole_volume"] = bool(whole)
        resident = out.get("whole_volume", {}).get("Mvoxels_per_s") if whole else value
        out["drivers"]["resident_Mvoxels_per_s"] = resident
        out["drivers"]["frac_of_resident"] = out["drivers"]["Mvoxels_per_s"] / resident if resident else None
    if "train" in side_legs:
        out["train"] = side_legs["train"]
    elif rank == 0 and world == 1 and not args.no_train:
        pipe = segs = vol = None
        torch.cuda.empty_cache()
        try:
            out["train"] = train_leg(dev, local_rank)
        except Exception as exc:  # noqa: BLE001 - a secondary leg: its failure is reported in the line, the headline stands
            out["train"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
