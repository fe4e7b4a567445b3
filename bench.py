#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s of predict + segment on a synthetic 1024^3 uint8 volume
processed in 128^3 output blocks (BASELINE.json metric), one process per GPU.

A "step" is one 128^3 output block taken through the whole hot path with its input already
resident in HBM: reflect-padded (156,220,220) read -> 3-D U-Net (bf16 MFMA) -> uint8
affinities -> seeded-watershed fragments -> mean-affinity agglomeration at thresholds
[0.2, 0.35, 0.5].  Blocks are independent: with N ranks every rank takes its own K blocks
(weak scaling, no data-path collective); the only collectives are the timing barrier and
the max-over-ranks reduction.

  python bench.py --gpus 1 --steps 256 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (implicit-GEMM
conv kernels, HIP-event timed inside the timed region) and `cpu_baseline` (the repo's CPU
restatement timed on this node's host cores; N=1 only).
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
THRESHOLDS = [0.2, 0.35, 0.5]   # reference segment.py:17
BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, dense bf16 MFMA
# HBM-side bytes per conv launch cannot be counted inside this process: they come from the committed
# rocprofv3 PMC passes of the same kernels on the same block (tools/pmc_traffic.py; FETCH_SIZE and WRITE_SIZE
# in separate passes, KiB -> bytes, FETCH_SIZE doubled for gfx950 wide reads as the guide prescribes).
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r01_g_conv_traffic_pmc.json")


def pmc_traffic(precision):
    if precision != "bf16":
        return None, None
    try:
        with open(TRAFFIC_PROFILE) as f:
            return float(json.load(f)["traffic_bytes_per_conv_launch"]), os.path.relpath(TRAFFIC_PROFILE, ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


SQ_PROFILE = os.path.join(ROOT, "profiles", "r01_g_conv_sq_pmc.json")


def pmc_mfma_busy(precision):
    """MFMA-busy fraction of the persistent implicit-GEMM launches (the dominant kernels), from the committed PMC
    passes (tools/run_pmc_passes.sh + tools/pmc_sq.py): SIMD cycles with the matrix pipe busy / SIMD cycles."""
    if precision != "bf16":
        return None
    try:
        with open(SQ_PROFILE) as f:
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


def cpu_baseline(model_flops_per_voxel, affs_u8_host):
    """CPU restatement (oracle/) timed on this node's host cores on a bounded sample.

    predict: torch-CPU fp32 network on one (32,196,196)->(4,104,104) block (1.53 TFLOP, the
    reference's training block shape), all cores; converted to 128^3-block voxels/s through the
    measured FLOP/s (the small block has a worse halo ratio than the benchmark's).
    segment: C restatement of ws.py fragments + specified mean-affinity agglomeration on
    (32,128,128) slabs of the affinities the GPU predicted, one slab per core concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import unet_ref as R
    from oracle import seg_ref as S
    from bootstrapper_amd.synth import synthetic_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(NET_CONFIG, 0)
    cfg = R.default_cfg(12, 5)
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, size=(32, 196, 196), dtype=np.uint8)
    # flops of the sample block, same accounting as the device planner (algorithmic)
    from bootstrapper_amd.unet import Model
    m = Model(NET_CONFIG)
    sample_flops = m.flops(raw.shape)
    R.predict_block(cfg, sd, raw, ["affs_head"])  # thread pool and allocator warm-up
    n_pred = 8
    t0 = time.perf_counter()
    for _ in range(n_pred):
        R.predict_block(cfg, sd, raw, ["affs_head"])
    t_pred = time.perf_counter() - t0
    cpu_flops = n_pred * sample_flops / t_pred
    pred_vox_s = cpu_flops / model_flops_per_voxel

    slabs = [np.ascontiguousarray(affs_u8_host[:, z:z + 32]) for z in range(0, 128, 32)]
    work = [slabs[i % len(slabs)] for i in range(cores)]

    a_slab_vox = slabs[0][0].size
    seg_budget = 6.0  # seconds of wall time: every core keeps segmenting slabs until then

    def seg_worker(a):
        n, t_end = 0, time.perf_counter() + seg_budget
        while time.perf_counter() < t_end:
            frags, _ = S.ws_fragments_u8(a, True, 10)
            S.agglomerate_mean_u8(a, frags, THRESHOLDS)
            n += a[0].size
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        nvox = sum(ex.map(seg_worker, work))
    t_seg = time.perf_counter() - t0
    seg_vox_s = nvox / t_seg
    both = 1.0 / (1.0 / pred_vox_s + 1.0 / seg_vox_s)
    return {
        "value": both / 1e6, "unit": "Mvoxels/s", "cores": cores, "kind": "port",
        "sample": (f"predict: torch-CPU fp32 restatement, {n_pred} blocks (32,196,196)->(4,104,104), {t_pred:.1f} s, "
                   f"{cpu_flops / 1e9:.0f} GFLOP/s -> {pred_vox_s / 1e3:.2f} kvox/s at 128^3 blocks; "
                   f"segment: C restatement on (32,128,128) slabs of GPU-predicted affinities, {cores} cores side by side, "
                   f"{nvox / a_slab_vox:.0f} slabs in {t_seg:.1f} s -> {seg_vox_s / 1e6:.2f} Mvox/s"),
        "predict_kvox_s": pred_vox_s / 1e3, "segment_Mvox_s": seg_vox_s / 1e6,
    }


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
    shape = (32, 196, 196)
    model = Model(NET_CONFIG, device=local_rank, precision="f32").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
    tr = Trainer(model, shape)
    g = torch.Generator(device=dev).manual_seed(rank)
    out = (6,) + tuple(tr.out_shape)
    batch = {"raw": torch.rand(shape, generator=g, device=dev) * 2 - 1,
             "gt_affs": (torch.rand(out, generator=g, device=dev) > 0.5).float(),
             "affs_weights": torch.rand(out, generator=g, device=dev)}

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        tr.training_step(batch)
    barrier()
    t0 = time.perf_counter()
    loss = 0.0
    for _ in range(args.steps):
        loss = tr.training_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fwd = model.flops(shape)
    step_flops = 3.0 * fwd  # forward + input gradients + weight gradients
    achieved = step_flops * args.steps / dt / 1e12
    out_json = {"metric": "training samples/s, 3d_affs U-Net fp32, (32,196,196) blocks, batch 1 per GPU", "value": world * args.steps / dt,
                "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "3d_affs U-Net (94.7M params) forward + WeightedMSELoss + backward + Adam, fp32, block (32,196,196) -> "
                                       "(6,4,104,104), flat-gradient all-reduce over RCCL for N > 1", "last_loss": loss},
                "roofline": {"bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3, "traffic": None,
                             "kernel": "whole step: conv_igemm (forward, input gradients) + wgrad_kernel, f32 MFMA",
                             "algorithmic_tflop_per_step": step_flops / 1e12}}
    if rank == 0:
        print(json.dumps(out_json), flush=True)
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="predict", choices=["predict", "train"],
                    help="predict = the headline predict + segment benchmark; train = training-step samples/s (secondary)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256,
                    help="blocks per GPU in the timed region (the 1024^3 volume is 512 blocks; the segmentation of the last\n"
                         "blocks drains after the last predict, ~0.1 s once per run, so short runs under-report)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--volume", type=int, default=1024, help="edge of the synthetic cubic volume")
    ap.add_argument("--seg-lanes", type=int, default=16)
    ap.add_argument("--pred-lanes", type=int, default=1, help="U-Net replicas / predict streams per GPU")
    ap.add_argument("--seg-cus", type=int, default=0, help="CUs reserved for the segmentation lanes (0: shared CUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl (= RCCL, the real thing) or gloo (rehearsal on a box with fewer GPUs)")
    ap.add_argument("--no-segment", action="store_true", help="predict only (diagnostic; not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seg-stages", default="ws,agg", help="diagnostic: which segmentation stages the lanes run (ws, agg, or none: lane events only)")
    ap.add_argument("--seg-burst", type=int, default=16,
                    help="launch the segmentation of this many blocks together, one per lane, once the last of them is predicted\n"
                         "(0: block by block).  The lanes then share the chip with the predict stream a fifth of the time\n"
                         "instead of always (more than 20 lanes run out of hardware queues)")
    args = ap.parse_args()
    if args.mode == "train":
        if "--steps" not in sys.argv:
            args.steps = 10
        return train_main(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
    if args.backend == "gloo":  # rehearsal on a box with fewer GPUs than ranks: the ranks share the GPUs there are
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from bootstrapper_amd.pipeline import BlockPipeline, block_grid

    sd = synthetic_state_dict(NET_CONFIG, 0)
    models = [Model(NET_CONFIG, device=local_rank, precision=args.precision).load_state_dict(sd)
              for _ in range(max(1, args.pred_lanes))]
    model = models[0]
    in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
    flops_block = model.flops(in_block)
    nvox_block = int(np.prod(OUT_BLOCK))

    vol_shape = (args.volume,) * 3
    vol = synthetic_volume(vol_shape, seed=0, device=dev)  # every rank holds the same volume in HBM
    grid = block_grid(vol_shape, OUT_BLOCK)
    # interleaved block -> rank map (reference predict.py:46-49: worker_id % num_gpus)
    n_warm = max(args.warmup, max(1, args.pred_lanes))
    mine = [grid[(rank + i * world) % len(grid)] for i in range(n_warm + args.steps)]

    pipe = BlockPipeline(model, OUT_BLOCK, CONTEXT, THRESHOLDS, n_seg_lanes=args.seg_lanes,
                         segment=not args.no_segment, device=local_rank, models=models, seg_cus=args.seg_cus,
                         seg_stages=tuple(args.seg_stages.split(",")), seg_burst=args.seg_burst)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warmup (also allocates every workspace and the profiling events)
    for m in models:
        m.profile(True)
    pipe.run(vol, mine[:n_warm])
    pipe.finish()
    for m in models:
        m.profile_totals(reset=True)
    barrier()
    t0 = time.perf_counter()
    pipe.run(vol, mine[n_warm:])
    pipe.finish()
    barrier()
    dt = time.perf_counter() - t0
    totals = None
    for m in models:
        t = m.profile_totals(reset=True)
        m.profile(False)
        totals = t if totals is None else {k: tuple(a + b for a, b in zip(totals[k], t[k])) for k in t}

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    conv_ms, conv_flops, conv_launches = totals["conv"]
    traffic, traffic_src = pmc_traffic(args.precision)
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    value = world * args.steps * nvox_block / dt / 1e6
    out = {
        "metric": "Mvoxels/s predict+segment, 1024^3 vol in 128^3 blocks" if not args.no_segment else "Mvoxels/s predict only (diagnostic)",
        "value": value, "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"synthetic {args.volume}^3 uint8 volume, 128^3 output blocks (156,220,220 reads, reflect padded), "
                               "3d_affs U-Net (94.7M params, seeded random weights) + xy seeded watershed + mean-affinity "
                               "agglomeration at [0.2,0.35,0.5]",
                   "blocks_per_gpu": args.steps, "parallelism": f"blocks interleaved over {world} GPU(s), no collectives",
                   "seg_lanes": args.seg_lanes, "pred_lanes": len(models), "seg_cus": pipe.seg_cus, "seg_burst": pipe.seg_burst},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS if args.precision == "bf16" else 157.3,
                     "unit": "TFLOP/s", "frac": achieved / (BF16_DENSE_PEAK_TFLOPS if args.precision == "bf16" else 157.3),
                     "traffic": traffic, "traffic_unit": "bytes per launch (memory side of L2, Infinity-Cache hits included)",
                     "traffic_source": traffic_src,
                     "mfma_busy": pmc_mfma_busy(args.precision), "mfma_busy_source": os.path.relpath(SQ_PROFILE, ROOT),
                     "kernel": "bsmi::conv_igemm_kernel / conv_igemm_sk_kernel / first_pass_kernel (all convolution launches of the U-Net)",
                     "launches": int(conv_launches), "avg_launch_ms": conv_ms / max(conv_launches, 1),
                     "algorithmic_tflop_per_block": flops_block / 1e12,
                     "other_unet_ms_per_block": sum(totals[k][0] for k in ("input", "pool", "upsample", "head")) / max(args.steps, 1)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # affinities of one block for the CPU segment sample
        raw = torch.empty(0)
        from bootstrapper_amd.unet import extract_block_reflect
        raw = extract_block_reflect(vol, [o - c for o, c in zip(mine[0], CONTEXT)], in_block)
        affs = model.predict_u8(raw)[0][:3].cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(flops_block / nvox_block, affs)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
