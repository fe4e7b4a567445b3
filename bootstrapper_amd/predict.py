"""`bs predict`: blockwise affinity prediction of a Zarr volume on MI355X GPUs.

Behavioural mirror of /root/reference/bootstrapper/predict.py:
  :52-212  get_pred_config   block ROIs from net_config.json, output dataset names/shapes/chunks
  :22-49   predict_blockwise one worker per GPU, worker_id % num_gpus, retries, failure accounting
  :215-240 run_prediction    setup selection by id prefix/suffix/full name
and of the worker models/3d_affs/predict.py:59-162 (u8 -> [-1,1], reflect padding outside the
dataset, one model(input) per block, x255 -> uint8 store clipped to the dataset ROI).
The daisy TCP scheduler is replaced by a static interleave of the block list over the GPUs
(blocks are independent: read_write_conflict=False, predict.py:37); on each GPU the blocks
stream through `Model.predict_u8` while a write-behind pool copies, encodes and writes the finished blocks.
"""
import json
import os

import numpy as np

from .segment import load_toml
from .zarr_io import open_ds, prepare_ds

from .blockwise import MAX_RETRIES  # reference predict.py:38


def block_rois(net_config, voxel_size):
    """predict.py:114-131: block input/output shapes (voxels) and context, read/write ROI (world units)."""
    inc = net_config["shape_increase"]
    in_shape = [a + b for a, b in zip(inc, net_config["input_shape"])]
    out_shape = [a + b for a, b in zip(inc, net_config["output_shape"])]
    if len(in_shape) == 2:  # predict.py:118-124: z axis of the 2-D ("3ch") setups: adj_slices sections in, one out
        in_shape = [int(net_config["adj_slices"]), *in_shape]
        out_shape = [1, *out_shape]
    in_size = [s * v for s, v in zip(in_shape, voxel_size)]
    out_size = [s * v for s, v in zip(out_shape, voxel_size)]
    if any((a - b) % 2 for a, b in zip(in_size, out_size)):
        raise ValueError("input and output size must differ by an even amount")
    context = [(a - b) // 2 for a, b in zip(in_size, out_size)]
    return dict(input_shape=in_shape, output_shape=out_shape, context=context,
                read_roi=([-c for c in context], in_size), write_roi=([0, 0, 0], out_size))


def output_dataset_names(checkpoint, output_datasets_prefix, net_config, chain_str=""):
    """predict.py:143-155: '<prefix>/<iteration>[--from--<chain>]/<output name>'."""
    iteration = checkpoint.split("_")[-1]
    names = []
    for name in net_config["outputs"]:
        sub = f"{iteration}/{name}" if chain_str == "" else f"{iteration}--from--{chain_str}/{name}"
        names.append(os.path.join(output_datasets_prefix, sub))
    return names


def get_pred_config(config_file, setup_id, **kwargs):
    config = load_toml(config_file)[setup_id]
    for k, v in kwargs.items():
        if v is not None:
            config[k] = v
    setup_dir = config["setup_dir"]
    checkpoint = config["checkpoint"]
    if not os.path.exists(checkpoint) and not os.path.exists(checkpoint + ".ckpt"):
        raise ValueError(f"Checkpoint {checkpoint} does not exist!")  # no network: nothing can be downloaded
    with open(os.path.join(setup_dir, "net_config.json")) as f:
        net_config = json.load(f)
    input_datasets = config["input_datasets"]
    assert len(input_datasets) == len(net_config["inputs"]), (
        f"number of input datasets ({len(input_datasets)}) does not match number of network inputs "
        f"({net_config['inputs']})")
    in_ds = open_ds(input_datasets[0])
    voxel_size = in_ds.voxel_size
    rois = block_rois(net_config, voxel_size)

    def coords(v):
        return None if v is None else [int(x) for x in (v.split() if isinstance(v, str) else v)]

    roi_offset, roi_shape = coords(config.get("roi_offset")), coords(config.get("roi_shape"))
    if roi_offset is None:
        roi_offset, roi_shape = list(in_ds.roi[0]), list(in_ds.roi[1])
    outputs = output_dataset_names(checkpoint, config["output_datasets_prefix"], net_config, config.get("chain_str", ""))
    extra = {"pred_lanes": int(config["pred_lanes"])} if "pred_lanes" in config else {}  # an addition to the reference's keys (predict_blocks)
    return dict(setup_dir=setup_dir, checkpoint=checkpoint, net_config=net_config, input_datasets=input_datasets,
                output_datasets=outputs, output_roi=(roi_offset, roi_shape), voxel_size=list(voxel_size),
                num_workers=config.get("num_workers", 1), num_gpus=config.get("num_gpus", 1), **extra, **rois)


def prepare_outputs(cfg, in_ds):
    """predict.py:141-178: uint8 datasets (dims, *roi_shape) chunked by the output block."""
    vs = cfg["voxel_size"]
    off, shape = cfg["output_roi"]
    axes = in_ds.axis_names if "c^" in in_ds.axis_names else ["c^"] + in_ds.axis_names
    out = []
    for path, (name, val) in zip(cfg["output_datasets"], cfg["net_config"]["outputs"].items()):
        out.append(prepare_ds(path, shape=(val["dims"], *[s // v for s, v in zip(shape, vs)]), offset=off,
                              voxel_size=vs, axis_names=axes, units=in_ds.units,
                              chunk_shape=(val["dims"], *cfg["output_shape"]), dtype=np.dtype(val["dtype"])))
    return out


SECTIONS_PER_LAUNCH = 16  # 2-D setups: sections predicted by one pass of the (unit-depth) network


def launch_shapes(cfg):
    """(input, output) block shape of one engine launch.  3-D setups: the reference's block.  2-D setups: the
    reference predicts one section per block from adj_slices neighbours (models/2d_mtlsd/predict.py:84-91); every
    kernel of such a net has unit depth, so SECTIONS_PER_LAUNCH output sections go through the network as one
    (adj_slices, D, H, W) stack with identical results."""
    in_shape, out_shape = list(cfg["input_shape"]), list(cfg["output_shape"])
    if len(cfg["net_config"]["downsample_factors"][0]) == 2:
        out_shape[0] = SECTIONS_PER_LAUNCH
        in_shape[0] = SECTIONS_PER_LAUNCH + int(cfg["net_config"]["adj_slices"]) - 1
    return in_shape, out_shape


def enumerate_blocks(cfg):
    """Write-ROI origins (voxels, relative to the output ROI) covering it with fit='overhang'."""
    _, shape = cfg["output_roi"]
    nvox = [s // v for s, v in zip(shape, cfg["voxel_size"])]
    ob = launch_shapes(cfg)[1] if "net_config" in cfg else cfg["output_shape"]
    return [(z, y, x) for z in range(0, nvox[0], ob[0]) for y in range(0, nvox[1], ob[1]) for x in range(0, nvox[2], ob[2])]


def rank_blocks(blocks, rank, world):
    """This worker's share of the z-major block list: a contiguous run (the reference's daisy server hands blocks out
    one by one, predict.py:46-49; a contiguous run lets a worker read only the slab of the input it needs)."""
    q, r = divmod(len(blocks), world)
    start = rank * q + min(rank, r)
    return blocks[start:start + q + (1 if rank < r else 0)]


def predict_blocks(cfg, rank=0, world=1, device=None, precision="bf16x3"):
    """Worker body (one per GPU) -> TaskState of its blocks.  precision: bf16x3 (default; within 1e-4 of the reference's
    fp32 forward), f32, or bf16 (throughput mode, 4e-3)."""
    import torch
    from .blockwise import run_blocks
    from .unet import Model, extract_block_reflect
    device = rank % max(1, torch.cuda.device_count()) if device is None else device
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    import concurrent.futures as cf
    import threading
    from . import _trace
    # The checkpoint (379 MB for 3d_affs: torch.load, weight packing, Winograd weight transforms, upload: 0.7 s) is loaded on a
    # thread of its own while this one opens the datasets and starts the input read; the first block waits for both.
    def load_model():
        torch.cuda.set_device(device)
        with _trace.span("predict: checkpoint -> packed weights on the device"):
            return Model(cfg["net_config"], device=device, precision=precision).load_checkpoint(cfg["checkpoint"])
    side = cf.ThreadPoolExecutor(max_workers=3, thread_name_prefix="bsmi-load")
    model_future = side.submit(load_model)
    in_ds = open_ds(cfg["input_datasets"][0])
    outs = [open_ds(p, "r+") for p in cfg["output_datasets"]]
    vs = cfg["voxel_size"]
    roi_off, roi_shape = cfg["output_roi"]
    roi_vox = [s // v for s, v in zip(roi_shape, vs)]
    # dataset-relative voxel origin of the output ROI
    org = [(o - d) // v for o, d, v in zip(roi_off, in_ds.offset, vs)]
    in_shape, out_shape = launch_shapes(cfg)
    ctx = [(a - b) // 2 for a, b in zip(in_shape, out_shape)]
    mine = rank_blocks(enumerate_blocks(cfg), rank, world)
    if not mine:
        from .blockwise import TaskState
        model_future.result()
        side.shutdown()
        return TaskState("PredictBlockwiseTask", 0)
    # The slab of the input this worker's blocks read (with context), clipped to the dataset: z only -- y and x stay
    # whole, so the reflect padding about the dataset faces (gp.Pad on the array source) is the padding about the faces
    # of what is resident.  Every input channel becomes a (D, H, W) volume in HBM, in the order of net_config["inputs"]
    # (second-stage setups read several prediction datasets: models/3d_affs_from_2d_mtlsd/predict.py:80-81,139-142).
    nz = in_ds.shape[-3]
    z_need = (org[0] + min(b[0] for b in mine) - ctx[0], org[0] + max(b[0] for b in mine) + out_shape[0] + ctx[0])
    lo, hi = z_need
    if hi > nz:  # reads beyond the last section mirror back to section 2 (nz - 1) - i
        lo = min(lo, 2 * (nz - 1) - (hi - 1))
    if z_need[0] < 0:  # reads before the first section mirror to section -i
        hi = max(hi, 1 - z_need[0])
    whole = nz <= 2 * (in_shape[0] + 1)  # a dataset this thin may be mirrored more than once: keep all of it
    z_lo, z_hi = (0, nz) if whole else (max(0, lo), min(nz, hi))
    # The slab streams in behind the first blocks: runs of sections are decoded by the library's threads into page-locked
    # memory (`read_into`) and copied up on a side stream, in z order; a block starts once the sections it reads -- mirror
    # images included -- are resident (`loaded`: the slab is resident up to this section).
    vols, plans = [], []
    for path in cfg["input_datasets"]:
        ds = open_ds(path)
        lead = ds.shape[:-3]
        nlead = int(np.prod(lead)) if lead else 1
        if len(lead) > 1:
            raise ValueError(f"{path}: input datasets have at most one channel axis")
        t = torch.empty((nlead, z_hi - z_lo) + tuple(ds.shape[-2:]), dtype=torch.uint8 if ds.dtype == np.uint8 else torch.from_numpy(np.zeros(0, ds.dtype)).dtype,
                        device=dev)
        plans.append((ds, t, bool(lead)))
        vols += [t[c] for c in range(nlead)]
    loaded = {"z": z_lo, "error": None}
    loaded_cv = threading.Condition()

    def load_inputs():
        try:
            torch.cuda.set_device(device)
            from .volume import io_stream
            st = io_stream(dev)
            run = max(1, min(out_shape[0], (256 << 20) // max(1, int(np.prod(vols[0].shape[1:])) * len(vols) * vols[0].element_size())))
            bufs = [torch.empty((t.shape[0], run) + tuple(t.shape[2:]), dtype=t.dtype, pin_memory=True) for _, t, _ in plans]
            for za in range(z_lo, z_hi, run):
                zb = min(z_hi, za + run)
                for (ds, t, has_lead), buf in zip(plans, bufs):
                    host = buf[:, :zb - za]
                    with _trace.span("predict: input sections decoded", True):
                        if has_lead:
                            ds.read_into((slice(None), slice(za, zb)), host.numpy())
                        else:
                            ds.read_into((slice(za, zb),), host.numpy()[0])
                    with torch.cuda.stream(st):
                        t[:, za - z_lo:zb - z_lo].copy_(host, non_blocking=True)
                st.synchronize()
                with loaded_cv:
                    loaded["z"] = zb
                    loaded_cv.notify_all()
        except BaseException as exc:  # noqa: BLE001 - handed to the thread that waits for the sections
            with loaded_cv:
                loaded["error"] = exc
                loaded_cv.notify_all()
    load_future = side.submit(load_inputs)

    def await_sections(blk):
        """the block's reads, mirror images about the dataset's first / last section included, lie below this section"""
        lo_r, hi_r = org[0] + blk[0] - ctx[0], org[0] + blk[0] - ctx[0] + in_shape[0]
        need = z_hi if hi_r > nz else min(z_hi, max(hi_r, 1 - lo_r))
        with _trace.span("predict: block loop waits for input sections", True), loaded_cv:
            while loaded["z"] < need and loaded["error"] is None:
                loaded_cv.wait()
            if loaded["error"] is not None:
                raise loaded["error"]
    with _trace.span("predict: wait for the model"):
        model = model_future.result()
    two_d = model.two_d
    adj = int(cfg["net_config"].get("adj_slices", 1))
    if (len(vols) if not two_d else adj * len(vols)) != model._cfg.in_channels:
        raise ValueError(f"the input datasets hold {len(vols)} channels, the network takes {model._cfg.in_channels}")

    def read_block(v, blk):
        """block read with the z axis re-based on the resident slab; a mirror image beyond a dataset face is taken from
        the full-depth side of the slab (the slab reaches the face whenever a read does)"""
        off = [org[d] + blk[d] - ctx[d] for d in range(3)]
        if z_lo == 0 and z_hi == nz:
            return extract_block_reflect(v, off, in_shape)
        lo, hi = off[0], off[0] + in_shape[0]
        if lo >= z_lo and hi <= z_hi:
            return extract_block_reflect(v, [off[0] - z_lo, off[1], off[2]], in_shape)
        # the read crosses a dataset face: mirror indices by hand (rare: first / last layer of blocks)
        idx = torch.arange(lo, hi, device=v.device)
        period = 2 * (nz - 1)
        idx = idx % period
        idx = torch.where(idx >= nz, period - idx, idx) - z_lo
        plane = extract_block_reflect(v, [0, off[1], off[2]], [v.shape[0], in_shape[1], in_shape[2]])
        return plane.index_select(0, idx)

    # Write-behind: the device -> host copy of a block's outputs runs on a side stream into page-locked memory and the chunk
    # encoding + file writes (the library's threads, straight from that buffer: zarr_io.write_from) on a small pool, while the
    # next blocks are predicted.
    WRITERS = 8
    pool = cf.ThreadPoolExecutor(max_workers=WRITERS, thread_name_prefix="bsmi-write")
    pinned, copy_streams = {}, {}

    def write_block(blk, hi, u8, ready):
        with _trace.span("predict: writer waits for its block", True):
            ready.synchronize()
        # ^ on this pool thread: a copy stream parked behind a device-side wait is a queue the command
        tid = threading.get_ident()   # processor polls for the whole forward pass, at the predict stream's expense
        if tid not in pinned:
            pinned[tid] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in u8]
            from .volume import io_stream
            copy_streams[tid] = io_stream(dev)
        host = []
        with torch.cuda.stream(copy_streams[tid]):
            for buf, t in zip(pinned[tid], u8):
                h = buf[:, :hi[0], :hi[1], :hi[2]]
                h.copy_(t[:, :hi[0], :hi[1], :hi[2]], non_blocking=True)
                host.append(h)
        with _trace.span("predict: device -> host copy", True):
            copy_streams[tid].synchronize()
        with _trace.span("predict: encode + write", True):
            for ds, t in zip(outs, host):
                ds.write_from((slice(None),) + tuple(slice(blk[d], blk[d] + hi[d]) for d in range(3)), t.numpy())

    inflight, redo = [], []

    # Predict lanes (`pred_lanes` of the predict config, default 2; an addition to the reference's config): engines with the same
    # weights, block k on engine k mod K and that engine's stream, so that the forward passes of consecutive blocks overlap -- the
    # memory-bound launches of one beside the matrix launches of the other (+7 % on the resident pipeline).
    # The further engines are packed on a side thread while the first one already predicts.
    from .volume import predict_stream
    n_lanes = max(1, int(cfg.get("pred_lanes", os.environ.get("BSMI_PRED_LANES", 2))))
    if len(mine) < 64 and "pred_lanes" not in cfg:
        n_lanes = 1  # a second engine costs 0.4 s of packing on the side thread: a job of a second or two does not earn it back
    engines, lane_streams = [model], [torch.cuda.current_stream(dev)]
    clones = [side.submit(lambda: (torch.cuda.set_device(device), model.clone())[1]) for _ in range(n_lanes - 1)]
    counter = [0]
    wait_clones = os.environ.get("BSMI_PRED_LANES_WAIT") == "1"   # tests: every lane from the first block on

    def predict_and_submit(blk):
        await_sections(blk)
        while clones and (wait_clones or clones[0].done()):
            engines.append(clones.pop(0).result())
            lane_streams.append(predict_stream(dev, len(engines) - 1))   # volume.py's predict-lane streams (one set per process: hardware queues are few)
        lane = counter[0] % len(engines)
        counter[0] += 1
        with torch.cuda.stream(lane_streams[lane]):
            chans = [read_block(v, blk) for v in vols]
            if two_d:  # section z of the stack sees sections z .. z + adj - 1 of the read block as its channels
                chans = [c[i:i + out_shape[0]] for c in chans for i in range(adj)]
            # (workers that share a GPU -- the reference deals worker w to GPU w % num_gpus, predict.py:46-49 -- simply overlap on
            # it: the kernel whose scratch segment two overlapping passes corrupted is gone)
            u8 = engines[lane].predict_u8(chans[0] if len(chans) == 1 else torch.stack(chans))
            ready = torch.cuda.Event()
            ready.record(lane_streams[lane])
        hi = [min(out_shape[d], roi_vox[d] - blk[d]) for d in range(3)]
        return pool.submit(write_block, blk, hi, u8, ready)

    def settle(limit):
        """await the oldest writes: bounds the blocks in flight (their outputs stay alive until written)"""
        while len(inflight) > limit:
            blk, fut = inflight.pop(0)
            try:
                with _trace.span("predict: block loop waits for a writer", True):
                    fut.result()
            except Exception:  # noqa: BLE001 - the block is predicted and written again below
                redo.append(blk)

    def process(blk):
        inflight.append((blk, predict_and_submit(blk)))
        settle(2 * WRITERS)
    try:
        with _trace.span("predict: block loop"):
            state = run_blocks("PredictBlockwiseTask", mine, process, MAX_RETRIES)
        with _trace.span("predict: drain the writers"):
            settle(0)
        if redo:  # a write failed after its block had been counted: the whole block again, awaited, with the retries left
            again = run_blocks("PredictBlockwiseTask", redo, lambda blk: predict_and_submit(blk).result(), MAX_RETRIES - 1)
            state.completed_count += again.completed_count - len(redo)
            state.failed_count += again.failed_count
            state.failed_blocks += again.failed_blocks
    finally:
        pool.shutdown()
        load_future.result()
        for c in clones:
            c.cancel()
        side.shutdown()
        _trace.report()
    return state


def _worker(rank, world, cfg, precision, results):
    results[rank] = predict_blocks(cfg, rank, world, precision=precision).as_tuple()


def run_prediction(config_file, setup_ids=None, precision="bf16x3", **kwargs):
    from .blockwise import TaskState, check_task_states
    all_ids = list(load_toml(config_file).keys())
    valid = {**{s.split("-")[0]: s for s in all_ids}, **{s.split("-")[-1]: s for s in all_ids}, **{s: s for s in all_ids}}
    setups = sorted(setup_ids.strip().split()) if setup_ids else all_ids
    for s_id in setups:
        if s_id not in valid:
            raise ValueError(f"Setup ID {s_id} not found in {all_ids}")
        cfg = get_pred_config(config_file, valid[s_id], **kwargs)
        prepare_outputs(cfg, open_ds(cfg["input_datasets"][0]))
        world = max(1, int(cfg["num_gpus"]))
        if world == 1:
            state = predict_blocks(cfg, 0, 1, precision=precision)
        else:
            # one worker process per GPU (predict.py:46-49; on a box with fewer GPUs the workers share them); a worker
            # that dies takes the run down with it, like the reference's CalledProcessError
            import torch.multiprocessing as mp
            with mp.Manager() as mgr:
                results = mgr.dict()
                mp.spawn(_worker, args=(world, cfg, precision, results), nprocs=world, join=True)
                state = TaskState("PredictBlockwiseTask")
                for r in range(world):
                    state.merge(TaskState.from_tuple("PredictBlockwiseTask", results[r]))
        check_task_states({"PredictBlockwiseTask": state})  # reference blockwise.py:12-22


def worker_main(argv=None):
    """The per-setup worker script contract of the reference (models/3d_affs/predict.py:19-58):
        predict.py -c CKPT -i IN_ZARR... -o OUT_ZARR... [-ro "z y x" -rs "z y x"] [-n W] [-d]
    run from (or with --setup-dir pointing at) the setup directory that holds net_config.json.  Output datasets are
    created by the caller (predict.py:169-178); this worker only writes inside the ROI.  -n / -d are accepted for
    compatibility: blocks go through the device one after the other either way."""
    import argparse
    ap = argparse.ArgumentParser(prog="predict.py")
    ap.add_argument("--checkpoint", "-c", required=True)
    ap.add_argument("--input_datasets", "-i", required=True, action="append")
    ap.add_argument("--output_datasets", "-o", required=True, action="append")
    ap.add_argument("--roi_offset", "-ro", type=str)
    ap.add_argument("--roi_shape", "-rs", type=str)
    ap.add_argument("--num_workers", "-n", type=int, default=1)
    ap.add_argument("--daisy", "-d", action="store_true")
    ap.add_argument("--setup-dir", default=os.getcwd())
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f32", "bf16"])
    a = ap.parse_args(argv)
    with open(os.path.join(a.setup_dir, "net_config.json")) as f:
        net_config = json.load(f)
    if not os.path.exists(a.checkpoint) and not os.path.exists(a.checkpoint + ".ckpt"):
        raise FileNotFoundError(f"Neither {a.checkpoint} nor {a.checkpoint}.ckpt were found.")
    if len(a.output_datasets) != len(net_config["outputs"]):
        raise ValueError(f"{len(a.output_datasets)} output datasets for {len(net_config['outputs'])} network outputs")
    in_ds = open_ds(a.input_datasets[0])
    vs = in_ds.voxel_size
    if a.roi_offset is not None:
        roi = ([int(x) for x in a.roi_offset.split()], [int(x) for x in a.roi_shape.split()])
    else:
        roi = (list(in_ds.roi[0]), list(in_ds.roi[1]))
    cfg = dict(setup_dir=a.setup_dir, checkpoint=a.checkpoint, net_config=net_config, input_datasets=a.input_datasets,
               output_datasets=a.output_datasets, output_roi=roi, voxel_size=list(vs), **block_rois(net_config, vs))
    from .blockwise import check_task_states
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    state = predict_blocks(cfg, rank, world, precision=a.precision)
    check_task_states({"PredictBlockwiseTask": state})
    return 0


if __name__ == "__main__":
    raise SystemExit(worker_main())
