"""Per-GPU block pipeline: the HIP-stream replacement of the reference's chunk scheduler
for one worker.

Reference being replaced (paths relative to /root/reference/bootstrapper):
  predict.py:22-49      daisy task per block, one worker process per GPU
  models/3d_affs/predict.py:128-162  gp.Scan / DaisyRequestBlocks loop: one model(input)
                        per block, reflect-padded reads, uint8 writes into the block's write ROI
  post/watershed.py:206-354  fragments + agglomeration on the predicted affinities

Blocks are independent (read_write_conflict=False, predict.py:37): block i's U-Net runs on
the predict stream while the sequential-at-heart segmentation kernels of blocks i-1, i-2, ...
run on `n_seg_lanes` other streams, each with its own workspace.  No collectives.
"""
import ctypes as C

import torch

from . import _lib
from .unet import extract_block_reflect
from .post.engine import SegEngine


def cu_masked_stream(device, first_bit, n_bits, total_bits):
    """HIP stream limited to CU-mask bits [first_bit, first_bit + n_bits) (bit i = a CU of XCD i % 8),
    wrapped for torch.  Returns (torch stream, raw handle to destroy)."""
    words = (total_bits + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(first_bit, first_bit + n_bits):
        mask[i // 32] |= 1 << (i % 32)
    raw = C.c_void_p()
    _lib.check(_lib.lib.bsmi_stream_create_cu_mask(int(device), mask, words, C.byref(raw)))
    return torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", device)), raw


class BlockPipeline:
    def __init__(self, model, out_block, context, thresholds=(0.2, 0.35, 0.5), min_seed_distance=10,
                 n_seg_lanes=4, segment=True, device=0, keep_outputs=False, models=None, seg_cus=0, seg_stages=("ws", "agg"), seg_burst=0):
        # `models`: optional list of Model replicas (same weights), one per predict stream.  Two
        # predict streams let the last, partially filled round of workgroups of one block's
        # conv launch overlap with the other block's launches (each replica owns its activations).
        self.models = list(models) if models else [model]
        self.model = model = self.models[0]
        self.out_block = tuple(out_block)
        self.context = tuple(context)
        self.in_block = tuple(o + 2 * c for o, c in zip(out_block, context))
        if model.output_shape(self.in_block) != self.out_block:
            raise ValueError(f"network maps {self.in_block} to {model.output_shape(self.in_block)}, not {self.out_block}")
        self.thresholds = list(thresholds)
        self.msd = int(min_seed_distance)
        self.segment = bool(segment)
        self.seg_stages = tuple(seg_stages)  # diagnostic: run only part of the segmentation half
        # seg_burst = B > 0: the segmentation of B consecutive blocks is launched together, after the last of them has been
        # predicted (needs n_seg_lanes >= B to run them side by side); 0: every block's segmentation follows its predict
        self.seg_burst = int(seg_burst)
        self._pending = []
        self.dev = torch.device("cuda", device)
        self.keep = keep_outputs
        # The U-Net's big layers run as persistent workgroups that want a whole CU each (all of its
        # registers and LDS); one resident segmentation wave is enough to keep such a workgroup
        # off a CU.  So the two halves of the path get disjoint CU sets: `seg_cus` CUs (a multiple
        # of 8: the same number in every XCD) for the latency-bound segmentation lanes, the rest
        # for the predict streams.  seg_cus = 0 (default): shared CUs, predict streams at high priority.
        # Measured on the benchmark (8 lanes): 32 / 48 reserved CUs -> 46.7 / 54.7 Mvox/s against 76.2
        # shared: the lanes' wide kernels (seeds, RAG scan) need the whole chip for their short bursts.
        self._raw_streams = []
        n_cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        self.seg_cus = int(seg_cus) // 8 * 8 if (self.segment and n_seg_lanes > 0) else 0
        if self.seg_cus >= n_cus:
            raise ValueError("seg_cus must leave CUs for the predict streams")

        def make_stream(first, count, priority):
            if not self.seg_cus:
                return torch.cuda.Stream(self.dev, priority=priority)
            st, raw = cu_masked_stream(device, first, count, n_cus)
            self._raw_streams.append(raw)
            return st
        self.pred_streams = [make_stream(self.seg_cus, n_cus - self.seg_cus, -1) for _ in self.models]
        for m in self.models:
            m.set_persistent_grid(n_cus - self.seg_cus if self.seg_cus else -1)
        self.lanes = []
        if self.segment:
            for _ in range(n_seg_lanes):
                self.lanes.append(dict(engine=SegEngine(self.out_block, device), stream=make_stream(0, self.seg_cus, 0),
                                       done=None, affs=None))
        self.n_done = 0
        self.results = []

    def run(self, volume_u8, block_offsets):
        """volume_u8: uint8 CUDA tensor (D,H,W) resident in HBM; block_offsets: output-block
        origins (z,y,x) in voxels.  Returns after all work is queued; call finish()."""
        for i, off in enumerate(block_offsets):
            lane = self.lanes[i % len(self.lanes)] if self.segment else None
            pstream = self.pred_streams[i % len(self.models)]
            with torch.cuda.stream(pstream):
                if lane is not None:
                    # bound the run-ahead: a lane's block is predicted only after the lane has finished its block before
                    # last (bursts: the burst before last -- the last one is still being segmented while this one is
                    # predicted) resp. its last block (block by block)
                    gate = lane.get("done_before") if self.seg_burst else lane["done"]
                    if gate is not None:
                        pstream.wait_event(gate)
                raw = extract_block_reflect(volume_u8, [o - c for o, c in zip(off, self.context)], self.in_block)
                u8 = self.models[i % len(self.models)].predict_u8(raw)
                ready = torch.cuda.Event()
                ready.record(pstream)
            if lane is None:
                if self.keep:
                    self.results.append((off, u8, None, None))
                continue
            self._pending.append((lane, off, raw, u8))
            if len(self._pending) >= max(1, self.seg_burst):
                self._launch_segmentation(ready)
        if self._pending:
            self._launch_segmentation(ready)

    def _launch_segmentation(self, ready):
        """Segmentation of the pending blocks, each on its lane, once `ready` (the last predict of the group) has fired."""
        pending, self._pending = self._pending, []
        for lane, off, raw, u8 in pending:
            with torch.cuda.stream(lane["stream"]):
                lane["stream"].wait_event(ready)
                affs = u8[0][:3]
                if self.seg_stages == ("none",):  # diagnostic: the lane structure (events, streams) without any segmentation kernel
                    frags = segs = affs
                elif "ws" in self.seg_stages or lane.get("frags") is None:
                    frags, max_id = lane["engine"].ws_fragments(affs, True, self.msd)
                    lane["frags"] = frags
                else:
                    frags = lane["frags"]  # diagnostic mode: agglomerate the lane's first fragments again
                if self.seg_stages != ("none",):
                    segs = (lane["engine"].agglomerate_mean(affs, frags, self.thresholds) if "agg" in self.seg_stages
                            else frags[None])
                for t in (raw, affs, frags, segs) + tuple(u8):
                    t.record_stream(lane["stream"])
                done = torch.cuda.Event()
                done.record(lane["stream"])
                lane["done_before"] = lane["done"]
                lane["done"] = done
            if self.keep:
                self.results.append((off, u8, frags, segs))
            self.n_done += 1

    def finish(self):
        for ps in self.pred_streams:
            ps.synchronize()
        for lane in self.lanes:
            lane["stream"].synchronize()
            lane["engine"].status()
        out, self.results = self.results, []
        return out

    def close(self):
        """Destroy the CU-masked streams (after finish())."""
        torch.cuda.synchronize(self.dev)
        raws, self._raw_streams = self._raw_streams, []
        self.pred_streams, self.lanes = [], []
        for raw in raws:
            _lib.lib.bsmi_stream_destroy(self.dev.index, raw)

    def __del__(self):
        try:
            if getattr(self, "_raw_streams", None):
                self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def block_grid(vol_shape, out_block):
    """Output-block origins covering vol_shape (fit='overhang', predict.py:39), z-major."""
    offs = []
    for z in range(0, vol_shape[0], out_block[0]):
        for y in range(0, vol_shape[1], out_block[1]):
            for x in range(0, vol_shape[2], out_block[2]):
                offs.append((z, y, x))
    return offs
