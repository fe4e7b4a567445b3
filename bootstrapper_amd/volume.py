"""One rank's share of a volume, resident in HBM, taken through the whole hot path:

  blockwise predict -> fragments per block with context -> per-block RAG edge scoring ->
  global thresholded connected components -> LUT -> relabel

so that what comes out is ONE consistent segmentation of the volume, the thing `bs predict` followed by
`bs segment --ws` with `blockwise = true` produces.  Reference being replaced (paths relative to
/root/reference/bootstrapper):
  predict.py:22-49, models/3d_affs/predict.py:128-162      blockwise prediction
  post/watershed.py:56-139, post/blockwise/watershed_frags.py:196-246   fragments (block + context reads, fill 0)
  post/watershed.py:141-153, post/blockwise/waterz_agglom.py:106-170    RAG edge scoring
  post/watershed.py:155-203                                 connected components, LUT, relabel

Layout and sharding.  The volume is cut into slabs of whole block layers along z, one slab per rank (the reference
deals blocks to workers through daisy and lets them meet in the Zarr store and the RAG database; here a rank keeps
its affinities and fragments in HBM and meets its two z-neighbours only at the slab faces).  Per rank:
  affs   uint8 [3][Z + 2c][Y + 2c][X + 2c]   slab + context margin c (zeros outside the volume, `fill_value=0`)
  frags  int64 [Z + 2c][Y + 2c][X + 2c]      global fragment ids = block id * voxels per block + label
  segs   int64 [thresholds][Z][Y][X]
The exchange steps of the path, and the only communication: the context margins of affinities and fragments at the
slab faces (point to point with the z-neighbours), the scored edges to rank 0, the LUT back to every rank.
Blocks are independent inside a stage: they run side by side on `n_lanes` HIP streams, each lane with its own
segmentation workspace; a stage ends with one host synchronisation.

HBM residency.  Everything of a rank's slab stays on the device: 3 B (affinities) + 8 B (fragments) + 8 B (the interior copy the
relabel reads) + 8 B per threshold (segmentations) per voxel -- 43 B per voxel at three thresholds, 39 GB for a 1024^3 slab --
plus per block the node tables (label_cap x 32 B) and the scored edge list (edge_cap x 20 B), plus one workspace per lane.
`SlabSegmenter.hbm_bytes()` gives the figure; the constructor refuses a slab that does not fit the device's free memory and
says how many ranks the volume needs.  label_cap / edge_cap are starting sizes: a block that needs more makes the tables grow
(`_regrow`), only its own part is redone.
"""
import numpy as np
import torch

from . import _lib
from .post.blockwise import shrink_blocks
from .post.engine import SegEngine, lut_relabel


def slab_layers(n_layers, world):
    """Block layers (along z) per rank: contiguous, as even as possible, earlier ranks take the extra ones."""
    q, r = divmod(int(n_layers), int(world))
    counts = [q + (1 if i < r else 0) for i in range(world)]
    starts = [sum(counts[:i]) for i in range(world)]
    return starts, counts


def rank_grid(world, layers, rows):
    """(Rz, Ry): the ranks laid out over the block layers (z) and the block rows (y) of a volume so that as many of them as
    possible have blocks (Rz <= layers, Ry <= rows, Rz * Ry <= world); among equals the grid with more z parts.  The reference
    hands any block to any worker (post/watershed.py:118-153); here a rank owns a box of blocks and meets its neighbours at the
    faces, so a flat volume (CREMI: one layer of 10 x 10 blocks) is cut along y."""
    best = (1, 1)
    for rz in range(1, min(int(world), int(layers)) + 1):
        ry = min(int(world) // rz, int(rows))
        if (rz * ry, rz) > (best[0] * best[1], best[0]):
            best = (rz, ry)
    return best


def exchange_faces(low_out, high_out, low_in, high_in, lo_peer, hi_peer, group=None):
    """Send this rank's first / last layers along one axis to its two neighbours and receive theirs: `low_out` goes to rank
    `lo_peer` and arrives there as `high_in`, `high_out` goes to `hi_peer` and arrives as its `low_in` (a peer of None: the
    volume ends there).  Device tensors travel over RCCL point to point; under gloo (CPU tests, rehearsals) they are staged
    through host memory."""
    import torch.distributed as dist
    if lo_peer is None and hi_peer is None:
        return
    staged = dist.get_backend(group) != "nccl" and low_out.is_cuda
    ops, recvs = [], []

    def post(kind, t, peer):
        if kind == "send":
            buf = t.contiguous().cpu() if staged else t.contiguous()
            ops.append(dist.P2POp(dist.isend, buf, peer, group))
        else:
            buf = torch.empty(t.shape, dtype=t.dtype, device="cpu" if staged else t.device)
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
            recvs.append((t, buf))
    if lo_peer is not None:
        post("send", low_out, lo_peer)
        post("recv", low_in, lo_peer)
    if hi_peer is not None:
        post("send", high_out, hi_peer)
        post("recv", high_in, hi_peer)
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, buf in recvs:
        dst.copy_(buf)


def stitch_components(nodes, edges, scores, thresholds):
    """post/watershed.py:155-186 on the host: drop unscored edges, connected components per threshold.
    -> list of component ids aligned with `nodes` (ascending uint64)."""
    if edges.shape[0] == 0 or np.isnan(scores).all():
        return [nodes.copy() for _ in thresholds]
    # (unscored edges -- NaN -- stay in the arrays: no threshold admits them, and copying 460 000 edges to drop a few was 3 ms)
    # one library call for all thresholds: the node look-up of the edges once (host threads), the unions of a lower
    # threshold carried over to the higher ones -- this is what the other ranks wait for at 8 GPUs
    from .post.watershed import connected_components_multi
    return list(connected_components_multi(nodes, edges, scores, thresholds))


def gather_and_stitch(nodes, edges, scores, thresholds, rank=0, world=1, group=None):
    """The one many-to-one step of the path: every rank's fragment ids (ascending per rank) and
    scored edges go to rank 0, which computes the components per threshold; every rank gets (all nodes, [components per
    threshold]) back -- the fragment-segment LUTs of post/watershed.py:184-186."""
    import torch.distributed as dist
    mine = (np.asarray(nodes, np.uint64), np.asarray(edges, np.uint64).reshape(-1, 2), np.asarray(scores, np.float32))
    if world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0, group=group)
    else:
        parts = [mine]
    luts = [None]
    if rank == 0:
        nodes_all = np.concatenate([p[0] for p in parts])
        edges_all = np.concatenate([p[1] for p in parts])
        scores_all = np.concatenate([p[2] for p in parts])
        if nodes_all.size > 1 and not np.all(nodes_all[1:] > nodes_all[:-1]):
            nodes_all = np.sort(nodes_all)   # ranks that share block layers (a y cut) hold interleaved id ranges
            if not np.all(nodes_all[1:] > nodes_all[:-1]):
                raise ValueError("two ranks report the same fragment id")
        luts[0] = (nodes_all, stitch_components(nodes_all, edges_all, scores_all, thresholds))
    if world > 1:
        dist.broadcast_object_list(luts, src=0, group=group)
    return luts[0]


# HIP streams of the pipelines, one set per device for the life of the process.  Every torch stream that has been used is
# a hardware queue, and the runtime is told to keep one per stream up to GPU_MAX_HW_QUEUES (24): a second pipeline (a
# warm-up one, say) that drew fresh streams from torch's pool pushed the count past that, streams began to share queues
# and the predict stream ran 12 % slower from then on.
_LANE_STREAMS = {}
_PREDICT_STREAMS = {}


def lane_streams(device, n):
    pool = _LANE_STREAMS.setdefault(torch.device(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


_IO_NEXT = {}


def io_stream(device):
    """A stream for a driver's copy / encode helper thread (write-behind of `bs predict`, the readers and dataset writers of
    `bs segment`): one of the lane streams, dealt round robin.  The helpers move data while the lanes are idle (before and after
    the block stages), and a stream of their own each would be more hardware queues than the runtime keeps (GPU_MAX_HW_QUEUES =
    24: 20 lanes + the predict and default streams; bench.py's `drivers` leg, in a process that had used all of them, ran `bs
    predict` 15 % slower with eight fresh copy streams)."""
    dev = torch.device(device)
    pool = lane_streams(dev, 8)
    i = _IO_NEXT.get(dev, 0)
    _IO_NEXT[dev] = (i + 1) % len(pool)
    return pool[i]


def predict_stream(device, lane=0):
    """the stream of predict lane `lane` (0: THE predict stream) of a device"""
    dev = torch.device(device)
    key = dev if lane == 0 else (dev, int(lane))
    if key not in _PREDICT_STREAMS:
        _PREDICT_STREAMS[key] = torch.cuda.Stream(dev, priority=-1)
    return _PREDICT_STREAMS[key]


class SlabSegmenter:
    """Fragments, RAG scoring, stitching and relabelling of one rank's slab of affinities (see the module docstring)."""

    def __init__(self, slab_shape, block, context, total_layers, layer0, thresholds=(0.2, 0.35, 0.5),
                 fragments_in_xy=True, min_seed_distance=10, filter_fragments=0.0, remove_debris=0, discretize_queue=256,
                 n_lanes=8, device=0, rank=0, world=1, group=None, edge_cap=1 << 17, label_cap=1 << 16, exchange_affs=True,
                 epsilon_agglomerate=0.0, sigma=None, noise_eps=None, bias=None, noise_seed=0, seed_eps=None,
                 grid=None, total_rows=None, row0=0, obj_group=None, host_scores=True, cc_inclusive=True, queue_bins_formula="n_minus_1",
                 lazy_outputs=False):
        """slab_shape: this rank's box of the volume (whole blocks but for the volume's far faces).  The ranks form a grid
        `grid` = (Rz, Ry) over z and y (default: (world, 1), slabs of block layers), rank = rz * Ry + ry; the volume has
        `total_layers` block layers and `total_rows` block rows, this rank's first ones are `layer0`, `row0`.
        obj_group: process group of the object collectives (edges to rank 0, LUT back); under an RCCL default group a gloo
        group beside it, so that pickled objects do not travel through device tensors.
        host_scores: the merge loop of a block's edge scoring runs on host threads for all blocks at once (`_collect`; the lanes
        only build the region graphs) instead of as one wave per block on the lanes: the scores are needed on the host anyway,
        and a host core replays that loop in a fraction of the 12 ms a wave takes."""
        self.host_scores = bool(host_scores)
        # The two choices that waterz / funlib.segment make and that cannot be checked in this repository (DESIGN.md section 2;
        # tools/gen_goldens_waterz.py makes the vectors that decide; segment config keys of the same names):
        #   cc_inclusive       an edge joins two fragments when score <= threshold (True, the specification) or score < threshold
        #   queue_bins_formula bin of a score in the N-bin queue: "n_minus_1" (int)(score (N-1)) (the specification) or "n" min(N-1, (int)(score N))
        self.cc_inclusive = bool(cc_inclusive)
        self.queue_bins_formula = str(queue_bins_formula)
        from .post.engine import QUEUE_BINS_FORMULAS
        if self.queue_bins_formula not in QUEUE_BINS_FORMULAS:
            raise ValueError(f"queue_bins_formula {queue_bins_formula!r}: one of {sorted(QUEUE_BINS_FORMULAS)}")
        if self.queue_bins_formula != "n_minus_1" and not self.host_scores:
            raise NotImplementedError("the device merge loop (host_scores=False) implements the specified bin rule only")
        self.shape = tuple(int(s) for s in slab_shape)
        self.block = tuple(int(b) for b in block)
        self.ctx = tuple(int(c) for c in context)
        self.thresholds = [float(t) for t in thresholds]
        self.fragments_in_xy, self.msd = bool(fragments_in_xy), int(min_seed_distance)
        self.filter_fragments, self.remove_debris = float(filter_fragments), int(remove_debris)
        self.bins = int(discretize_queue)
        self.epsilon = float(epsilon_agglomerate or 0.0)
        self.shift = dict(sigma=sigma, noise_eps=noise_eps, bias=bias, seed_eps=seed_eps)
        self.noise_seed = int(noise_seed)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.obj_group = obj_group if obj_group is not None else group
        self.grid = (int(world), 1) if grid is None else (int(grid[0]), int(grid[1]))
        if self.grid[0] * self.grid[1] != self.world:
            raise ValueError(f"rank grid {self.grid} for {self.world} ranks")
        self.rz, self.ry = divmod(self.rank, self.grid[1])
        # neighbours along z / y (None: the volume ends at that face)
        self.peers = ((self.rank - self.grid[1] if self.rz > 0 else None, self.rank + self.grid[1] if self.rz < self.grid[0] - 1 else None),
                      (self.rank - 1 if self.ry > 0 else None, self.rank + 1 if self.ry < self.grid[1] - 1 else None))
        # False: the caller fills the context margins of the affinities itself (a driver reads them from the dataset,
        # where also the data beyond the ROI is real); the fragments' margins are always exchanged
        self.exchange_affs = bool(exchange_affs)
        self.dev = torch.device("cuda", int(device))
        self.boxes = shrink_blocks(self.shape, self.block)
        counts = self.counts = [-(-s // b) for s, b in zip(self.shape, self.block)]
        # global z-major block ids (`block.block_id` of the reference's tasks; daisy numbers differently)
        rows = counts[1] if total_rows is None else int(total_rows)
        self.block_ids = [((int(layer0) + iz) * rows + (int(row0) + iy)) * counts[2] + ix
                          for iz in range(counts[0]) for iy in range(counts[1]) for ix in range(counts[2])]
        self.total_blocks = int(total_layers) * rows * counts[2]
        self.nvb = int(np.prod(self.block))
        padded = tuple(s + 2 * c for s, c in zip(self.shape, self.ctx))
        K = len(self.boxes)
        read_vox = int(np.prod([min(b, s) + 2 * c for b, s, c in zip(self.block, self.shape, self.ctx)]))
        self.edge_cap = max(64, min(int(edge_cap), 3 * read_vox))
        self.label_cap = max(64, min(int(label_cap), self.nvb))
        need = self.hbm_bytes(self.shape, self.ctx, len(self.thresholds), K, self.label_cap, self.edge_cap)
        free, _total = torch.cuda.mem_get_info(self.dev)
        free += torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)   # torch's cached, unused blocks count as free
        if need > free:
            per_layer = need / max(1, self.counts[0])
            raise MemoryError(f"a slab of {self.shape} voxels needs {need / 2**30:.1f} GiB of HBM ({need / max(1, int(np.prod(self.shape))):.0f} B per voxel: "
                              f"affinities, fragments, {len(self.thresholds)} segmentations, per-block tables) and {free / 2**30:.1f} GiB are free on "
                              f"{self.dev}: give each rank at most {max(1, int(free // per_layer))} block layer(s), i.e. use more workers / GPUs")
        self.affs = torch.zeros((3,) + padded, dtype=torch.uint8, device=self.dev)
        self.frags = torch.zeros(padded, dtype=torch.int64, device=self.dev)
        # outputs of the stitch, allocated with the slab: a first stitch that allocates them pays for it inside whatever times
        # the pipeline (whole 1024^3 volume: 26 GB of segmentations + 8.6 GB of interior fragments, a second of hipMalloc)
        # (lazy_outputs: a driver that writes the segmentations out threshold by threshold -- `stitch(consume=...)` -- needs one
        # of them at a time and allocates it beside its reads: `ensure_outputs`)
        self.segs = self._fr = self._one = None
        if not lazy_outputs:
            self.segs = torch.empty((len(self.thresholds),) + self.shape, dtype=torch.int64, device=self.dev)
            self._fr = torch.empty(self.shape, dtype=torch.int64, device=self.dev)
        self.nums = torch.zeros(K, dtype=torch.int64, device=self.dev)
        self.sizes = torch.zeros((K, self.label_cap), dtype=torch.int64, device=self.dev)
        self.sums = torch.zeros((K, self.label_cap, 3), dtype=torch.int64, device=self.dev)
        self.edges = torch.empty((K, self.edge_cap, 2), dtype=torch.int64, device=self.dev)
        self.scores = torch.empty((K, self.edge_cap), dtype=torch.float32, device=self.dev)   # host_scores: the edges' voxel-pair counts (int32 view)
        self.esums = torch.empty((K, self.edge_cap), dtype=torch.int64, device=self.dev) if self.host_scores else None
        self.counts_dev = torch.zeros((K, 4), dtype=torch.int64, device=self.dev)
        self.frag_done = [None] * K
        read = tuple(min(b, s) + 2 * c for b, s, c in zip(self.block, self.shape, self.ctx))
        self.lanes = []
        # Hardware queues: the runtime keeps one per stream up to GPU_MAX_HW_QUEUES (24); past that, streams share queues and
        # the predict stream's launches wait behind lane work.  20 lanes + the predict and default streams fit; RCCL brings
        # streams of its own (one rank under `--force-dist`: 20 lanes 55.3 Mvoxels/s, predict 34.5 ms per block; 16 lanes
        # 65.0, 28.9 ms), so a rank of an RCCL job stops at 16.
        import torch.distributed as dist
        import os
        dist_cap = int(os.environ.get("BSMI_DIST_LANES", "16"))   # (experiments: more lanes under RCCL with a larger GPU_MAX_HW_QUEUES)
        if int(n_lanes) > dist_cap and dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend()):
            n_lanes = dist_cap
        for stream in lane_streams(self.dev, max(1, min(int(n_lanes), K))):
            self.lanes.append(dict(engine=SegEngine(read, self.dev.index), stream=stream,
                                   a=torch.empty((3,) + read, dtype=torch.uint8, device=self.dev),
                                   f=torch.empty(read, dtype=torch.int64, device=self.dev),
                                   lab=torch.empty(tuple(min(b, s) for b, s in zip(self.block, self.shape)), dtype=torch.int64,
                                                   device=self.dev)))
        # The zero fills of the slab above are queued on the CURRENT stream; what fills the slab next runs on other streams (the
        # readers' copy streams, the lanes): without this wait a fill could land on top of data (found with the streamed driver,
        # whose third pass read affinities into a slab whose memset had not run yet: an all-zero block layer)
        torch.cuda.current_stream(self.dev).synchronize()
        self.nodes = None
        self.rag_edges = self.rag_scores = None
        self.luts = None
        self._lane_load = [0] * len(self.lanes)   # tasks queued per lane by the run in progress (_take_lane)
        self._lane_of = {}
        self._lane_limit = len(self.lanes)        # lanes _take_lane may use (overlap mode: few while the predict stream runs)
        self.overlap_lanes = 3
        # host seconds of the last run's parts, per rank (bench.py prints every rank's, so that a slow rank shows): the face
        # exchanges (posting + completion of the point-to-point transfers as the host sees them), the two block stages up to
        # the collected edges (exchanges included), the stitch (edge gather, connected components on rank 0, LUT broadcast, relabel)
        self.timers = {"exchange": 0.0, "blocks": 0.0, "stitch": 0.0}

    @staticmethod
    def hbm_bytes(shape, ctx, n_thresholds, n_blocks, label_cap=1 << 16, edge_cap=1 << 17):
        """device bytes of a slab (module docstring), the lanes' workspaces not included"""
        vox = int(np.prod(shape))
        padded = int(np.prod([s + 2 * c for s, c in zip(shape, ctx)]))
        return 11 * padded + 8 * vox * (1 + n_thresholds) + n_blocks * (32 * label_cap + 28 * edge_cap + 40)

    # -- views ---------------------------------------------------------------------------
    def interior(self, t):
        sl = tuple(slice(c, c + s) for c, s in zip(self.ctx, self.shape))
        return t[(Ellipsis,) + sl]

    def write_view(self, k):
        """affs[:, write box of block k]: where the predict stage stores the block's first three channels."""
        b, e = self.boxes[k]
        return self.affs[(slice(None),) + tuple(slice(c + lo, c + hi) for c, lo, hi in zip(self.ctx, b, e))]

    def _read_slices(self, k):
        b, e = self.boxes[k]
        return tuple(slice(lo, hi + 2 * c) for c, lo, hi in zip(self.ctx, b, e))

    def _buf(self, t, shape, lead=()):
        n = int(np.prod(lead + shape))
        return t.reshape(-1)[:n].view(lead + shape)

    # -- exchange ------------------------------------------------------------------------
    def _exchange(self, t):
        """context margins of `t` ([..., Zp, Yp, Xp]) at the faces shared with other ranks <- the neighbours' outermost layers.  The
        caller has made sure those layers are complete; -> an event on the current stream that fires when the margins are."""
        import time
        t_in = time.perf_counter()
        if self.world > 1 and (t is not self.affs or self.exchange_affs):
            # z faces first, then the y faces over the whole padded z extent: the margins just received travel on, which
            # fills the corners with the diagonal neighbour's data
            c, Z = self.ctx[0], self.shape[0]
            if c > 0:
                exchange_faces(t[..., c:2 * c, :, :], t[..., Z:Z + c, :, :], t[..., 0:c, :, :], t[..., Z + c:Z + 2 * c, :, :],
                               self.peers[0][0], self.peers[0][1], self.group)
            c, Y = self.ctx[1], self.shape[1]
            if c > 0:
                exchange_faces(t[..., :, c:2 * c, :], t[..., :, Y:Y + c, :], t[..., :, 0:c, :], t[..., :, Y + c:Y + 2 * c, :],
                               self.peers[1][0], self.peers[1][1], self.group)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.timers["exchange"] += time.perf_counter() - t_in
        return ev

    # -- stages --------------------------------------------------------------------------
    def _sync(self):
        """every lane drained and its workspace status read (which also clears it) -> the first overflow error, if any"""
        err = None
        for lane in self.lanes:
            lane["stream"].synchronize()
            try:
                lane["engine"].status()
            except _lib.BsmiError as exc:
                if exc.code != _lib.ERR_OVERFLOW:
                    raise
                err = err or exc
        return err

    @staticmethod
    def _only_edge_buffer(err):
        """is `err` nothing but "the caller's edge buffer is too small" (flag 32 of bsmi_seg_status)?"""
        import re
        m = re.search(r"flags 0x([0-9a-f]+)", str(err))
        return bool(m) and int(m.group(1), 16) == 32

    def _neighbours(self, k):
        """blocks whose write box touches the read box of block k (k itself included)"""
        cy, cx = self.counts[1], self.counts[2]
        iz, r = divmod(k, cy * cx)
        iy, ix = divmod(r, cx)
        out = []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    z, y, x = iz + dz, iy + dy, ix + dx
                    if 0 <= z < self.counts[0] and 0 <= y < cy and 0 <= x < cx:
                        out.append((z * cy + y) * cx + x)
        return out

    def _on_face(self, k):
        """does block k read context that belongs to another rank's slab?"""
        iz, r = divmod(k, self.counts[1] * self.counts[2])
        iy = r // self.counts[2]
        (zlo, zhi), (ylo, yhi) = self.peers
        return ((self.ctx[0] > 0 and ((zlo is not None and iz == 0) or (zhi is not None and iz == self.counts[0] - 1))) or
                (self.ctx[1] > 0 and ((ylo is not None and iy == 0) or (yhi is not None and iy == self.counts[1] - 1))))

    def _take_lane(self, kind, k):
        """The lane that runs task (`kind`, block k): the one with the least work queued so far (a lane is an in-order
        stream).  A fixed block -> lane map (k mod lanes) put a block's fragments AND its edge scoring on one lane: with 20
        blocks on 16 lanes, four lanes got four tasks and twelve got two, and the stage took four task times instead of the
        three that 40 tasks on 16 lanes need (tools/probe_volume.py: tail of the driver's job 145 -> 105 ms under the probe)."""
        i = min(range(min(len(self.lanes), self._lane_limit)), key=lambda j: (self._lane_load[j], j))
        self._lane_load[i] += 1
        self._lane_of[(kind, k)] = i
        return self.lanes[i]

    def lane_of(self, kind, k):
        """the lane task (kind in "f", "s"; block k) was queued on by the last run"""
        return self.lanes[self._lane_of[(kind, k)]]

    def _launch_fragments(self, k, wait=()):
        """post/blockwise/watershed_frags.py:196-246 for block k, asynchronous on its lane.  An all-zero read box yields
        no fragment, as the early return of the reference does."""
        b, e = self.boxes[k]
        lane = self._take_lane("f", k)
        wshape = tuple(hi - lo for lo, hi in zip(b, e))
        rshape = tuple(w + 2 * c for w, c in zip(wshape, self.ctx))
        with torch.cuda.stream(lane["stream"]):
            for ev in wait:
                lane["stream"].wait_event(ev)
            a = self._buf(lane["a"], rshape, (3,))
            a.copy_(self.affs[(slice(None),) + self._read_slices(k)])
            eng = lane["engine"]
            if any(v is not None for v in self.shift.values()):
                # watershed_frags.py:116-145: the watershed sees the shifted affinities, everything after it the plain ones
                from .post.shifts import boundary_mask_affinities
                gen = torch.Generator(device=self.dev).manual_seed(self.noise_seed + self.block_ids[k])
                src = boundary_mask_affinities(a, self.fragments_in_xy, dtype=torch.float64, generator=gen, min_seed_distance=self.msd,
                                               **self.shift)
            else:
                src = a
            fr, _ = eng.ws_fragments(src, self.fragments_in_xy, self.msd)
            if self.epsilon > 0:
                eng.rag_agglomerate(a, fr, self.epsilon, 256)   # watershed_frags.py:182-183
            lab = self._buf(lane["lab"], wshape)
            eng.postprocess_fragments(a, fr, self.filter_fragments, self.remove_debris, self.ctx, wshape,
                                      self.block_ids[k] * self.nvb, out=lab, num=self.nums[k:k + 1])
            self.frags[tuple(slice(c + lo, c + hi) for c, lo, hi in zip(self.ctx, b, e))].copy_(lab)
            eng.label_stats(lab, self.block_ids[k] * self.nvb, self.label_cap, size=self.sizes[k], sums=self.sums[k])
            self.frag_done[k] = torch.cuda.Event()
            self.frag_done[k].record(lane["stream"])

    def _launch_scores(self, k, wait=()):
        """post/blockwise/waterz_agglom.py:106-170 for block k, asynchronous on its lane"""
        b, e = self.boxes[k]
        lane = self._take_lane("s", k)
        rshape = tuple(hi - lo + 2 * c for lo, hi, c in zip(b, e, self.ctx))
        with torch.cuda.stream(lane["stream"]):
            for ev in wait:
                lane["stream"].wait_event(ev)
            a = self._buf(lane["a"], rshape, (3,))
            a.copy_(self.affs[(slice(None),) + self._read_slices(k)])
            f = self._buf(lane["f"], rshape)
            f.copy_(self.frags[self._read_slices(k)])
            if self.host_scores:   # the graph only: `_collect` scores all blocks' graphs on host threads
                lane["engine"].rag_graph_async(a, f, self.edges[k], self.esums[k], self.scores[k].view(torch.int32), self.counts_dev[k])
            else:
                lane["engine"].rag_merge_scores_async(a, f, 1.0, self.bins, self.edges[k], self.scores[k], self.counts_dev[k])

    def _collect(self):
        """end of the two block stages: one synchronisation, overflow checks, the edges every block owns (the block that
        created the smaller-id fragment, see post/blockwise.py) to the host"""
        err = self._sync()
        nums = self.nums.cpu().numpy()
        if nums.max(initial=0) >= self.nvb:   # ids are block id * voxels per block + label, as the reference's (watershed_frags.py:222-224)
            raise _lib.BsmiError(_lib.ERR_OVERFLOW, f"a block produced {int(nums.max())} fragments: not below its {self.nvb} voxels")
        n_edges = self.counts_dev[:, 0].cpu().numpy()
        if (err is None or self._only_edge_buffer(err)) and (nums.max(initial=0) > self.label_cap or n_edges.max(initial=0) > self.edge_cap):
            # the per-block tables were sized for typical blocks (label_cap fragments, edge_cap edges); a block needs more:
            # grow them to what was measured and redo that block's part only (the reference has no such limit)
            err = self._regrow(nums, n_edges)
        if err is not None:
            raise err
        self.block_nums = nums
        if self.host_scores:
            from .post.engine import rag_merge_scores_host
            m = int(n_edges.max(initial=0))
            if m == 0:
                self.rag_edges, self.rag_scores = np.zeros((0, 2), np.uint64), np.zeros(0, np.float32)
                return 0
            E = self.edges[:, :m].cpu().numpy().view(np.uint64)
            S = self.esums[:, :m].cpu().numpy()
            C = self.scores.view(torch.int32)[:, :m].cpu().numpy()
            sc = rag_merge_scores_host(n_edges, E, S, C, 1.0, self.bins, bins_formula=self.queue_bins_formula)
            # a block owns the edges whose smaller id is one of its own (first .. first + nvb - 1): one run of its sorted graph
            es, ss = [], []
            for k in range(len(self.boxes)):
                first = np.uint64(self.block_ids[k]) * np.uint64(self.nvb) + np.uint64(1)
                lo, hi = np.searchsorted(E[k, :n_edges[k], 0], [first, first + np.uint64(self.nvb)])
                es.append(E[k, lo:hi])
                ss.append(sc[k, lo:hi])
            self.rag_edges = np.concatenate(es)
            self.rag_scores = np.concatenate(ss)
            return len(self.rag_scores)
        ne = self.counts_dev[:, 0]
        take = torch.arange(self.edge_cap, device=self.dev)[None, :] < ne[:, None]
        bid = torch.tensor(self.block_ids, dtype=torch.int64, device=self.dev)[:, None].expand(-1, self.edge_cap)
        own = take & (torch.div(self.edges[:, :, 0] - 1, self.nvb, rounding_mode="floor") == bid)
        self.rag_edges = self.edges[own].cpu().numpy().view(np.uint64)
        self.rag_scores = self.scores[own].cpu().numpy()
        return len(self.rag_scores)

    def _regrow(self, nums, n_edges):
        """Blocks with more fragments than `label_cap` or more edges than `edge_cap`: larger tables (what the fullest block
        needs and a quarter more), node statistics / edge scoring of those blocks again.  -> a remaining overflow error or None"""
        K = len(self.boxes)
        if nums.max(initial=0) > self.label_cap:
            old, self.label_cap = self.label_cap, min(self.nvb, int(nums.max()) * 5 // 4 + 64)
            sizes = torch.zeros((K, self.label_cap), dtype=torch.int64, device=self.dev)
            sums = torch.zeros((K, self.label_cap, 3), dtype=torch.int64, device=self.dev)
            sizes[:, :old].copy_(self.sizes)
            sums[:, :old].copy_(self.sums)
            self.sizes, self.sums = sizes, sums
            # the new tables were filled on the current stream, the blocks' statistics are redone on the lanes' streams: without
            # this wait a lane could write a block's row before the zero fill / the copy of the old rows reached it
            torch.cuda.current_stream(self.dev).synchronize()
            for k in np.nonzero(nums > old)[0]:
                b, e = self.boxes[k]
                lane = self._take_lane("n", int(k))
                with torch.cuda.stream(lane["stream"]):
                    lab = self.frags[tuple(slice(c + lo, c + hi) for c, lo, hi in zip(self.ctx, b, e))].contiguous()
                    lane["engine"].label_stats(lab, self.block_ids[k] * self.nvb, self.label_cap, size=self.sizes[k], sums=self.sums[k])
        if n_edges.max(initial=0) > self.edge_cap:
            old, self.edge_cap = self.edge_cap, int(n_edges.max()) * 5 // 4 + 64
            edges = torch.empty((K, self.edge_cap, 2), dtype=torch.int64, device=self.dev)
            scores = torch.empty((K, self.edge_cap), dtype=torch.float32, device=self.dev)
            edges[:, :old].copy_(self.edges)
            scores[:, :old].copy_(self.scores)
            self.edges, self.scores = edges, scores
            if self.host_scores:
                esums = torch.empty((K, self.edge_cap), dtype=torch.int64, device=self.dev)
                esums[:, :old].copy_(self.esums)
                self.esums = esums
            torch.cuda.current_stream(self.dev).synchronize()   # (as above: the copies of the old rows before the lanes write new ones)
            for k in np.nonzero(n_edges > old)[0]:
                self._launch_scores(int(k))
        return self._sync()

    def run_blocks(self, ready=None, overlap=False):
        """Both block stages of the slab.  ready[k]: an event that fires when the affinities of blocks 0..k are in the
        slab (None: they all are).  overlap = False: stage by stage, once every block is predicted.  overlap = True: a
        block's fragments are launched as soon as the blocks its read box touches are predicted, its edge scoring as soon as
        their fragments exist -- launched by the host when the events have fired, never parked behind a device-side wait --
        so the lanes work while the predict stream still runs (measured on the benchmark, 64 blocks: 45.8 against 44.6
        Mvoxels/s end to end, but the predict stream's launches run 8 % slower beside the lanes -- its persistent conv
        workgroups want whole CUs -- so the roofline figure of the conv kernels drops from 0.57 to 0.49; not the default).  The blocks at a slab face shared with another rank wait for the exchange of that face."""
        K = len(self.boxes)
        self._lane_load = [0] * len(self.lanes)
        self._lane_of = {}
        self._lane_limit = len(self.lanes)
        last = [max(self._neighbours(k)) for k in range(K)]
        face = [k for k in range(K) if self._on_face(k)]
        inner = [k for k in range(K) if not self._on_face(k)]
        inner_set = set(inner)
        if ready is None:  # whatever filled the slab did so on the current stream
            here = torch.cuda.Event()
            here.record(torch.cuda.current_stream(self.dev))
            ready = [here] * K
        # Launch order = dependency order: a lane is an in-order stream, so a block's edge scoring is queued as soon as the
        # fragments of every block it reads have been queued, not behind fragments that wait for later predictions.
        nb = [self._neighbours(k) for k in range(K)]
        queued, scored = set(), set()

        def score_what_can_be(candidates, extra=()):
            for j in candidates:
                if j not in scored and all(i in queued for i in nb[j]):
                    self._launch_scores(j, list(extra) + [self.frag_done[i] for i in nb[j]])
                    scored.add(j)
        # Several ranks: a rank that fails before a face exchange must not leave its neighbours waiting in it (they would sit
        # there until the process-group timeout).  Every phase in front of an exchange is guarded; the ranks agree (all-reduce
        # of a flag) before each exchange and leave TOGETHER when one of them has failed -- into the accounted retry path of
        # run_blocks_accounted, where the exchanges are repeated by all.
        err = None

        def guarded(phase):
            nonlocal err
            if err is not None:
                return
            try:
                phase()
            except Exception as exc:  # noqa: BLE001
                from .blockwise import is_fatal
                if self.world == 1 or is_fatal(exc):
                    raise
                err = exc

        def launch_inner():
            for k in inner:
                self._launch_fragments(k, (ready[K - 1],))
                queued.add(k)
        def launch_overlapped():
            # Host-driven: a block's stage is launched once its inputs EXIST (event queries), so no lane ever sits behind
            # a device-side wait -- parked queues are polled by the command processor at the predict stream's expense.
            import time
            todo_f = list(inner)
            todo_s = [j for j in inner if all(i in inner_set for i in nb[j])]   # the others follow the face blocks below
            while todo_f or todo_s:
                moved = False
                # while blocks are still being predicted only a few lanes take tasks: every flood / merge workgroup keeps a
                # whole-CU conv workgroup off its CU, and a block per predict time is all the lanes need to keep up with
                self._lane_limit = len(self.lanes) if ready[K - 1].query() else max(1, min(self.overlap_lanes, len(self.lanes)))
                while todo_f and ready[last[todo_f[0]]].query():
                    k = todo_f.pop(0)
                    self._launch_fragments(k, ())
                    queued.add(k)
                    moved = True
                for j in list(todo_s):
                    if all(i in queued and self.frag_done[i].query() for i in nb[j]):
                        self._launch_scores(j, ())
                        scored.add(j)
                        todo_s.remove(j)
                        moved = True
                if not moved:
                    time.sleep(0.0002)
        if overlap:
            guarded(launch_overlapped)
            self._lane_limit = len(self.lanes)
        if not overlap:
            guarded(launch_inner)
        if face:
            guarded(lambda: ready[K - 1].synchronize())
            self._agree(err)
            got = self._exchange(self.affs)

            def launch_face():
                for k in face:
                    self._launch_fragments(k, (got,))
                    queued.add(k)
                    if overlap:
                        score_what_can_be([j for j in nb[k] if j in inner_set])
            guarded(launch_face)
        guarded(lambda: score_what_can_be(inner))
        if face:
            def await_outer():
                # the outermost blocks feed the neighbours' context -- on the faces that HAVE a neighbour (on a (world, 1) grid of
                # slabs the first and last block ROWS are nobody's context; waiting for them too held the exchange, and the
                # scoring queued behind it, until the whole fragments stage was over)
                (zlo, zhi), (ylo, yhi) = self.peers
                for k in range(K):
                    iz, r = divmod(k, self.counts[1] * self.counts[2])
                    iy = r // self.counts[2]
                    if ((zlo is not None and iz == 0) or (zhi is not None and iz == self.counts[0] - 1) or
                            (ylo is not None and iy == 0) or (yhi is not None and iy == self.counts[1] - 1)):
                        self.frag_done[k].synchronize()
            guarded(await_outer)
            self._agree(err)
            got = self._exchange(self.frags)
            score_what_can_be(face, (got,))
        if err is not None:   # no shared face (context 0 along every cut): nothing above agreed on, or raised, a guarded failure
            raise err
        return self._collect()

    def _agree(self, err):
        """In front of a collective step: has any rank failed so far?  Then every rank raises (the failed one its own error,
        the others a RuntimeError naming the situation) instead of some of them entering the exchange."""
        if self.world == 1:
            if err is not None:
                raise err
            return
        import torch.distributed as dist
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32,
                            device=self.dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            raise err if err is not None else RuntimeError("another rank failed in front of a face exchange: leaving the stage with it")

    def run_blocks_accounted(self, max_retries=None):
        """run_blocks with the reference's task accounting (blockwise.py:12-22, daisy retries): the stages first run
        as usual, all blocks in flight on the lanes; only if that fails somewhere are they run again block by block, each
        block awaited, retried and counted on its own -- a block of the scoring task that reads a failed fragments block
        is orphaned.  With several ranks they agree on which way to go (the face exchanges are collective).
        -> {task id: TaskState} of this rank's blocks."""
        import torch.distributed as dist
        from .blockwise import MAX_RETRIES, TaskState, run_blocks
        max_retries = MAX_RETRIES if max_retries is None else max_retries
        K = len(self.boxes)
        failed = 0
        try:
            self.run_blocks()
        except Exception as exc:  # noqa: BLE001
            from .blockwise import is_fatal
            if is_fatal(exc):
                raise
            failed = 1
        if self.world > 1:
            flag = torch.tensor([failed], dtype=torch.int32, device=self.dev if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            failed = int(flag.item())
        names = ("WatershedFrags", "WaterzAgglom")
        if not failed:
            states = {}
            for n in names:
                states[n] = TaskState(n, K)
                states[n].completed_count = K
            return states

        def one(launch, kind):
            def run(k):
                launch(k)
                lane = self.lane_of(kind, k)
                lane["stream"].synchronize()
                lane["engine"].status()
            return run
        here = self._exchange(self.affs)
        here.synchronize()
        st_f = run_blocks(names[0], list(range(K)), one(self._launch_fragments, "f"), max_retries)
        for k in st_f.failed_blocks:  # a failed block contributes no fragments
            b, e = self.boxes[k]
            self.frags[tuple(slice(c + lo, c + hi) for c, lo, hi in zip(self.ctx, b, e))].zero_()
            self.nums[k] = 0
        torch.cuda.synchronize(self.dev)
        self._exchange(self.frags).synchronize()
        self.counts_dev.zero_()
        torch.cuda.current_stream(self.dev).synchronize()   # before the lanes write their blocks' counts
        st_s = run_blocks(names[1], list(range(K)), one(self._launch_scores, "s"), max_retries, upstream_failed=st_f.failed_blocks,
                          depends_on=self._neighbours)
        self._collect()
        return {names[0]: st_f, names[1]: st_s}

    def node_table(self):
        """RAG nodes of the slab's blocks {id, position (voxels of the slab), size} (watershed_frags.py:230-246)."""
        nums = np.asarray([int(n) for n in self.block_nums], dtype=np.int64)
        if nums.sum() == 0:
            return np.zeros(0, np.uint64), np.zeros((0, 3)), np.zeros(0, np.int64)
        # only the used prefix of every block's table leaves the device (the tables are label_cap entries per block: 1 GB for the
        # 512 blocks of a 1024^3 volume, of which 30 MB are nodes)
        used = torch.arange(self.label_cap, device=self.dev)[None, :] < torch.from_numpy(nums).to(self.dev)[:, None]
        sizes = self.sizes[used].cpu().numpy()
        sums = self.sums[used].cpu().numpy().astype(np.float64)
        first = np.repeat(np.asarray([b for b, _ in self.boxes], np.float64), nums, axis=0)
        ids = np.concatenate([np.arange(1, n + 1, dtype=np.uint64) + np.uint64(bid * self.nvb) for n, bid in zip(nums, self.block_ids) if n])
        return ids, first + sums / sizes[:, None], sizes

    def ensure_outputs(self, one=False):
        """the stitch's buffers, if the constructor left them out: the interior copy of the fragments and all segmentations
        (one = False) or a single segmentation buffer (one = True)"""
        if self._fr is None:
            self._fr = torch.empty(self.shape, dtype=torch.int64, device=self.dev)
        if one and self._one is None:
            self._one = torch.empty(self.shape, dtype=torch.int64, device=self.dev)
        if not one and self.segs is None:
            self.segs = torch.empty((len(self.thresholds),) + self.shape, dtype=torch.int64, device=self.dev)

    def stitch(self, consume=None):
        """post/watershed.py:155-203: every rank's nodes and scored edges meet on rank 0, which runs the connected
        components per threshold; the LUT comes back and every rank relabels its slab.  -> segs [thresholds][Z][Y][X].
        consume(t, seg): the segmentations one at a time in ONE buffer (a driver that writes them out: 8 B per voxel of device
        memory instead of 8 B per voxel and threshold); `consume` must be done with the buffer when it returns."""
        nodes = np.concatenate([np.arange(1, int(n) + 1, dtype=np.uint64) + np.uint64(bid * self.nvb)
                                for n, bid in zip(self.block_nums, self.block_ids)] or [np.zeros(0, np.uint64)])
        # score < t  <=>  score <= the float32 just below t: the strict rule through the same library call
        thr = self.thresholds if self.cc_inclusive else [float(np.nextafter(np.float32(t), np.float32(-np.inf))) for t in self.thresholds]
        self.nodes, self.luts = gather_and_stitch(nodes, self.rag_edges, self.rag_scores, thr, self.rank, self.world, self.obj_group)
        self.ensure_outputs(one=consume is not None)
        fr = self._fr
        fr.copy_(self.interior(self.frags))
        keys = torch.from_numpy(self.nodes.view(np.int64)).to(self.dev)
        if consume is not None:
            for t, comp in enumerate(self.luts):
                lut_relabel(fr, keys, torch.from_numpy(comp.view(np.int64)).to(self.dev), out=self._one)
                torch.cuda.current_stream(self.dev).synchronize()
                consume(t, self._one)
            return None
        if 1 <= len(self.luts) <= 8:   # every threshold's LUT in one pass over the fragments
            from .post.engine import lut_relabel_multi
            lut_relabel_multi(fr, keys, torch.from_numpy(np.stack([c.view(np.int64) for c in self.luts])).to(self.dev), out=self.segs)
        else:
            for t, comp in enumerate(self.luts):
                lut_relabel(fr, keys, torch.from_numpy(comp.view(np.int64)).to(self.dev), out=self.segs[t])
        torch.cuda.synchronize(self.dev)
        return self.segs

    def prime(self):
        """Warm-up of the slab-sized paths behind the block stages (the torch reductions of `_collect`, the gather / LUT /
        relabel of `stitch`) on the still empty slab: their kernels are picked by tensor size, and the first use of one in a
        process loads it (on a fresh machine from disk) -- 0.1 s that would otherwise land in the first job."""
        # ... and of every lane's workspace: a lane's hash tables, heaps and scratch volumes are fresh allocations whose
        # first touch (page-table set-up) would otherwise fall into the first job -- 20-30 ms of the driver's 20-block job
        # on some boxes.  One block's two tasks on the still empty slab per lane: trivial work, every buffer touched.
        self.prime_lanes()
        self._collect()
        return self.stitch()

    def prime_lanes(self):
        """block 0's two tasks on every lane (whatever the slab holds): every lane's workspace touched, every kernel loaded"""
        if len(self.boxes):
            for i in range(len(self.lanes)):
                self._lane_load = [1] * len(self.lanes)
                self._lane_load[i] = 0
                self._launch_fragments(0)
                self._lane_load = [1] * len(self.lanes)
                self._lane_load[i] = 0
                self._launch_scores(0)
            self._lane_load = [0] * len(self.lanes)
            for lane in self.lanes:
                lane["stream"].synchronize()

    def run(self, ready=None, overlap=False):
        if ready is not None and not overlap:
            # Stage by stage: nothing is queued on the lanes before the last block is predicted.  Lanes parked behind a
            # wait on that event are 16 hardware queues the command processor keeps polling, and the predict stream's
            # launches pay for it: 42.9 against 39.1 ms per block (bench, 64 blocks), although the lanes do nothing.
            ready[-1].synchronize()
        import time
        self.timers = {"exchange": 0.0, "blocks": 0.0, "stitch": 0.0}
        t0 = time.perf_counter()
        self.run_blocks(ready, overlap)
        t1 = time.perf_counter()
        segs = self.stitch()
        self.timers["blocks"], self.timers["stitch"] = t1 - t0, time.perf_counter() - t1
        return segs


class VolumePipeline:
    """Predict + segment a box of blocks of a raw volume resident in HBM: this rank's slab of the job."""

    def __init__(self, model, out_block, net_context, job_blocks, seg_context=(16, 16, 16), thresholds=(0.2, 0.35, 0.5),
                 min_seed_distance=10, filter_fragments=0.0, remove_debris=0, n_lanes=20, device=0, rank=0, world=1,
                 group=None, job_origin=(0, 0, 0), segment=True, overlap=False, obj_group=None):
        """job_blocks: (layers per rank, blocks in y, blocks in x): the job is `world` such slabs stacked along z,
        its first voxel at `job_origin` of the raw volume.
        model: a Model, or a list of Models with the same weights = PREDICT LANES: block k is predicted by engine k mod K on that
        engine's own stream, so that the forward passes of K blocks overlap -- the memory-bound launches of one pass (Winograd
        transforms, pooling, the narrow stages) beside the matrix launches of another.  (Each engine has its own activation
        buffers; overlapping passes are safe since round 4)"""
        self.models = list(model) if isinstance(model, (list, tuple)) else [model]
        model = self.model = self.models[0]
        self.out_block = tuple(int(b) for b in out_block)
        self.net_context = tuple(int(c) for c in net_context)
        self.in_block = tuple(o + 2 * c for o, c in zip(self.out_block, self.net_context))
        if model.output_shape(self.in_block) != self.out_block:
            raise ValueError(f"network maps {self.in_block} to {model.output_shape(self.in_block)}, not {self.out_block}")
        self.job_blocks = tuple(int(g) for g in job_blocks)
        self.rank, self.world = int(rank), int(world)
        self.dev = torch.device("cuda", int(device))
        slab = tuple(g * b for g, b in zip(self.job_blocks, self.out_block))
        self.origin = (int(job_origin[0]) + self.rank * slab[0], int(job_origin[1]), int(job_origin[2]))
        self.pred_stream = predict_stream(self.dev)
        self.pred_streams = [predict_stream(self.dev, i) for i in range(len(self.models))]
        self.seg = SlabSegmenter(slab, self.out_block, seg_context if segment else (0, 0, 0), self.job_blocks[0] * self.world,
                                 self.job_blocks[0] * self.rank, thresholds, True, min_seed_distance, filter_fragments,
                                 remove_debris, 256, n_lanes if segment else 1, device, rank, world, group, obj_group=obj_group)
        self.segment = bool(segment)
        self.overlap = bool(overlap)
        self.t_predict = 0.0

    def predict(self, volume_u8):
        """models/3d_affs/predict.py:128-162 for every block of the slab: reflect-padded read, U-Net, uint8 affinities
        into the slab (first three channels: what the segmentation reads, post/watershed.py:70).  Asynchronous on the
        predict stream; -> one event per block."""
        from .unet import extract_block_reflect
        ready = []
        K = len(self.models)
        n = len(self.seg.boxes)
        self._t0 = torch.cuda.Event(enable_timing=True)
        self._t0.record(self.pred_stream)
        for lane in range(1, K):  # the other lanes start where the predict stream is now
            self.pred_streams[lane].wait_event(self._t0)
        for k, (b, _) in enumerate(self.seg.boxes):
            lane = k % K
            st = self.pred_streams[lane]
            with torch.cuda.stream(st):
                off = [o + lo - c for o, lo, c in zip(self.origin, b, self.net_context)]
                raw = extract_block_reflect(volume_u8, off, self.in_block)
                u8 = self.models[lane].predict_u8(raw)
                self.seg.write_view(k).copy_(u8[0][:3])
                last = k + 1 == n
                if last and K > 1:
                    # ready[-1] is also "every block is predicted" (run(), bench.py): recorded on the predict stream, after the
                    # last block of every lane
                    for j in range(max(0, n - K), n - 1):
                        self.pred_stream.wait_event(ready[j])
                    done = torch.cuda.Event()
                    done.record(st)
                    self.pred_stream.wait_event(done)
                    st = self.pred_stream
                ev = torch.cuda.Event(enable_timing=last)
                ev.record(st)
                ready.append(ev)
        return ready

    def run(self, volume_u8):
        """-> segs int64 [thresholds][Z][Y][X] of this rank's slab (None without segmentation).  t_predict: seconds until
        the last block was predicted (the lanes already segment meanwhile)."""
        ready = self.predict(volume_u8)
        segs = self.seg.run(ready, self.overlap) if self.segment else None
        ready[-1].synchronize()
        self.t_predict = self._t0.elapsed_time(ready[-1]) * 1e-3
        return segs
