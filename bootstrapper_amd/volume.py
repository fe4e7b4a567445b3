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


def exchange_faces(low_out, high_out, low_in, high_in, rank, world, group=None):
    """Send this rank's first / last layers to the z-neighbours and receive theirs: `low_out` goes to rank - 1 and
    arrives there as `high_in`, `high_out` goes to rank + 1 and arrives as its `low_in`.  Device tensors travel over
    RCCL point to point; under gloo (CPU tests, rehearsals) they are staged through host memory."""
    import torch.distributed as dist
    if world == 1:
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
    if rank > 0:
        post("send", low_out, rank - 1)
        post("recv", low_in, rank - 1)
    if rank < world - 1:
        post("send", high_out, rank + 1)
        post("recv", high_in, rank + 1)
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, buf in recvs:
        dst.copy_(buf)


def stitch_components(nodes, edges, scores, thresholds):
    """post/watershed.py:155-186 on the host: drop unscored edges, connected components per threshold.
    -> list of component ids aligned with `nodes` (ascending uint64)."""
    from .post.watershed import connected_components
    keep = ~np.isnan(scores)
    edges, scores = edges[keep], scores[keep]
    out = []
    for t in thresholds:
        out.append(nodes.copy() if edges.shape[0] == 0 else connected_components(nodes, edges, scores, t))
    return out


def gather_and_stitch(nodes, edges, scores, thresholds, rank=0, world=1, group=None):
    """The one many-to-one step of the path: every rank's fragment ids (ascending; ranks hold ascending id ranges) and
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
            raise ValueError("fragment ids of the ranks are not in ascending order")
        luts[0] = (nodes_all, stitch_components(nodes_all, edges_all, scores_all, thresholds))
    if world > 1:
        dist.broadcast_object_list(luts, src=0, group=group)
    return luts[0]


class SlabSegmenter:
    """Fragments, RAG scoring, stitching and relabelling of one rank's slab of affinities (see the module docstring)."""

    def __init__(self, slab_shape, block, context, total_layers, layer0, thresholds=(0.2, 0.35, 0.5),
                 fragments_in_xy=True, min_seed_distance=10, filter_fragments=0.0, remove_debris=0, discretize_queue=256,
                 n_lanes=8, device=0, rank=0, world=1, group=None, edge_cap=1 << 17, label_cap=1 << 16):
        self.shape = tuple(int(s) for s in slab_shape)
        self.block = tuple(int(b) for b in block)
        self.ctx = tuple(int(c) for c in context)
        self.thresholds = [float(t) for t in thresholds]
        self.fragments_in_xy, self.msd = bool(fragments_in_xy), int(min_seed_distance)
        self.filter_fragments, self.remove_debris = float(filter_fragments), int(remove_debris)
        self.bins = int(discretize_queue)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.dev = torch.device("cuda", int(device))
        self.boxes = shrink_blocks(self.shape, self.block)
        counts = [-(-s // b) for s, b in zip(self.shape, self.block)]
        # global z-major block ids (`block.block_id` of the reference's tasks; daisy numbers differently)
        self.block_ids = [((int(layer0) + iz) * counts[1] + iy) * counts[2] + ix
                          for iz in range(counts[0]) for iy in range(counts[1]) for ix in range(counts[2])]
        self.total_blocks = int(total_layers) * counts[1] * counts[2]
        self.nvb = int(np.prod(self.block))
        padded = tuple(s + 2 * c for s, c in zip(self.shape, self.ctx))
        self.affs = torch.zeros((3,) + padded, dtype=torch.uint8, device=self.dev)
        self.frags = torch.zeros(padded, dtype=torch.int64, device=self.dev)
        self.segs = None
        K = len(self.boxes)
        self.edge_cap, self.label_cap = int(edge_cap), int(label_cap)
        self.nums = torch.zeros(K, dtype=torch.int64, device=self.dev)
        self.sizes = torch.zeros((K, self.label_cap), dtype=torch.int64, device=self.dev)
        self.sums = torch.zeros((K, self.label_cap, 3), dtype=torch.int64, device=self.dev)
        self.edges = torch.empty((K, self.edge_cap, 2), dtype=torch.int64, device=self.dev)
        self.scores = torch.empty((K, self.edge_cap), dtype=torch.float32, device=self.dev)
        self.counts = torch.zeros((K, 4), dtype=torch.int64, device=self.dev)
        read = tuple(min(b, s) + 2 * c for b, s, c in zip(self.block, self.shape, self.ctx))
        self.lanes = []
        for _ in range(max(1, min(int(n_lanes), K))):
            self.lanes.append(dict(engine=SegEngine(read, self.dev.index), stream=torch.cuda.Stream(self.dev),
                                   a=torch.empty((3,) + read, dtype=torch.uint8, device=self.dev),
                                   f=torch.empty(read, dtype=torch.int64, device=self.dev),
                                   lab=torch.empty(tuple(min(b, s) for b, s in zip(self.block, self.shape)), dtype=torch.int64,
                                                   device=self.dev)))
        self.nodes = None
        self.rag_edges = self.rag_scores = None
        self.luts = None

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
        """context margins of `t` ([..., Zp, Yp, Xp]) at the slab's z faces <- the neighbours' outermost layers"""
        c, Z = self.ctx[0], self.shape[0]
        if self.world == 1 or c == 0:
            return
        torch.cuda.synchronize(self.dev)
        exchange_faces(t[..., c:2 * c, :, :], t[..., Z:Z + c, :, :], t[..., 0:c, :, :], t[..., Z + c:Z + 2 * c, :, :],
                       self.rank, self.world, self.group)

    # -- stages --------------------------------------------------------------------------
    def _after(self, event):
        for lane in self.lanes:
            if event is not None:
                lane["stream"].wait_event(event)

    def _sync(self):
        for lane in self.lanes:
            lane["stream"].synchronize()
            lane["engine"].status()

    def fragments(self, after=None):
        """post/blockwise/watershed_frags.py:196-246 for every block of the slab (asynchronous on the lanes; ends
        synchronised).  An all-zero read box yields no fragment, as the early return of the reference does."""
        self._exchange(self.affs)
        self._after(after)
        for k, (b, e) in enumerate(self.boxes):
            lane = self.lanes[k % len(self.lanes)]
            wshape = tuple(hi - lo for lo, hi in zip(b, e))
            rshape = tuple(w + 2 * c for w, c in zip(wshape, self.ctx))
            with torch.cuda.stream(lane["stream"]):
                a = self._buf(lane["a"], rshape, (3,))
                a.copy_(self.affs[(slice(None),) + self._read_slices(k)])
                eng = lane["engine"]
                fr, _ = eng.ws_fragments(a, self.fragments_in_xy, self.msd)
                lab = self._buf(lane["lab"], wshape)
                eng.postprocess_fragments(a, fr, self.filter_fragments, self.remove_debris, self.ctx, wshape,
                                          self.block_ids[k] * self.nvb, out=lab, num=self.nums[k:k + 1])
                self.frags[tuple(slice(c + lo, c + hi) for c, lo, hi in zip(self.ctx, b, e))].copy_(lab)
                eng.label_stats(lab, self.block_ids[k] * self.nvb, self.label_cap, size=self.sizes[k], sums=self.sums[k])
        self._sync()
        nums = self.nums.cpu().numpy()
        if nums.max(initial=0) > self.label_cap or nums.max(initial=0) >= self.nvb:
            raise _lib.BsmiError(_lib.ERR_OVERFLOW, f"a block produced {int(nums.max())} fragments (label_cap {self.label_cap})")
        self.block_nums = nums
        return nums

    def score_edges(self):
        """post/blockwise/waterz_agglom.py:106-170 for every block; keeps the edges each block owns (the block that
        created the smaller-id fragment, see post/blockwise.py).  Ends synchronised; -> number of edges kept."""
        self._exchange(self.frags)
        for k, (b, e) in enumerate(self.boxes):
            lane = self.lanes[k % len(self.lanes)]
            rshape = tuple(hi - lo + 2 * c for lo, hi, c in zip(b, e, self.ctx))
            with torch.cuda.stream(lane["stream"]):
                a = self._buf(lane["a"], rshape, (3,))
                a.copy_(self.affs[(slice(None),) + self._read_slices(k)])
                f = self._buf(lane["f"], rshape)
                f.copy_(self.frags[self._read_slices(k)])
                lane["engine"].rag_merge_scores_async(a, f, 1.0, self.bins, self.edges[k], self.scores[k], self.counts[k])
        self._sync()
        ne = self.counts[:, 0]
        take = torch.arange(self.edge_cap, device=self.dev)[None, :] < ne[:, None]
        bid = torch.tensor(self.block_ids, dtype=torch.int64, device=self.dev)[:, None].expand(-1, self.edge_cap)
        own = take & (torch.div(self.edges[:, :, 0] - 1, self.nvb, rounding_mode="floor") == bid)
        self.rag_edges = self.edges[own].cpu().numpy().view(np.uint64)
        self.rag_scores = self.scores[own].cpu().numpy()
        return len(self.rag_scores)

    def node_table(self):
        """RAG nodes of the slab's blocks {id, position (voxels of the slab), size} (watershed_frags.py:230-246)."""
        ids, pos, size = [], [], []
        sizes = self.sizes.cpu().numpy()
        sums = self.sums.cpu().numpy()
        for k, (b, _) in enumerate(self.boxes):
            n = int(self.block_nums[k])
            if n == 0:
                continue
            ids.append(np.arange(1, n + 1, dtype=np.uint64) + np.uint64(self.block_ids[k] * self.nvb))
            size.append(sizes[k, :n])
            pos.append(np.asarray(b, np.float64) + sums[k, :n].astype(np.float64) / sizes[k, :n, None])
        if not ids:
            return np.zeros(0, np.uint64), np.zeros((0, 3)), np.zeros(0, np.int64)
        return np.concatenate(ids), np.concatenate(pos), np.concatenate(size)

    def stitch(self):
        """post/watershed.py:155-203: every rank's nodes and scored edges meet on rank 0, which runs the connected
        components per threshold; the LUT comes back and every rank relabels its slab.  -> segs [thresholds][Z][Y][X]."""
        nodes = np.concatenate([np.arange(1, int(n) + 1, dtype=np.uint64) + np.uint64(bid * self.nvb)
                                for n, bid in zip(self.block_nums, self.block_ids)] or [np.zeros(0, np.uint64)])
        self.nodes, self.luts = gather_and_stitch(nodes, self.rag_edges, self.rag_scores, self.thresholds, self.rank, self.world,
                                                  self.group)
        fr = self.interior(self.frags).contiguous()
        keys = torch.from_numpy(self.nodes.view(np.int64)).to(self.dev)
        if self.segs is None:
            self.segs = torch.empty((len(self.thresholds),) + self.shape, dtype=torch.int64, device=self.dev)
        for t, comp in enumerate(self.luts):
            lut_relabel(fr, keys, torch.from_numpy(comp.view(np.int64)).to(self.dev), out=self.segs[t])
        torch.cuda.synchronize(self.dev)
        return self.segs

    def run(self, after=None):
        self.fragments(after)
        self.score_edges()
        return self.stitch()


class VolumePipeline:
    """Predict + segment a box of blocks of a raw volume resident in HBM: this rank's slab of the job."""

    def __init__(self, model, out_block, net_context, job_blocks, seg_context=(16, 16, 16), thresholds=(0.2, 0.35, 0.5),
                 min_seed_distance=10, filter_fragments=0.0, remove_debris=0, n_lanes=16, device=0, rank=0, world=1,
                 group=None, job_origin=(0, 0, 0), segment=True):
        """job_blocks: (layers per rank, blocks in y, blocks in x): the job is `world` such slabs stacked along z,
        its first voxel at `job_origin` of the raw volume."""
        self.model = model
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
        self.pred_stream = torch.cuda.Stream(self.dev, priority=-1)
        self.seg = SlabSegmenter(slab, self.out_block, seg_context if segment else (0, 0, 0), self.job_blocks[0] * self.world,
                                 self.job_blocks[0] * self.rank, thresholds, True, min_seed_distance, filter_fragments,
                                 remove_debris, 256, n_lanes if segment else 1, device, rank, world, group)
        self.segment = bool(segment)
        self.t_predict = self.t_segment = 0.0

    def predict(self, volume_u8):
        """models/3d_affs/predict.py:128-162 for every block of the slab: reflect-padded read, U-Net, uint8 affinities
        into the slab (first three channels: what the segmentation reads, post/watershed.py:70)."""
        from .unet import extract_block_reflect
        with torch.cuda.stream(self.pred_stream):
            for k, (b, _) in enumerate(self.seg.boxes):
                off = [o + lo - c for o, lo, c in zip(self.origin, b, self.net_context)]
                raw = extract_block_reflect(volume_u8, off, self.in_block)
                u8 = self.model.predict_u8(raw)
                self.seg.write_view(k).copy_(u8[0][:3])
            done = torch.cuda.Event()
            done.record(self.pred_stream)
        return done

    def run(self, volume_u8):
        import time
        t0 = time.perf_counter()
        done = self.predict(volume_u8)
        if not self.segment:
            self.pred_stream.synchronize()
            self.t_predict = time.perf_counter() - t0
            return None
        done.synchronize()
        t1 = time.perf_counter()
        segs = self.seg.run()
        t2 = time.perf_counter()
        self.t_predict, self.t_segment = t1 - t0, t2 - t1
        return segs
