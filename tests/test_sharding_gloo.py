"""N>1 path on CPU (gloo): the workers' block shards (bootstrapper_amd.predict.rank_blocks: contiguous runs of the z-major
block list, disjoint and covering), the max-over-ranks timing reduction of bench.py, the gradient sum of the training
step, and the communication of the volume segmentation (face exchange, edge gather, LUT broadcast)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bootstrapper_amd.predict import enumerate_blocks, rank_blocks
    cfg = {"output_roi": ([0, 0, 0], [125 * 40, 1250 * 4, 1250 * 4]), "voxel_size": [40, 4, 4], "output_shape": [4, 320, 320]}
    blocks = enumerate_blocks(cfg)
    shard = rank_blocks(blocks, rank, world)
    # exchange shard sizes + a checksum of block ids, and reduce a fake per-rank time with MAX (bench.py's reduction)
    ids = torch.tensor([sum(hash(b) % 1000003 for b in shard), len(shard), blocks.index(shard[0])], dtype=torch.int64)
    gathered = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, shard, [g.tolist() for g in gathered], float(t.item()), blocks))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rank_block_sharding(world):
    """Every block goes to exactly one rank; a rank's blocks are one contiguous run of the z-major list (it reads only the
    slab of the input that run needs); run lengths differ by at most one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    blocks = res[0][4]
    assert len(blocks) == 32 * 4 * 4   # (125, 1250, 1250) voxels in (4, 320, 320) blocks, fit = overhang
    joined = []
    for rank, shard, gathered, t, _ in res:
        assert gathered == res[0][2] and t == float(world)
        assert shard == blocks[gathered[rank][2]:gathered[rank][2] + len(shard)]   # contiguous
        joined += shard
    assert joined == blocks
    sizes = [len(r[1]) for r in res]
    assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bootstrapper_amd.training import sum_gradients
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    scale = sum_gradients(g)
    q.put((rank, g.tolist(), scale))
    dist.destroy_process_group()


def test_gradient_sum_two_ranks_gloo():
    """the data-parallel reduction of the training step (bootstrapper_amd.training.sum_gradients), world size 2"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, 29633, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g, scale in res:
        assert g == [3.0 * i for i in range(10)] and scale == 0.5


def _volume_worker(rank, world, port, q):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bootstrapper_amd.volume import exchange_faces, gather_and_stitch, slab_layers
    # face exchange: a (C, Z + 2c, Y, X) slab with context c = 2; the margins must receive the neighbours' outer layers
    c, Z = 2, 6
    vol = torch.arange(3 * world * Z * 4 * 5, dtype=torch.int64).reshape(3, world * Z, 4, 5)
    slab = torch.zeros(3, Z + 2 * c, 4, 5, dtype=torch.int64)
    slab[:, c:c + Z] = vol[:, rank * Z:(rank + 1) * Z]
    exchange_faces(slab[:, c:2 * c], slab[:, Z:Z + c], slab[:, 0:c], slab[:, Z + c:Z + 2 * c], rank - 1 if rank > 0 else None,
                   rank + 1 if rank < world - 1 else None)
    want = torch.zeros_like(slab)
    lo, hi = max(rank * Z - c, 0), min((rank + 1) * Z + c, world * Z)
    want[:, lo - (rank * Z - c):hi - (rank * Z - c)] = vol[:, lo:hi]
    faces_ok = bool(torch.equal(slab, want))
    # stitching: a chain of fragments cut over the ranks; the components must come out as if one rank had it all
    rng = np.random.default_rng(5)
    n = 40
    nodes = np.arange(1, n + 1, dtype=np.uint64) * 3
    edges = np.stack([nodes[:-1], nodes[1:]], axis=1)
    extra = rng.integers(0, n, size=(30, 2))
    extra = np.sort(extra[extra[:, 0] != extra[:, 1]], axis=1)
    edges = np.concatenate([edges, nodes[extra]])
    scores = rng.random(len(edges)).astype(np.float32)
    scores[::7] = np.nan                                  # unscored edges are dropped (post/watershed.py:163-171)
    starts, counts = slab_layers(n, world)
    mine = slice(starts[rank], starts[rank] + counts[rank])
    own = (edges[:, 0] >= nodes[mine][0]) & (edges[:, 0] <= nodes[mine][-1])      # an edge travels with its smaller fragment
    got_nodes, got = gather_and_stitch(nodes[mine], edges[own], scores[own], [0.3, 0.6], rank, world)
    one_nodes, one = gather_and_stitch(nodes, edges, scores, [0.3, 0.6])
    q.put((rank, faces_ok, bool(np.array_equal(got_nodes, one_nodes)), [bool(np.array_equal(a, b)) for a, b in zip(got, one)],
           [int(len(np.unique(a))) for a in one]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_exchange_and_stitch_gloo(world):
    """The communication of bootstrapper_amd.volume on CPU tensors: slab faces to the z-neighbours, edges to rank 0, LUT back."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + world
    ps = [ctx.Process(target=_volume_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, faces_ok, nodes_ok, comps_ok, ncomp in res:
        assert faces_ok and nodes_ok and all(comps_ok), (rank, faces_ok, nodes_ok, comps_ok)
        assert 1 < ncomp[1] < ncomp[0] < 40


def test_slab_layers_and_rank_grid():
    from bootstrapper_amd.volume import slab_layers, rank_grid
    assert slab_layers(8, 1) == ([0], [8])
    assert slab_layers(8, 3) == ([0, 3, 6], [3, 3, 2])
    assert slab_layers(2, 4) == ([0, 1, 2, 2], [1, 1, 0, 0])
    # ranks over block layers x block rows: every started rank has blocks; a flat volume is cut along y
    assert rank_grid(2, 4, 3) == (2, 1) and rank_grid(8, 8, 8) == (8, 1)
    assert rank_grid(8, 1, 10) == (1, 8)          # CREMI-shaped: one layer of 10 x 10 blocks
    assert rank_grid(8, 3, 10) == (2, 4)          # three layers on eight GPUs: nobody idle
    assert rank_grid(8, 1, 3) == (1, 3) and rank_grid(7, 2, 2) == (2, 2) and rank_grid(1, 5, 5) == (1, 1)


def _grid_worker(rank, world, port, q, grid):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bootstrapper_amd.volume import exchange_faces, slab_layers
    # a (Z, Y, X) volume cut into grid[0] x grid[1] boxes with context c: z faces first, then y faces over the padded z extent
    c, Z, Y, X = 2, 4, 6, 5
    gz, gy = grid
    vol = torch.arange(gz * Z * gy * Y * X, dtype=torch.int64).reshape(gz * Z, gy * Y, X) + 1
    rz, ry = divmod(rank, gy)
    box = torch.zeros(Z + 2 * c, Y + 2 * c, X, dtype=torch.int64)
    box[c:c + Z, c:c + Y] = vol[rz * Z:(rz + 1) * Z, ry * Y:(ry + 1) * Y]
    zlo, zhi = (rank - gy if rz > 0 else None), (rank + gy if rz < gz - 1 else None)
    ylo, yhi = (rank - 1 if ry > 0 else None), (rank + 1 if ry < gy - 1 else None)
    exchange_faces(box[c:2 * c], box[Z:Z + c], box[0:c], box[Z + c:Z + 2 * c], zlo, zhi)
    exchange_faces(box[:, c:2 * c], box[:, Y:Y + c], box[:, 0:c], box[:, Y + c:Y + 2 * c], ylo, yhi)
    want = torch.zeros_like(box)
    padded = torch.zeros(gz * Z + 2 * c, gy * Y + 2 * c, X, dtype=torch.int64)
    padded[c:-c, c:-c] = vol
    want.copy_(padded[rz * Z:rz * Z + Z + 2 * c, ry * Y:ry * Y + Y + 2 * c])
    q.put((rank, bool(torch.equal(box, want))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grid", [(2, 2), (1, 3), (3, 2), (8, 1), (2, 4)])   # the last two: the eight ranks of one MI355X node
def test_face_exchange_on_a_rank_grid_gloo(grid):
    """The two-phase face exchange of SlabSegmenter on a (Rz, Ry) grid of ranks: every box ends with its whole context
    margin, edges and corners (the diagonal neighbour's voxels) included, zeros beyond the volume."""
    world = grid[0] * grid[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 50 + 7 * world
    ps = [ctx.Process(target=_grid_worker, args=(r, world, port, q, grid)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
