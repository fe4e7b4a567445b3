"""N>1 path on CPU: two gloo ranks shard the block list exactly as the GPU workers do
(bootstrapper_amd.predict.predict_blocks: blocks[rank::world]; bench.py: grid[(rank + i*world) % len]),
the shards are disjoint and cover the volume, and the max-over-ranks timing reduction works.
No collective touches the data path."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bootstrapper_amd.pipeline import block_grid
    from bootstrapper_amd.predict import enumerate_blocks
    grid = block_grid((512, 512, 512), (128, 128, 128))
    steps = len(grid) // world
    mine = [grid[(rank + i * world) % len(grid)] for i in range(steps)]
    cfg = {"output_roi": ([0, 0, 0], [125 * 40, 1250 * 4, 1250 * 4]), "voxel_size": [40, 4, 4], "output_shape": [4, 320, 320]}
    blocks = enumerate_blocks(cfg)
    shard = blocks[rank::world]
    # exchange shard sizes + a checksum of block ids, and reduce a fake per-rank time with MAX
    ids = torch.tensor([sum(hash(b) % 1000003 for b in mine), len(mine), len(shard)], dtype=torch.int64)
    gathered = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, mine, shard, [g.tolist() for g in gathered], float(t.item()), len(grid), len(blocks)))
    dist.destroy_process_group()


def test_two_rank_block_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, m0, s0, g0, t0, ngrid, nblocks), (_, m1, s1, g1, t1, _, _) = res
    assert not set(m0) & set(m1) and len(set(m0) | set(m1)) == ngrid == 64
    assert not set(s0) & set(s1) and len(s0) + len(s1) == nblocks == 512
    assert g0 == g1 and t0 == t1 == 2.0


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
    exchange_faces(slab[:, c:2 * c], slab[:, Z:Z + c], slab[:, 0:c], slab[:, Z + c:Z + 2 * c], rank, world)
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


@pytest.mark.parametrize("world", [2, 3])
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


def test_slab_layers():
    from bootstrapper_amd.volume import slab_layers
    assert slab_layers(8, 1) == ([0], [8])
    assert slab_layers(8, 3) == ([0, 3, 6], [3, 3, 2])
    assert slab_layers(2, 4) == ([0, 1, 2, 2], [1, 1, 0, 0])
