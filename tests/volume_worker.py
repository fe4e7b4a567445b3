"""Worker of tests/test_volume_gpu.py::test_ranks_equal_one_rank (one process per rank, backend gloo, both on cuda:0)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from bootstrapper_amd.volume import SlabSegmenter, slab_layers
    from tests.test_volume_gpu import blobby_affs
    shape, block, ctx, thr = (32, 96, 80), (8, 32, 32), (2, 4, 4), [0.3, 0.45]
    affs = blobby_affs(shape, 33)
    layers, rows = shape[0] // block[0], shape[1] // block[1]
    # the ranks as a grid over block layers x block rows: "RzxRy" (default: slabs of layers)
    grid = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (world, 1)
    rz, ry = divmod(rank, grid[1])
    zs, zc = slab_layers(layers, grid[0])
    ys, yc = slab_layers(rows, grid[1])
    z0, z1 = zs[rz] * block[0], (zs[rz] + zc[rz]) * block[0]
    y0, y1 = ys[ry] * block[1], (ys[ry] + yc[ry]) * block[1]
    seg = SlabSegmenter((z1 - z0, y1 - y0, shape[2]), block, ctx, layers, zs[rz], thr, True, 4, 0.35, 12, 256, n_lanes=4, rank=rank,
                        world=world, grid=grid, total_rows=rows, row0=ys[ry])
    seg.interior(seg.affs).copy_(torch.from_numpy(affs[:, z0:z1, y0:y1]).cuda())
    segs = seg.run()
    mine = (seg.interior(seg.frags).cpu().numpy(), segs.cpu().numpy(), seg.luts, (z0, z1, y0, y1))
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        from oracle.blockwise_ref import cpu_blockwise
        frags2 = np.zeros(shape, dtype=np.int64)
        segs2 = np.zeros((len(thr),) + shape, dtype=np.int64)
        for f, sg, _, (a0, a1, b0, b1) in parts:
            frags2[a0:a1, b0:b1] = f
            segs2[:, a0:a1, b0:b1] = sg
        one = SlabSegmenter(shape, block, ctx, layers, 0, thr, True, 4, 0.35, 12, 256, n_lanes=4)
        one.interior(one.affs).copy_(torch.from_numpy(affs).cuda())
        segs1 = one.run().cpu().numpy()
        frags_ref, _, _, _, segs_ref = cpu_blockwise(affs, block, ctx, 4, 0.35, 12, thr)
        verdict = {"frags_equal": bool(np.array_equal(frags2, one.interior(one.frags).cpu().numpy())),
                   "segs_equal": bool(np.array_equal(segs2, segs1)),
                   "cpu_equal": bool(np.array_equal(frags2.view(np.uint64), frags_ref) and
                                     all(np.array_equal(segs2[t].view(np.uint64), segs_ref[t]) for t in range(len(thr)))),
                   "luts_equal": bool(all(np.array_equal(a, b) for a, b in zip(parts[1][2], one.luts)))}
        with open(os.path.join(out_dir, "verdict.json"), "w") as f:
            json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
