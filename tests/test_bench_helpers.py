"""bench.py's checkers (CPU): the id regridding and the scored-edge comparison that hold the GPU pipeline to the CPU restatement
inside the benchmark must themselves be right -- a checker that cannot fail checks nothing."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_regrid_ids_and_edge_check():
    B = _bench()
    nv = int(np.prod(B.OUT_BLOCK))
    sub, job = (3, 3, 3), (8, 8, 8)
    # id of label l in block (z, y, x) of the sub-box run -> id of the same block in the job's grid
    for (z, y, x, l) in [(0, 0, 0, 1), (2, 1, 0, 77), (1, 2, 2, nv), (0, 2, 1, 5)]:
        cid = np.uint64(((z * 3 + y) * 3 + x) * nv + l)
        gid = np.uint64(((z * 8 + y) * 8 + x) * nv + l)
        assert B.regrid_ids(np.array([cid]), sub, job)[0] == gid
    assert B.regrid_ids(np.array([0], np.uint64), sub, job)[0] == 0
    # job_blocks_for: the box of blocks of a run
    assert B.job_blocks_for(20) == (5, 2, 2)
    assert np.prod(B.job_blocks_for(64)) == 64 and np.prod(B.job_blocks_for(64, 8)) == 64 and B.job_blocks_for(64, 8) == (1, 8, 8)
    # scored edges: block (0,0,0) of a 3x3x3 sub-box is the one whose 26 neighbours have the same read box in both runs
    rng = np.random.default_rng(0)
    def edges_of(block_idx, grid, n):
        b = ((block_idx[0] * grid[1] + block_idx[1]) * grid[2] + block_idx[2]) * nv
        u = rng.integers(1, 1000, n).astype(np.uint64) + np.uint64(b)
        return np.stack([u, u + np.uint64(1 + 7)], axis=1)
    lab = (rng.permutation(999)[:40] + 1).astype(np.uint64)                   # distinct pairs
    cpu_e = np.stack([lab, lab + np.uint64(9)], axis=1)                       # block (0,0,0) in both grids: ids coincide
    other = np.stack([lab + np.uint64(((1 * 3 + 1) * 3 + 1) * nv), lab + np.uint64(((1 * 3 + 1) * 3 + 1) * nv + 3)], axis=1)   # block (1,1,1) of the sub grid: not compared
    cpu_edges = np.concatenate([cpu_e, other])
    cpu_scores = rng.random(len(cpu_edges)).astype(np.float32)
    cpu_scores[3] = np.nan
    gpu_edges, gpu_scores = cpu_e[::-1].copy(), cpu_scores[:40][::-1].copy()   # another order: the check sorts
    blocks, n, bad = B.check_edges(gpu_edges, gpu_scores, cpu_edges, cpu_scores, sub, job)
    assert (blocks, n, bad) == (1, 40, 0)
    gpu_scores[5] = np.nextafter(gpu_scores[5], np.float32(2))               # one score off by one ulp: caught
    assert B.check_edges(gpu_edges, gpu_scores, cpu_edges, cpu_scores, sub, job)[2] == 1
    assert B.check_edges(gpu_edges[1:], cpu_scores[:40][::-1][1:], cpu_edges, cpu_scores, sub, job)[2] == 1   # a missing edge: caught
    # fragments: equal after an id remap, and a difference is seen
    a = rng.integers(0, 5, (128, 128, 128)).astype(np.uint64)
    perm = np.array([0, 9, 7, 5, 3], np.uint64)
    assert B.same_partition(a, perm[a]) and not B.same_partition(a, np.where(a == 4, np.uint64(3), a))
