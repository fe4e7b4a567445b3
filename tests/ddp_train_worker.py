"""Worker of tests/test_train_gpu.py::test_data_parallel_two_ranks (one process per rank, backend gloo, both on cuda:0)."""
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
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.training import Trainer
    d = np.load(os.path.join(ROOT, "tests", "golden", "train_affs_f4i2.npz"))
    meta = json.loads(bytes(d["config"]).decode())
    sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    nc = {"in_channels": 1, "num_fmaps": meta["num_fmaps"], "fmap_inc_factor": meta["fmap_inc_factor"],
          "downsample_factors": [[1, 2, 2]] * 3, "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
          "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3, "outputs": {"3d_affs": {"dims": 6}}}
    m = Model(nc, precision="f32").load_state_dict(sd)
    tr = Trainer(m, meta["in_shape"], lr=1e-3)
    raw = torch.from_numpy(d["x"]).cuda()
    gt = torch.from_numpy(d["gt0"][0]).cuda()
    w = torch.from_numpy(d["w0"][0]).cuda()
    if rank == 1:  # a different sample on the second rank
        raw = raw.flip(2)
        gt = 1 - gt
    tr.forward_backward(raw, [gt], [w])
    local = {k: tr.read(k, "grad") for k in sd}
    tr.optimizer_step()
    summed = {k: tr.read(k, "grad") for k in sd}
    params = {k: tr.read(k, "param") for k in sd}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{"l:" + k: v for k, v in local.items()},
             **{"s:" + k: v for k, v in summed.items()}, **{"p:" + k: v for k, v in params.items()})
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
