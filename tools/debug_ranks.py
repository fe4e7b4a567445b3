"""Dev tool: VolumePipeline (real net, predict + segment) on a 4x2x2-block job as one rank and as two ranks (gloo, sharing cuda:0):
are the affinities, fragments and segmentations the same?   one rank:  python tools/debug_ranks.py OUT
two ranks: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 tools/debug_ranks.py OUT"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import bench
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.volume import VolumePipeline
out = sys.argv[1]
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
torch.cuda.set_device(0)
model = Model(bench.NET_CONFIG, device=0, precision="bf16x3").load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
L = int(os.environ.get("LAYERS", "4"))
vol = synthetic_volume((128 * L, 256, 256), seed=0, device=torch.device("cuda", 0))
job = (L // world, 2, 2)
pipe = VolumePipeline(model, bench.OUT_BLOCK, bench.CONTEXT, job, bench.SEG_CONTEXT, bench.THRESHOLDS, n_lanes=int(os.environ.get("LANES", "8")), device=0, rank=rank, world=world,
                      min_seed_distance=10, filter_fragments=0.1, remove_debris=64)
import hashlib
print(f"rank {rank}/{world}: volume sha1 {hashlib.sha1(vol.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
if world > 1 and os.environ.get("SERIAL") == "1":
    # the ranks predict one after the other (no two processes on the card at the same time)
    for turn in range(world):
        if turn == rank:
            ready = pipe.predict(vol)
            ready[-1].synchronize()
        dist.barrier()
    segs = pipe.seg.run(ready, False)
else:
    segs = pipe.run(vol)
a = pipe.seg.interior(pipe.seg.affs).cpu().numpy(); f = pipe.seg.interior(pipe.seg.frags).cpu().numpy(); s = segs.cpu().numpy()
np.savez(os.path.join(out, f"w{world}_r{rank}.npz"), affs=a, frags=f, segs=s)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
if world > 1 and rank == 0 and os.path.exists(os.path.join(out, "w1_r0.npz")):
    one = np.load(os.path.join(out, "w1_r0.npz"))
    parts = [np.load(os.path.join(out, f"w{world}_r{r}.npz")) for r in range(world)]
    for key, ax in (("affs", 1), ("frags", 0), ("segs", 1)):
        two = np.concatenate([p[key] for p in parts], axis=ax)
        d = two != one[key]
        print(key, "equal" if not d.any() else f"DIFFER {int(d.sum())}", flush=True)
        if d.any():
            if key == "affs":
                dd = np.abs(two.astype(np.int16) - one[key].astype(np.int16))
                print("   largest |difference| of the u8 affinities", int(dd.max()), "mean over the differing", float(dd[d].mean()), "histogram 1/2/3+", int((dd == 1).sum()), int((dd == 2).sum()), int((dd >= 3).sum()))
            idx = np.argwhere(d)
            print("   first", idx[0], "last", idx[-1], "z range", idx[:, -3].min(), idx[:, -3].max())
