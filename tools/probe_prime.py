"""Dev tool: does SlabSegmenter.prime() before the job change the predict stage?  usage: probe_prime.py steps warmup prime(0/1)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for, FILTER_FRAGMENTS, REMOVE_DEBRIS
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd import volume as V
steps, warmup, prime = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((1024,) * 3, seed=0, device=dev)
kw = dict(min_seed_distance=10, filter_fragments=FILTER_FRAGMENTS, remove_debris=REMOVE_DEBRIS)
pipe = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, (warmup, 1, 1), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm.run(vol); del warm
if prime == 1:
    pipe.seg.prime()
elif prime == 2:
    pipe.seg._collect()
elif prime == 3:
    pipe.seg.stitch()
elif prime == 4:   # stitch without the relabel
    import numpy as np
    from bootstrapper_amd.volume import gather_and_stitch
    gather_and_stitch(np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64), np.zeros(0, np.float32), THRESHOLDS)
elif prime == 5:   # only the interior copy
    pipe.seg._fr.copy_(pipe.seg.interior(pipe.seg.frags))
torch.cuda.synchronize()
print("mem allocated / reserved GB", torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
for rep in range(2):
    t0 = time.perf_counter(); ready = pipe.predict(vol); ready[-1].synchronize(); t1 = time.perf_counter()
    print(f"steps {steps} warmup {warmup} prime {prime}: predict {1e3 * (t1 - t0) / steps:.2f} ms per block")
m.profile(1); m.profile_totals(reset=True)
ready = pipe.predict(vol); ready[-1].synchronize()
tot = m.profile_totals(reset=True)
print({k: (round(v[0] / steps, 3), v[2] // steps) for k, v in tot.items()})
raw = synthetic_volume((156, 220, 220), 0)
m.profile(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m.predict_u8(raw)
torch.cuda.synchronize(); print(f"bare predict_u8 on the default stream: {1e2 * (time.perf_counter() - t0):.2f} ms per block")
