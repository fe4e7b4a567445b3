"""Dev tool: where the first segmentation pass of a pipeline spends more than the second (stage timers)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for, FILTER_FRAGMENTS, REMOVE_DEBRIS
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.volume import VolumePipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((1024,) * 3, seed=0, device=dev)
kw = dict(min_seed_distance=10, filter_fragments=FILTER_FRAGMENTS, remove_debris=REMOVE_DEBRIS)
pipe = VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm = VolumePipeline(m, OUT_BLOCK, CONTEXT, (5, 1, 1), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm.run(vol); del warm
torch.cuda.synchronize()
t0 = time.perf_counter(); ready = pipe.predict(vol); ready[-1].synchronize(); t1 = time.perf_counter()
print(f"predict {1e3 * (t1 - t0):.1f} ms")
for rep in range(4):
    torch.cuda.synchronize(); a = time.perf_counter()
    pipe.seg.run_blocks(None, False); torch.cuda.synchronize(); b = time.perf_counter()
    pipe.seg.stitch(); torch.cuda.synchronize(); c = time.perf_counter()
    print(f"rep {rep}: blocks {1e3 * (b - a):.1f} ms, stitch {1e3 * (c - b):.1f} ms")
print("-- predict, then sleep X ms, then the block stages")
for x in (0, 10, 30, 100, 300, 0, 30):
    ready = pipe.predict(vol); ready[-1].synchronize()
    time.sleep(x * 1e-3)
    a = time.perf_counter()
    pipe.seg.run_blocks(None, False); torch.cuda.synchronize(); b = time.perf_counter()
    print(f"sleep {x:4d} ms: blocks {1e3 * (b - a):.1f} ms")
print("-- memory traffic on a side stream while the last blocks are predicted")
side = torch.cuda.Stream(dev)
b1 = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b2 = torch.empty_like(b1)
for mode in ("none", "copy", "none", "copy"):
    ready = pipe.predict(vol)
    if mode == "copy":
        ready[-4].synchronize()
        with torch.cuda.stream(side):
            for _ in range(40):
                b2.copy_(b1)          # 256 MB each way per copy
    ready[-1].synchronize()
    a = time.perf_counter()
    pipe.seg.run_blocks(None, False); torch.cuda.synchronize(); b = time.perf_counter()
    side.synchronize()
    print(f"{mode}: blocks {1e3 * (b - a):.1f} ms")
