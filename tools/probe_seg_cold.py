"""Dev tool: where the first segmentation pass of a pipeline spends more than a repeated one (stage timers)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for, FILTER_FRAGMENTS, REMOVE_DEBRIS
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd import volume as V

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((1024,) * 3, seed=0, device=dev)
kw = dict(min_seed_distance=10, filter_fragments=FILTER_FRAGMENTS, remove_debris=REMOVE_DEBRIS)
pipe = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, (5, 1, 1), SEG_CONTEXT, THRESHOLDS, n_lanes=16, **kw)
warm.run(vol); del warm
orig_collect = V.SlabSegmenter._collect
tc = {}
def timed_collect(self):
    for lane in self.lanes: lane["stream"].synchronize()
    tc["lanes"] = time.perf_counter()
    r = orig_collect(self); torch.cuda.synchronize(); tc["collect"] = time.perf_counter(); return r
V.SlabSegmenter._collect = timed_collect
def one(label):
    torch.cuda.synchronize(); a = time.perf_counter()
    pipe.seg.run_blocks(None, False); torch.cuda.synchronize(); b = time.perf_counter()
    pipe.seg.stitch(); torch.cuda.synchronize(); c = time.perf_counter()
    print(f"{label:28s} lanes {1e3 * (tc['lanes'] - a):7.1f} ms, collect {1e3 * (tc['collect'] - tc['lanes']):6.1f} ms, stitch {1e3 * (c - b):6.1f} ms")
ready = pipe.predict(vol); ready[-1].synchronize()
one("after predict")
one("repeat 1"); one("repeat 2"); one("repeat 3")
ready = pipe.predict(vol); ready[-1].synchronize()
one("after predict again")
one("repeat")
