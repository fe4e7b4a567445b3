"""Dev tool: where the end of the job goes after the last block's edges are scored: the host part of `_collect`, the
gather / connected components, the LUT upload and the three relabels of `stitch` (bench geometry).
usage: probe_stitch.py [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd import volume as V
from bootstrapper_amd.post.engine import lut_relabel, lut_relabel_multi

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
pipe = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=16)
T = {}
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[label or name] = T.get(label or name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, w)
for rep in range(3):
    ready = pipe.predict(vol)
    ready[-1].synchronize()
    s = pipe.seg
    t0 = time.perf_counter()
    s.run_blocks(ready, False)          # launches + _collect
    t1 = time.perf_counter()
    # stitch, piece by piece
    nodes = np.concatenate([np.arange(1, int(n) + 1, dtype=np.uint64) + np.uint64(bid * s.nvb) for n, bid in zip(s.block_nums, s.block_ids)])
    t2 = time.perf_counter()
    s.nodes, s.luts = V.gather_and_stitch(nodes, s.rag_edges, s.rag_scores, s.thresholds, 0, 1, None)
    t3 = time.perf_counter()
    fr = s._fr
    fr.copy_(s.interior(s.frags)); torch.cuda.synchronize()
    t4 = time.perf_counter()
    keys = torch.from_numpy(s.nodes.view(np.int64)).to(dev)
    vals = [torch.from_numpy(c.view(np.int64)).to(dev) for c in s.luts]
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    for t, v in enumerate(vals):
        lut_relabel(fr, keys, v, out=s.segs[t])
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    lut_relabel_multi(fr, keys, torch.stack(vals), out=s.segs)
    torch.cuda.synchronize()
    t7 = time.perf_counter()
    print(f"        one pass for the three: {1e3*(t7-t6):.2f} ms")
    print(f"rep {rep}: blocks+collect {1e3*(t1-t0):.1f}  nodes {1e3*(t2-t1):.2f}  components {1e3*(t3-t2):.2f}  interior copy {1e3*(t4-t3):.2f}  "
          f"LUT upload {1e3*(t5-t4):.2f}  3 relabels {1e3*(t6-t5):.2f} ms;  {len(nodes)} nodes, {len(s.rag_scores)} edges")
# the host part of _collect alone: run the blocks, wait for the lanes, then time it
ready = pipe.predict(vol); ready[-1].synchronize()
s = pipe.seg
orig = s._sync
def sync_timed():
    r = orig(); T["t_sync_done"] = time.perf_counter(); return r
s._sync = sync_timed
s.run_blocks(ready, False)
print(f"_collect after the lanes are idle: {1e3*(time.perf_counter()-T['t_sync_done']):.2f} ms")
