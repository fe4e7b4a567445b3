"""Dev tool: host-side timeline of the segmentation stage after the last predicted block, without a profiler: when each
launch phase of `run_blocks` returns, when the lanes drain, what `_collect` and `stitch` cost (bench geometry).
usage: probe_tail.py [steps] [lanes]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd import volume as V

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
pipe = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=lanes)
s = pipe.seg
marks = []
def mark(name):
    marks.append((name, time.perf_counter()))
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        r = fn(*a, **k)
        what = label(*a, **k) if callable(label) else label
        if what:
            mark(what)
        return r
    setattr(obj, name, w)
_nf = [0]
def _count(*a, **k):
    _nf[0] += 1
    return f"fragment task {_nf[0]} queued" if _nf[0] in (1, len(s.boxes)) else None
wrap(s, "_launch_fragments", _count)
wrap(s, "_sync", "lanes drained + status")
wrap(s, "_collect", "_collect done")
from bootstrapper_amd.post import engine as E, watershed as W
wrap(E, "rag_merge_scores_host", "host merge loops done")
_orig_cpu = torch.Tensor.cpu
def _cpu(t, *a, **k):
    r = _orig_cpu(t, *a, **k); mark(f"D2H {tuple(t.shape)}"); return r
torch.Tensor.cpu = _cpu
wrap(W, "connected_components_multi", "components done")
# device-side times without a profiler: an event behind every block's flood, fragment task and score task on its lane
evs = {"flood": [], "frag task": [], "score task": []}
def after(obj, name, kind):
    fn = getattr(obj, name)
    def w(*a, **k):
        r = fn(*a, **k)
        e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream(dev)); evs[kind].append(e)
        return r
    setattr(obj, name, w)
for lane in s.lanes:
    after(lane["engine"], "ws_fragments", "flood")
    after(lane["engine"], "label_stats", "frag task")
    after(lane["engine"], "rag_graph_async", "score task")
for rep in range(3):
    ready = pipe.predict(vol)
    ready[-1].synchronize()
    del marks[:]
    _nf[0] = 0
    for v in evs.values():
        del v[:]
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record(torch.cuda.current_stream(dev))
    t0 = time.perf_counter()
    s.run_blocks(ready, False)
    mark("run_blocks returns")
    s.stitch()
    mark("stitch done")
    # when did the fragment tasks finish on the device?  (events recorded at the end of each)
    print(f"rep {rep}: " + "; ".join(f"{n} {1e3 * (t - t0):.1f}" for n, t in marks))
    for kind, es in evs.items():
        ts = sorted(ev0.elapsed_time(e) for e in es)
        if ts:
            print(f"        {kind}s end (device, ms after the start): first {ts[0]:.1f}, median {ts[len(ts) // 2]:.1f}, last {ts[-1]:.1f}")
