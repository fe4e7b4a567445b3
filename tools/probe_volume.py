"""Dev tool: timeline of the volume pipeline (when each block was predicted, when its fragments and its edge scores
were done) to see how far the segmentation lanes get while the predict stream runs.
usage: probe_volume.py [steps] [precision] [lanes]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd import volume as V

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision=prec).load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
# timing events on everything
orig_lf, orig_ls = V.SlabSegmenter._launch_fragments, V.SlabSegmenter._launch_scores
marks = {}
def lf(self, k, wait=()):
    orig_lf(self, k, wait)
    lane = self.lane_of("f", k)
    ev = torch.cuda.Event(enable_timing=True); ev.record(lane["stream"]); marks[("frag", k)] = ev
def ls(self, k, wait=()):
    orig_ls(self, k, wait)
    lane = self.lane_of("s", k)
    ev = torch.cuda.Event(enable_timing=True); ev.record(lane["stream"]); marks[("score", k)] = ev
V.SlabSegmenter._launch_fragments, V.SlabSegmenter._launch_scores = lf, ls
stage_t = {}
def timed(name, fn):
    def w(self, *a, **k):
        t = time.perf_counter()
        r = fn(self, *a, **k)
        stage_t[name] = (t, time.perf_counter())
        return r
    return w
V.SlabSegmenter._collect = timed("collect", V.SlabSegmenter._collect)
V.SlabSegmenter.stitch = timed("stitch", V.SlabSegmenter.stitch)
for rep in range(2):
    pipe = V.VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(steps), SEG_CONTEXT, THRESHOLDS, n_lanes=lanes)
    marks.clear()
    orig_predict = pipe.predict
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ready = pipe.predict(vol)
    t_enq = time.perf_counter() - t0
    pev = []
    # re-record timing events is not possible after the fact: time the ready events through elapsed_time needs timing events;
    # instead poll: host timestamps when each event completes
    segs = None
    import threading
    times = {}
    def poll():
        pend = {("pred", k): e for k, e in enumerate(ready)}
        while pend or not done_flag[0]:
            for key, e in list(pend.items()):
                if e.query():
                    times[key] = time.perf_counter() - t0
                    del pend[key]
            for key, e in list(marks.items()):
                if key not in times and e.query():
                    times[key] = time.perf_counter() - t0
            if done_flag[0] and not pend and all(k in times for k in marks):
                break
            time.sleep(0.0005)
    done_flag = [False]
    th = threading.Thread(target=poll); th.start()
    segs = pipe.seg.run(ready)
    t_all = time.perf_counter() - t0
    done_flag[0] = True
    th.join()
    print(f"rep {rep}: enqueue of predict {t_enq*1e3:.1f} ms, total {t_all*1e3:.1f} ms ({t_all/steps*1e3:.1f} ms per block)")
    if rep == 1:
        for name, (a, b) in stage_t.items():
            print(f"{name}: {1e3 * (a - t0):8.1f} -> {1e3 * (b - t0):8.1f} ms")
        for k in range(steps):
            print(f"block {k:3d}: predicted {times[('pred',k)]*1e3:8.1f}  fragments {times.get(('frag',k),0)*1e3:8.1f}  scores {times.get(('score',k),0)*1e3:8.1f}")
    del pipe
