"""Dev tool: which launch goes wrong when two engines of one process predict side by side?  Engine 0's per-launch outputs alone
vs. its per-launch outputs of a forward made while engine 1 keeps predicting."""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
m0 = Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(sd)
m1 = Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(sd)
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
torch.cuda.synchronize()
m0.profile(True)
m0.predict_u8(A); torch.cuda.synchronize()
prof0 = m0.read_profile()
nsteps = len([p for p in prof0 if p[0] != 4])
ref = [m0.debug_activation(s) for s in range(nsteps)]
ref_u8 = m0.predict_u8(A)[0].clone(); torch.cuda.synchronize()
if os.environ.get('PROFILE_OFF', '1') == '1':
    m0.profile(False)
stop = False
def burn():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        while not stop:
            m1.predict_u8(A); torch.cuda.current_stream().synchronize()
t = threading.Thread(target=burn); t.start(); time.sleep(1.0)
s0 = torch.cuda.Stream()
for trial in range(3):
    with torch.cuda.stream(s0):
        u = m0.predict_u8(A)[0]; s0.synchronize()
    print(f'trial {trial}: final u8 differing voxels', int((u != ref_u8).sum()), flush=True)
    prof = prof0
    first = None
    for s in range(nsteps):
        got = m0.debug_activation(s)
        d = np.abs(got - ref[s])
        if d.max() > 0:
            bad = np.argwhere(d > 0)
            print(f"trial {trial}: launch {s} (type {prof[s][0]}, {prof[s][1]:.3f} ms) shape {got.shape}: {len(bad)} values differ, max {d.max():.4g}; z range {bad[:,0].min()}..{bad[:,0].max()}, channels {bad[:,3].min()}..{bad[:,3].max()}, rows of M (first) {bad[0]}", flush=True)
            first = s if first is None else first
            if s > (first or 0) + 1: break
    if first is None: print(f"trial {trial}: every launch equal", flush=True)
stop = True; t.join()
