"""Dev tool: victim / aggressor of the concurrent-forward defect: engine 0 (precision P0) is checked while engine 1 (precision P1) keeps predicting."""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
p0, p1 = os.environ.get("P0", "bf16x3"), os.environ.get("P1", "f32")
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
m0 = Model(bench.NET_CONFIG, device=0, precision=p0).load_state_dict(sd)
m1 = Model(bench.NET_CONFIG, device=0, precision=p1).load_state_dict(sd)
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
torch.cuda.synchronize()
r_u8, r_f = m0.predict_u8(A, want_f32=True); torch.cuda.synchronize()
r_u8, r_f = r_u8[0].clone(), r_f[0].clone()
m1.predict_u8(A); torch.cuda.synchronize()
stop = False
def burn():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        while not stop:
            m1.predict_u8(A); torch.cuda.current_stream().synchronize()
t = threading.Thread(target=burn); t.start(); time.sleep(1.0)
s0 = torch.cuda.Stream(); bad = n = 0; worst = 0.0
with torch.cuda.stream(s0):
    t0 = time.time()
    while time.time() - t0 < 12:
        u, f = m0.predict_u8(A, want_f32=True); s0.synchronize()
        d = float((f[0] - r_f).abs().max()); n += 1
        if d > 0: bad += 1; worst = max(worst, d)
stop = True; t.join()
print(f"victim {p0} beside aggressor {p1}: {bad} of {n} predictions differ (largest f32 difference {worst:.3g})", flush=True)
