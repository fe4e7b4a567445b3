"""Per-launch timing of one full-size block (dev tool)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
shape = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (156, 220, 220)
t0 = time.time(); sd = synthetic_state_dict(NC, 0); print("weights", time.time() - t0)
t0 = time.time(); m = Model(NC, precision=prec).load_state_dict(sd); print("pack+upload", time.time() - t0)
raw = synthetic_volume(shape, 0)
m.profile(True)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    u8 = m.predict_u8(raw)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"iter {it}: {dt*1e3:.1f} ms, {m.flops(shape)/dt/1e12:.1f} TFLOP/s")
names = {0: "input", 1: "conv", 2: "pool", 3: "up", 4: "head"}
tot = 0
for t, ms, fl in m.read_profile():
    tot += ms
    print(f"{names[t]:6s} {ms:9.3f} ms  {fl/1e9:10.1f} GFLOP  {fl/ms/1e9 if ms>0 else 0:8.1f} TFLOP/s")
print("sum", tot)
a = u8[0].float()
print("affs u8 mean/std/min/max", a.mean().item(), a.std().item(), a.min().item(), a.max().item())
