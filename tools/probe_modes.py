"""Max |difference| of the precision modes against the f32 mode on one block of the full 3d_affs net (dev tool).
usage: probe_modes.py [D,H,W] [mode ...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
shape = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (156, 220, 220)
modes = sys.argv[2:] or ["bf16x3", "bf16"]
sd = synthetic_state_dict(NC, 0)
raw = synthetic_volume(shape, 0)
m = Model(NC, precision="f32").load_state_dict(sd)
_, ref = m.predict_u8(raw, want_f32=True)
ref = ref[0].clone()
for mode in modes:
    m.set_precision(mode)
    m.predict_u8(raw)
    torch.cuda.synchronize(); t0 = time.time()
    _, f = m.predict_u8(raw, want_f32=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    _, f2 = m.predict_u8(raw, want_f32=True)
    d = (f[0] - ref).abs()
    print("   repeatable:", bool(torch.equal(f[0], f2[0])))
    print(f"{shape} {mode}: {dt*1e3:.1f} ms, max |diff| vs f32 {float(d.max()):.3e}, mean {float(d.mean()):.3e}", flush=True)
