"""Dev tool: one engine predicting with guarded allocations (BSMI_GUARD_MB, csrc/dev_guard.h).
Run once without the variable (writes the prediction to REF), then with it: is the prediction the same (no read past a
buffer reaches a result) and are the guard zones intact (no write past a buffer)?"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd import _lib
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
ref_path = os.environ.get("REF", "/tmp/bsmi_guard_ref.npy")
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
m = Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(sd)
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
shape = tuple(int(x) for x in os.environ.get("SHAPE", "156,220,220").split(","))
A = extract_block_reflect(vol, [10, 20, 30], shape)
outs = []
for _ in range(3):
    u = m.predict_u8(A)[0]; torch.cuda.synchronize(); outs.append(u.cpu().numpy())
assert all(np.array_equal(outs[0], o) for o in outs)
guard = os.environ.get("BSMI_GUARD_MB")
if not guard:
    np.save(ref_path, outs[0]); print("reference written", outs[0].shape, int(outs[0].astype(np.int64).sum()))
else:
    ref = np.load(ref_path)
    d = np.abs(outs[0].astype(np.int32) - ref.astype(np.int32))
    print(f"{prec} {shape} guard {guard} MiB: {int((d > 0).sum())} of {d.size} u8 values differ from the unguarded run (largest {int(d.max())}); "
          f"guard zones written to: {_lib.lib.bsmi_debug_check_guards()}", flush=True)
