"""Dev tool: where the set-up time of `bs predict` goes (checkpoint -> first block)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
t = time.perf_counter()
sd = synthetic_state_dict(NET_CONFIG, 0)
print(f"synthetic weights {time.perf_counter() - t:.2f} s")
path = os.path.join(tempfile.mkdtemp(), "m.ckpt")
torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, path)
raw = synthetic_volume((156, 220, 220), 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
ck = torch.load(path, map_location="cpu", weights_only=True)
t1 = time.perf_counter()
m = Model(NET_CONFIG, precision="bf16x3")
m.load_state_dict({k: v.numpy() for k, v in ck["model_state_dict"].items()})
t2 = time.perf_counter()
m._finalize(m.precision) if hasattr(m, "_finalize") else None
torch.cuda.synchronize()
t3 = time.perf_counter()
m.predict_u8(raw); torch.cuda.synchronize()
t4 = time.perf_counter()
m.predict_u8(raw); torch.cuda.synchronize()
t5 = time.perf_counter()
print(f"torch.load {t1-t0:.2f} s, load_state_dict {t2-t1:.2f} s, finalize (pack + upload) {t3-t2:.2f} s, first block (plan) {t4-t3:.2f} s, second block {t5-t4:.3f} s")
