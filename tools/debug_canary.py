"""Dev tool: LDS canaries beside an engine (see bsmi_debug_lds_canary): does any of its kernels write outside its LDS allocation?"""
import os, sys, threading, time, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd import _lib
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
m = Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
m.predict_u8(A); torch.cuda.synchronize()
stop = False
def burn():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        while not stop:
            m.predict_u8(A); torch.cuda.current_stream().synchronize()
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
lds = int(os.environ.get("LDS", "8192"))
s1 = torch.cuda.Stream()
def canaries(n):
    for _ in range(n):
        _lib.check(_lib.lib.bsmi_debug_lds_canary(lds, 1024, 40, C.c_void_p(bad.data_ptr()), C.c_void_p(s1.cuda_stream)))
    s1.synchronize()
canaries(20)
print(f"{prec}: canaries alone: {int(bad.item())} changed words", flush=True)
t = threading.Thread(target=burn); t.start(); time.sleep(0.5)
canaries(400)
stop = True; t.join()
print(f"{prec}: canaries of {lds} B beside the engine: {int(bad.item())} changed words", flush=True)
