"""Dev tool: are kernels that are not ours corrupted beside a bf16x3 forward pass?  torch matmuls / reductions / a segmentation engine's
fragments on another stream are compared with their own results made alone."""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.post.engine import SegEngine
m = Model(bench.NET_CONFIG, device=0, precision="bf16x3").load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
affs = m.predict_u8(A)[0][:3].contiguous(); torch.cuda.synchronize()
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
eng = SegEngine((128, 128, 128), 0)
ref_mm = (x @ x).clone(); ref_sum = x.float().sum(dim=0).clone()
fr, mx = eng.ws_fragments(affs, True, 10); ref_fr = fr.clone(); torch.cuda.synchronize()
stop = False
def burn():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        while not stop:
            m.predict_u8(A); torch.cuda.current_stream().synchronize()
t = threading.Thread(target=burn); t.start(); time.sleep(0.5)
s1 = torch.cuda.Stream(); bad = {"matmul": 0, "reduction": 0, "fragments": 0}; n = 0
with torch.cuda.stream(s1):
    t0 = time.time()
    while time.time() - t0 < 15:
        n += 1
        bad["matmul"] += not torch.equal(x @ x, ref_mm)
        bad["reduction"] += not torch.equal(x.float().sum(dim=0), ref_sum)
        f2, _ = eng.ws_fragments(affs, True, 10); s1.synchronize()
        bad["fragments"] += not torch.equal(f2, ref_fr)
stop = True; t.join()
print(f"beside a bf16x3 engine, {n} rounds: differing results {bad}", flush=True)
