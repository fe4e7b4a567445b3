"""Dev tool: does ANOTHER process's work on the card change this process's predictions?  A block is predicted alone, then again and
again while a child process keeps the card busy with plain torch matmuls (no libbsmi)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
model = Model(bench.NET_CONFIG, device=0, precision=os.environ.get("PREC", "bf16x3")).load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
def run():
    u8 = model.predict_u8(A)[0].clone(); torch.cuda.synchronize(); return u8
ref = run()
assert torch.equal(ref, run())
burner = "import torch,time\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nt=time.time()\nwhile time.time()-t<%d:\n    y=x@x\n    torch.cuda.synchronize()\n" % int(os.environ.get("BURN", "25"))
kids = [subprocess.Popen([sys.executable, "-c", burner]) for _ in range(int(os.environ.get("KIDS", "2")))]
time.sleep(8)
bad = 0; worst = 0
for i in range(40):
    u = run()
    d = (u.int() - ref.int()).abs()
    if int(d.max()):
        bad += 1; worst = max(worst, int(d.max()))
for k in kids: k.wait()
print(f"{os.environ.get('PREC', 'bf16x3')}: {bad} of 40 predictions beside {len(kids)} foreign processes differ from the one made alone (largest u8 difference {worst})", flush=True)
assert torch.equal(ref, run())
print("alone again: equal", flush=True)
