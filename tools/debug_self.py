"""Dev tool: predictions of one block while OTHER processes run the same engine on the card (compare tools/debug_foreign.py)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
model = Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
def run():
    u8 = model.predict_u8(A)[0].clone(); torch.cuda.synchronize(); return u8
ref = run()
assert torch.equal(ref, run())
if os.environ.get("CHILD"):
    t = time.time(); bad = 0; n = 0
    while time.time() - t < float(os.environ["CHILD"]):
        n += 1; bad += not torch.equal(ref, run())
    print(f"  child: {bad} of {n} differ", flush=True)
    sys.exit(0)
kids = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, CHILD="30")) for _ in range(int(os.environ.get("KIDS", "3")))]
time.sleep(25)   # the children load their models
bad = 0; worst = 0; n = 0
t = time.time()
while time.time() - t < 20:
    u = run(); n += 1
    d = (u.int() - ref.int()).abs()
    if int(d.max()):
        bad += 1; worst = max(worst, int(d.max()))
for k in kids: k.wait()
print(f"{prec}: {bad} of {n} predictions beside {len(kids)} processes running the same engine differ (largest u8 difference {worst})", flush=True)
