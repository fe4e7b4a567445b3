"""Dev tool: is a block's prediction a function of the block alone?  A, B, A again (and after a block of zeros / of 255s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
model = Model(bench.NET_CONFIG, device=0, precision="bf16x3").load_state_dict(synthetic_state_dict(bench.NET_CONFIG, 0))
vol = synthetic_volume((512, 512, 512), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
B = extract_block_reflect(vol, [300, 250, 200], (156, 220, 220))
def run(x):
    u8, f = model.predict_u8(x, want_f32=True)
    torch.cuda.synchronize()
    return u8[0].clone(), f[0].clone()
a1 = run(A); b1 = run(B); a2 = run(A)
z = run(torch.zeros_like(A)); a3 = run(A)
o = run(torch.full_like(A, 255)); a4 = run(A)
for name, x in (("A after B", a2), ("A after zeros", a3), ("A after 255s", a4)):
    du = int((x[0] != a1[0]).sum()); df = float((x[1] - a1[1]).abs().max())
    print(name, "u8 differing", du, "f32 max diff", df, flush=True)
