"""Dev tool: one block's fragment kernels alone on an idle GPU (160^3 read box, smooth synthetic affinities), for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/probe_seeds.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bootstrapper_amd.post.engine import SegEngine
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((1, 3, 168, 168, 168), device=dev, generator=g)
for _ in range(3):
    x = torch.nn.functional.avg_pool3d(x, 3, 1, 1)
x = x[0, :, 4:164, 4:164, 4:164]
x = (x - x.mean()) / x.std() * 60 + 150
affs = x.clamp(0, 255).to(torch.uint8).contiguous()
eng = SegEngine((160, 160, 160), 0)
for _ in range(5):
    fr, n = eng.ws_fragments(affs, True, 10)
torch.cuda.synchronize()
print("fragments", int(n))
