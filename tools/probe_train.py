"""Dev tool: time the training step of the full 3d_affs net on the reference's training block."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.training import Trainer
from bootstrapper_amd.synth import synthetic_state_dict
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
shape = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (32, 196, 196)
m = Model(NC, precision="f32").load_state_dict(synthetic_state_dict(NC, 0))
tr = Trainer(m, shape)
out = tr.out_shape
print("in", shape, "out", out, "fwd GFLOP", m.flops(shape) / 1e9)
g = torch.Generator(device="cuda").manual_seed(0)
raw = torch.rand(shape, generator=g, device="cuda") * 2 - 1
gt = (torch.rand((6,) + tuple(out), generator=g, device="cuda") > 0.5).float()
w = torch.rand((6,) + tuple(out), generator=g, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    loss = tr.forward_backward(raw, [gt], [w])
    torch.cuda.synchronize(); t1 = time.time()
    tr.optimizer_step()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"iter {it}: loss {loss:.6f} fwd+bwd {1e3*(t1-t0):.1f} ms, adam+repack {1e3*(t2-t1):.1f} ms")
