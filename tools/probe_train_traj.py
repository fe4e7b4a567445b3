"""Dev probe: loss trajectory of the full 3d_affs net on one fixed sample, split-bf16 against f32 arithmetic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NET_CONFIG
from bootstrapper_amd.unet import Model
from bootstrapper_amd.training import Trainer
from bootstrapper_amd.synth import synthetic_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = (32, 196, 196)
dev = torch.device("cuda", 0)
for arithmetic in ("split-bf16", "f32"):
    m = Model(NET_CONFIG, device=0, precision="f32").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
    tr = Trainer(m, shape, arithmetic=arithmetic)
    g = torch.Generator(device=dev).manual_seed(0)
    out = (6,) + tuple(tr.out_shape)
    batch = {"raw": torch.rand(shape, generator=g, device=dev) * 2 - 1, "gt_affs": (torch.rand(out, generator=g, device=dev) > 0.5).float(),
             "affs_weights": torch.rand(out, generator=g, device=dev)}
    t0 = time.perf_counter()
    traj = []
    for i in range(steps):
        loss = tr.training_step(batch)
        if i % max(1, steps // 10) == 0 or i == steps - 1:
            traj.append((i, round(float(loss), 6)))
    torch.cuda.synchronize()
    print(arithmetic, f"{(time.perf_counter() - t0) / steps * 1e3:.1f} ms/step", traj, flush=True)
    tr.close()
    del m
