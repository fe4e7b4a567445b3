"""Worker of tests/test_guards_gpu.py: with BSMI_GUARD_MB set, every device allocation of the network engine and of the training
step lies between zones of 0xFF bytes; a forward pass in each precision, a training step in each mode -> zones written to (0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bootstrapper_amd import _lib
from bootstrapper_amd.unet import Model
from bootstrapper_amd.training import Trainer
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC

assert os.environ.get("BSMI_GUARD_MB")
sd = synthetic_state_dict(NC, 0)
raw = synthetic_volume((64, 148, 148), 0)
ref = None
for prec in ("f32", "bf16x3", "bf16"):
    m = Model(NC, precision=prec).load_state_dict(sd)
    u = m.predict_u8(raw)[0]
    torch.cuda.synchronize()
    assert torch.isfinite(u.float()).all()
    if prec == "f32":
        ref = u.clone()
    else:   # a read past a buffer that reached a result would have poisoned it (0xFF = NaN in every format in use)
        assert (u.int() - ref.int()).abs().max().item() <= (1 if prec == "bf16x3" else 12), prec
    del m
shape = (32, 196, 196)
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.random(shape, dtype=np.float32) * 2 - 1).astype(np.float32)).cuda()
for det in (False, True):
    m = Model(NC, precision="f32").load_state_dict(sd)
    tr = Trainer(m, shape, lr=1e-3, deterministic=det)
    out = (6,) + tuple(tr.out_shape)
    gt = torch.from_numpy((rng.random(out) > 0.5).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.random(out).astype(np.float32)).cuda()
    for _ in range(2):
        loss = tr.forward_backward(x, [gt], [w])
        assert np.isfinite(loss)
        tr.optimizer_step()
    torch.cuda.synchronize()
    assert torch.isfinite(tr.grads).all() and torch.isfinite(tr.params).all()
    bad = _lib.lib.bsmi_debug_check_guards()
    assert bad == 0, f"{bad} guard zones written to (deterministic={det})"
    tr.close()
    del m
# the segmentation engines' workspaces (seeds, floods, region graphs, merge loops, sorts): a box of 1 x 2 x 2 blocks, predict + segment
from bootstrapper_amd.volume import VolumePipeline
m = Model(NC).load_state_dict(sd)
vol = synthetic_volume((128, 256, 256), 0)
pipe = VolumePipeline([m, m.clone()], (128, 128, 128), (14, 46, 46), (1, 2, 2), (16, 16, 16), [0.2, 0.35, 0.5], min_seed_distance=10,
                      filter_fragments=0.1, remove_debris=64, n_lanes=4)
segs = pipe.run(vol)
torch.cuda.synchronize()
assert int(segs.max()) > 1000
bad = _lib.lib.bsmi_debug_check_guards()
assert bad == 0, f"{bad} guard zones written to (predict + segment)"
print("guards intact", flush=True)
