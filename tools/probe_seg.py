"""Timing of the segmentation stages on one full-size block (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.post.engine import SegEngine
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
shape = (156, 220, 220)
m = Model(NC, precision="bf16").load_state_dict(synthetic_state_dict(NC, 0))
raw = synthetic_volume(shape, 0)
u8 = m.predict_u8(raw)[0]
affs = u8[:3].contiguous()
print("affs mean", affs.float().mean().item(), "mask frac", ((affs[1].int() + affs[2].int()) >= 256).float().mean().item())
eng = SegEngine((128, 128, 128))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    frags, mx = eng.ws_fragments(affs, True, 10)
    torch.cuda.synchronize(); t1 = time.time()
    segs = eng.agglomerate_mean(affs, frags, [0.2, 0.35, 0.5])
    eng.status(); t2 = time.time()
    print(f"iter {it}: fragments {1e3*(t1-t0):.1f} ms, agglomerate {1e3*(t2-t1):.1f} ms, max_id {int(mx)}, "
          f"nfrag {len(torch.unique(frags))-1}, nseg {[len(torch.unique(s))-1 for s in segs]}")
# synthetic affinity volume (blobby) for comparison
a2 = torch.stack([synthetic_volume((128, 128, 128), 10 + c, corr=(2, 12, 12)) for c in range(3)])
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    frags, mx = eng.ws_fragments(a2, True, 10)
    torch.cuda.synchronize(); t1 = time.time()
    segs = eng.agglomerate_mean(a2, frags, [0.2, 0.35, 0.5])
    eng.status(); t2 = time.time()
    print(f"synthetic affs iter {it}: fragments {1e3*(t1-t0):.1f} ms, agglomerate {1e3*(t2-t1):.1f} ms, max_id {int(mx)}, "
          f"nfrag {len(torch.unique(frags))-1}, nseg {[len(torch.unique(s))-1 for s in segs]}")
