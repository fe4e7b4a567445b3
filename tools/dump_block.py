"""Dev tool: predict one full-size block and save the float outputs (compare kernel variants across processes)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
prec, out = sys.argv[1], sys.argv[2]
m = Model(NC, precision=prec).load_state_dict(synthetic_state_dict(NC, 0))
raw = synthetic_volume((156, 220, 220), 0)
u8, f = m.predict_u8(raw, want_f32=True)
torch.cuda.synchronize()
np.save(out, f[0].cpu().numpy())
