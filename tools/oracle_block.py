"""Dev tool: the CPU oracle on one full-size block (156,220,220) -> float outputs (about 20 s on 16 cores)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import unet_ref as R
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
sd = synthetic_state_dict(NC, 0)
raw = synthetic_volume((156, 220, 220), 0).cpu().numpy()
torch.set_num_threads(16)
t0 = time.time()
out = R.predict_block(R.default_cfg(12, 5), sd, raw, ["affs_head"])[0]
print("oracle seconds", time.time() - t0, out.shape)
np.save(sys.argv[1], out)
