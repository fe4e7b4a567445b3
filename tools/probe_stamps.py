"""Dev tool: per-tile time split of the conv kernels (needs a -DBSMI_STAMP build passed as BSMI_LIB)."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from bootstrapper_amd import _lib
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
raw_lib = C.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.lib._lib
m = Model(NC, precision="bf16").load_state_dict(synthetic_state_dict(NC, 0))
raw = synthetic_volume((156, 220, 220), 0)
m.predict_u8(raw); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
raw_lib.bsmi_debug_stamps(buf, 1)
m.predict_u8(raw); torch.cuda.synchronize()
raw_lib.bsmi_debug_stamps(buf, 1)
n = max(1, buf[3])
print("tiles", buf[3], "per tile us: loop %.1f drain+barrier %.1f epilogue %.1f store-drain %.1f" % (buf[4] / n / 100, buf[0] / n / 100, buf[1] / n / 100, buf[2] / n / 100))
print("totals ms (sum over tiles / 256 CUs): loop %.2f drain %.2f epi %.2f stdrain %.2f" % tuple(b / 100 / 1e3 / 256 for b in (buf[4], buf[0], buf[1], buf[2])))
