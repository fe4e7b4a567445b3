"""Dev tool: where a tile of the halo-resident conv kernel (conv_h16.hip) spends its time, per layer (in-kernel stamps).
Needs a -DBSMI_STAMP build of the library passed as BSMI_LIB (see probe_stamps.py)."""
import ctypes as C
import sys
sys.path.insert(0, ".")
import torch
from bootstrapper_amd import _lib
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC

lib = C.CDLL(_lib.LIB_PATH)
m = Model(NC, precision="bf16x3").load_state_dict(synthetic_state_dict(NC, 0))
raw = synthetic_volume((156, 220, 220), 0)
for _ in range(2):
    m.predict_u8(raw)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.bsmi_debug_stamps_h16(out, 1)
n = 3
for _ in range(n):
    m.predict_u8(raw)
torch.cuda.synchronize()
lib.bsmi_debug_stamps_h16(out, 0)
halo, kc, kw, tiles, phases, ksteps, pro, epi = [int(out[i]) for i in range(8)]
us = lambda t, d: t / 100.0 / max(d, 1)
print(f"all halo-resident launches of a forward: {tiles / n:.0f} tiles, {phases / max(tiles,1):.1f} phases and {ksteps / max(tiles,1):.1f} K-steps per tile")
print(f"per phase: halo issue + wait + barrier {us(halo, phases):.2f} us;  per K-step: issue + MFMAs {us(kc, ksteps):.3f} us, wait + barrier {us(kw, ksteps):.3f} us")
print(f"per tile: prologue {us(pro, tiles):.2f} us, halo {us(halo, tiles):.1f} us, K-steps {us(kc, tiles):.1f} + {us(kw, tiles):.1f} us, epilogue {us(epi, tiles):.2f} us")
