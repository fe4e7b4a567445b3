"""Dev tool: where a tile of the batched Winograd GEMM launches spends its time (in-kernel stamps of conv_x3_body).
Needs a -DBSMI_STAMP build of the library passed as BSMI_LIB:
  make -C bootstrapper_amd/csrc CXXFLAGS_EXTRA=-DBSMI_STAMP OUT=../libbsmi_stamp.so BUILD=build_stamp
  BSMI_LIB=$PWD/bootstrapper_amd/libbsmi_stamp.so python tools/probe_stamps.py"""
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
lib.bsmi_debug_stamps(out, 1)
n = 3
for _ in range(n):
    m.predict_u8(raw)
torch.cuda.synchronize()
lib.bsmi_debug_stamps(out, 0)
loop, drain, epi, tiles, pro, sdrain, ksteps = [int(out[i]) for i in range(7)]
us = lambda t: t / 100.0 / max(tiles, 1)   # 100 MHz ticks -> microseconds per tile
print(f"batched-GEMM tiles per forward: {tiles / n:.0f}, K-steps per tile {ksteps / max(tiles, 1):.1f}")
print(f"per tile: prologue {us(pro):.2f} us, K loop {us(loop):.2f} us ({loop / 100.0 / max(ksteps, 1):.3f} us per K-step), drain + barrier {us(drain):.2f} us, "
      f"epilogue (strips + store issue) {us(epi):.2f} us, store drain {us(sdrain):.2f} us")
tot = pro + loop + drain + epi + sdrain
print(f"share of the tile outside the K loop: {1 - loop / tot:.3f}")
