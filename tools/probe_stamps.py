"""Dev tool: per-tile time split of the implicit-GEMM conv kernels and the shader clock they ran at, alone and under the
block pipeline (segmentation lanes beside the predict stream).  Needs a -DBSMI_STAMP build passed as BSMI_LIB:
  make -C bootstrapper_amd/csrc CXXFLAGS_EXTRA=-DBSMI_STAMP OUT=../libbsmi_stamp.so BUILD=build_stamp
  BSMI_LIB=$PWD/bootstrapper_amd/libbsmi_stamp.so python tools/probe_stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONTEXT, NET_CONFIG, OUT_BLOCK, THRESHOLDS  # noqa: E402
from bootstrapper_amd import _lib  # noqa: E402
from bootstrapper_amd.pipeline import BlockPipeline, block_grid  # noqa: E402
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume  # noqa: E402
from bootstrapper_amd.unet import Model  # noqa: E402

raw_lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()


def report(tag):
    raw_lib.bsmi_debug_stamps(buf, 1)
    n = max(1, buf[3])
    loop_us = buf[4] / 100
    print(f"{tag}: tiles {buf[3]}; per tile us: loop {loop_us / n:.1f} drain+barrier {buf[0] / n / 100:.1f} epilogue {buf[1] / n / 100:.1f} "
          f"store-drain {buf[2] / n / 100:.1f}; shader clock in the K loop {buf[5] / max(loop_us, 1e-9) / 1e3:.3f} GHz")


dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
grid = block_grid(vol.shape, OUT_BLOCK)
for segment in (False, True):
    pipe = BlockPipeline(m, OUT_BLOCK, CONTEXT, THRESHOLDS, n_seg_lanes=8, segment=segment)
    pipe.run(vol, grid[:4]); pipe.finish()
    raw_lib.bsmi_debug_stamps(buf, 1)
    pipe.run(vol, grid[4:36]); pipe.finish()
    report("with lanes" if segment else "predict only")
