"""bootstrapper_amd: MI355X-native engine for the blockwise affinity-prediction and
watershed-segmentation hot path of ucsdmanorlab/bootstrapper.

The compute lives in libbsmi.so (hand-written HIP for gfx950, C ABI in include/bsmi.h);
this package is the thin host-side mirror of the reference's Python interfaces.
"""
__version__ = "0.1.0"

import os as _os

# One hardware queue per HIP stream of the block pipeline (predict lanes + segmentation lanes).
# With the runtime default (4) streams share queues and a long sequential segmentation kernel
# stalls whatever is queued behind it.  Must be in the environment before the HIP runtime
# starts, i.e. before the first CUDA/HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
