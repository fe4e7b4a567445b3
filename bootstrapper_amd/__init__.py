"""bootstrapper_amd: MI355X-native engine for the blockwise affinity-prediction and
watershed-segmentation hot path of ucsdmanorlab/bootstrapper.

The compute lives in libbsmi.so (hand-written HIP for gfx950, C ABI in include/bsmi.h);
this package is the thin host-side mirror of the reference's Python interfaces.
"""
__version__ = "0.1.0"
