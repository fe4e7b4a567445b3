"""CPU-only checks of the C-ABI library: it loads, exports every symbol that
include/bsmi.h declares, and its shape / flop arithmetic (no device work) is right."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

AFFS_NET_CONFIG = {
    "in_channels": 1, "num_fmaps": 12, "fmap_inc_factor": 5,
    "downsample_factors": [[1, 2, 2], [1, 2, 2], [1, 2, 2]],
    "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4,
    "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3,
    "input_shape": [32, 196, 196], "output_shape": [4, 104, 104], "shape_increase": [0, 216, 216],
    "outputs": {"3d_affs": {"dtype": "uint8", "dims": 6}},
}


def test_library_exports_every_declared_symbol():
    from bootstrapper_amd import _lib
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert len(headers) >= 2
    hdr = "\n".join(open(h).read() for h in headers)
    declared = set(re.findall(r"\b(bsmi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} declared in bsmi.h but not exported"
    assert set(_lib.SYMBOLS) == declared
    assert _lib.lib.bsmi_version() >= 1


def test_shape_arithmetic_matches_reference_table():
    # (156,220,220) -> 128^3, 18,834.4 GFLOP algorithmic; net_config shapes
    from bootstrapper_amd.unet import Model
    m = Model(AFFS_NET_CONFIG)
    assert m.output_shape((156, 220, 220)) == (128, 128, 128)
    assert m.output_shape((32, 196, 196)) == (4, 104, 104)
    assert m.output_shape((32, 412, 412)) == (4, 320, 320)
    assert m.output_shape((124, 188, 188)) == (96, 96, 96)
    assert abs(m.flops((156, 220, 220)) / 1e9 - 18834.4) < 0.5


def test_indivisible_shape_is_rejected_like_reference():
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd._lib import BsmiError
    m = Model(AFFS_NET_CONFIG)
    with pytest.raises(BsmiError, match="Can not downsample shape"):
        m.output_shape((156, 222, 220))
    with pytest.raises(BsmiError):
        m.output_shape((10, 50, 50))


def test_unknown_weight_key_and_shape_mismatch():
    import numpy as np
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd._lib import BsmiError
    m = Model(AFFS_NET_CONFIG)
    with pytest.raises(BsmiError, match="Unexpected key"):
        m.load_state_dict({"unet.nope.weight": np.zeros((1,), np.float32)})
    with pytest.raises(BsmiError, match="size mismatch"):
        m.load_state_dict({"unet.l_conv.0.conv_pass.0.bias": np.zeros((13,), np.float32)})
