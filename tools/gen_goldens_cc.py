#!/opt/conda/bin/python3.9
"""Golden vectors for the thresholded-affinity connected components (`bs segment --cc`), produced by running the
REFERENCE post/cc.py (numba.jit replaced by the identity decorator: JIT compilation does not change semantics).
Run in the build container only:  /opt/conda/bin/python3.9 tools/gen_goldens_cc.py
Writes tests/golden/cc_cases.npz (inputs + the reference's outputs; no reference source)."""
import os
import sys
import types

import numpy as np

nb = types.ModuleType("numba")
nb.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
nb.boolean = np.bool_
sys.modules["numba"] = nb
sys.path.insert(0, "/root/reference/bootstrapper/post")
import cc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cc_cases.npz")
from scipy.ndimage import gaussian_filter  # noqa: E402

rng = np.random.default_rng(5)
arrs = {}
cases = [("blobs", (6, 20, 24), 2.0, 0.5), ("noise", (4, 12, 10), 0.0, 0.5), ("fine", (5, 16, 16), 1.0, 0.35),
         ("dense", (3, 10, 12), 1.5, 0.2), ("sparse", (4, 14, 9), 1.0, 0.8), ("line", (1, 1, 30), 0.0, 0.5)]
for name, shape, sigma, thr in cases:
    a = rng.random((3,) + shape)
    if sigma:
        a = gaussian_filter(a, sigma=(0, min(sigma, 1), sigma, sigma))
        a = (a - a.min()) / (a.max() - a.min())
    u8 = (a * 255).astype(np.uint8)
    affs = u8.astype(np.float32) / 255.0                 # post/connected_components.py:49-52
    hard = affs > thr                                    # :77
    seg = cc.compute_connected_component_segmentation(hard)
    arrs[name + "/affs"] = u8
    arrs[name + "/thr"] = np.float64(thr)
    arrs[name + "/hard"] = hard
    arrs[name + "/seg"] = seg.astype(np.uint32)
    print(name, shape, thr, "components", int(seg.max()), "labelled", float((seg > 0).mean()))
np.savez_compressed(OUT, **arrs)
