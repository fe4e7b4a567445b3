#!/opt/conda/bin/python3.9
"""Golden vectors for the host-side logic of the hot path, produced by running the REFERENCE:
  post/naming.py   build_name / fmt            (zarr is only used by dump_params: empty placeholder module)
  segment.py       get_seg_config (DEFAULTS < TOML < -p overrides, coordinate parsing)
  post/merge_tree.py MergeTree.merge / find_merges (numba.njit replaced by the identity decorator:
                   JIT compilation does not change semantics)
Run in the build container only:  /opt/conda/bin/python3.9 tools/gen_goldens_host.py
Writes tests/golden/host_cases.json (inputs + the reference's outputs; no reference source).
"""
import json
import math
import os
import sys
import tempfile
import types
import warnings

warnings.filterwarnings("ignore")
REF = "/root/reference/bootstrapper"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "host_cases.json")

import numpy as np  # noqa: E402

sys.modules.setdefault("zarr", types.ModuleType("zarr"))
nb = types.ModuleType("numba")
nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
sys.modules["numba"] = nb
sys.path.insert(0, os.path.join(REF, "post"))
import naming  # noqa: E402
import merge_tree  # noqa: E402

sys.path.insert(0, REF)
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_segment", os.path.join(REF, "segment.py"))
ref_segment = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_segment)

out = {"naming": [], "fmt": [], "seg_config": [], "merge_tree": []}

name_cases = [
    {"merge_function": "mean", "threshold": 0.35, "fragments_in_xy": True, "min_seed_distance": 10,
     "epsilon_agglomerate": 0.0, "filter_fragments": 0.1, "remove_debris": 64, "seed_eps": None, "sigma": None,
     "noise_eps": None, "bias": None},
    {"fragments_in_xy": True, "min_seed_distance": 10, "sigma": None, "noise_eps": None, "bias": None},
    {"merge_function": "hist_quant_75", "threshold": 0.5, "fragments_in_xy": False, "min_seed_distance": 5,
     "sigma": [0, 2, 2], "noise_eps": 0.001, "bias": [-0.4, -0.4, -0.7]},
    {"fragments_in_xy": False, "bias": [0.5, 0.5, 0.5], "sigma": [1, 1, 1], "remove_debris": 0},
    {"threshold": 1e-05, "strides": [[1, 1, 1], [2, 9, 9]], "randomized_strides": True, "global_bias": [1.0, -0.5]},
    {"threshold": 0.2, "unknown_key": 3, "seed_eps": 0.25, "epsilon_agglomerate": 0.05},
    {},
]
for p in name_cases:
    out["naming"].append({"params": p, "name": naming.build_name(p)})
for v in [0.35, 1.0, 1e-05, 10, True, [1, 1, 1], [1, 2, 3], [[1, 1], [2, 9]], "mean", [0.5, 0.5], 123456789.0]:
    out["fmt"].append({"value": v, "fmt": naming.fmt(v)})

toml_a = '''
affs_dataset = "/data/v.zarr/predictions/30000/3d_affs"
fragments_dataset = "/data/v.zarr/fragments"
seg_dataset_prefix = "/data/v.zarr/segmentations"
mask_dataset = "/data/v.zarr/mask"
roi_offset = "0 0 0"
roi_shape = [4000, 5000, 5000]
blockwise = false
num_workers = 8

[ws_params]
thresholds = [0.1, 0.3]
min_seed_distance = 8
'''
toml_b = '''
affs_dataset = "a.zarr/affs"
fragments_dataset = "a.zarr/frags"
seg_dataset_prefix = "a.zarr/segmentations/x"
blockwise = true
block_shape = "128,128,128"
context = "16 16 16"

[db]
db_file = "a.zarr/rag.db"

[ws_params]
fragments_in_xy = false
'''
cases = [
    (toml_a, "ws", {}),
    (toml_a, "ws", {"roi_offset": "40 8 8", "roi_shape": "400 800 800", "param": ("thresholds=[0.5]", "merge_function=mean")}),
    (toml_a, "ws", {"blockwise": None, "num_workers": 2, "param": ("bias=[-0.1,-0.2,-0.3]", "sigma=None", "noise_eps=0.001")}),
    (toml_b, "ws", {"block_context": "8 8 8", "block_shape": "roi"}),
    (toml_b, "ws", {}),
]
for text, method, kwargs in cases:
    with tempfile.NamedTemporaryFile("w", suffix=".toml", delete=False) as f:
        f.write(text)
    try:
        cfg = ref_segment.get_seg_config(f.name, method, **kwargs)
        res = {"config": cfg}
    except Exception as e:  # noqa: BLE001
        res = {"error": type(e).__name__, "message": str(e)}
    os.unlink(f.name)
    kw = dict(kwargs)
    if "param" in kw:
        kw["param"] = list(kw["param"])
    out["seg_config"].append({"toml": text, "method": method, "kwargs": kw, **res})
for text, method, kwargs in [(toml_a, "ws", {"param": ("nope=1",)}),
                             (toml_b.replace('[db]\ndb_file = "a.zarr/rag.db"\n', ""), "ws", {}),
                             (toml_b, "cc", {})]:
    with tempfile.NamedTemporaryFile("w", suffix=".toml", delete=False) as f:
        f.write(text)
    try:
        ref_segment.get_seg_config(f.name, method, **kwargs)
        res = {"error": None}
    except Exception as e:  # noqa: BLE001
        res = {"error": type(e).__name__, "message": str(e)}
    os.unlink(f.name)
    kw = dict(kwargs)
    if "param" in kw:
        kw["param"] = list(kw["param"])
    out["seg_config"].append({"toml": text, "method": method, "kwargs": kw, **res})

rng = np.random.default_rng(5)
for n_leaves in (6, 40):
    leaves = sorted(int(v) for v in rng.choice(np.arange(1, 500), size=n_leaves, replace=False))
    mt = merge_tree.MergeTree(leaves)
    alive = list(leaves)
    merges = []
    score = 0.0
    while len(alive) > max(1, n_leaves // 4):
        i, j = sorted(rng.choice(len(alive), size=2, replace=False))
        a, b = alive[i], alive[j]
        score += float(rng.random()) * 0.1
        c = a  # waterz keeps the id of a
        mt.merge(a, b, c, score)
        merges.append([a, b, c, score])
        alive.remove(b)
    us = [int(v) for v in rng.choice(leaves + [9999], size=30)]
    vs = [int(v) for v in rng.choice(leaves + [9999], size=30)]
    res = mt.find_merges(us, vs)
    out["merge_tree"].append({"leaves": leaves, "merges": merges, "us": us, "vs": vs,
                              "scores": [None if math.isnan(x) else float(x) for x in res]})

with open(OUT, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", OUT, {k: len(v) for k, v in out.items()})
