"""Host-side logic against golden vectors produced by running the reference
(tools/gen_goldens_host.py).  CPU only."""
import json
import math
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "host_cases.json")) as f:
        return json.load(f)


def test_build_name_and_fmt(cases):
    from bootstrapper_amd.post.naming import build_name, fmt
    for c in cases["naming"]:
        assert build_name(c["params"]) == c["name"]
    for c in cases["fmt"]:
        assert fmt(c["value"]) == c["fmt"]


def test_get_seg_config(cases, tmp_path):
    from bootstrapper_amd.segment import get_seg_config
    for i, c in enumerate(cases["seg_config"]):
        p = tmp_path / f"c{i}.toml"
        p.write_text(c["toml"])
        kw = dict(c["kwargs"])
        if "param" in kw:
            kw["param"] = tuple(kw["param"])
        if c.get("error"):
            with pytest.raises(ValueError) as e:
                get_seg_config(str(p), c["method"], **kw)
            assert str(e.value) == c["message"]
        else:
            assert get_seg_config(str(p), c["method"], **kw) == c["config"]


def test_merge_tree(cases):
    from bootstrapper_amd.post.merge_tree import MergeTree
    for c in cases["merge_tree"]:
        mt = MergeTree(c["leaves"])
        for a, b, t, s in c["merges"]:
            mt.merge(a, b, t, s)
        got = mt.find_merges(c["us"], c["vs"])
        for g, ref in zip(got, c["scores"]):
            assert (ref is None and math.isnan(g)) or (ref is not None and g == ref)
    assert MergeTree([1, 2]).find_merge(1, 2) is None
