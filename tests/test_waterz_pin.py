"""The pin for what waterz / funlib.segment decide (queue tie order, `discretize_queue` binning, which parallel edge survives,
`<` or `<=` in the global connected components).  Those packages are absent from /root/reference and from this image, so the
oracle's choices are *specified*  and this file REPORTS the parity as unpinned -- a skip with that reason --
until someone runs tools/gen_goldens_waterz.py where waterz is installed and commits tests/golden/waterz_cases.npz; then the same
tests hold oracle/seg_ref.c (and through it the kernels) to real waterz.  CPU only."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "waterz_cases.npz")
UNPINNED = ("parity UNPINNED: tests/golden/waterz_cases.npz is absent (waterz / funlib.segment are not installed here); "
            "run tools/gen_goldens_waterz.py where they are and commit the file")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_goldens_waterz", os.path.join(ROOT, "tools", "gen_goldens_waterz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _same_partition(a, b):
    a, b = a.ravel(), b.ravel()
    if not np.array_equal(a == 0, b == 0):
        return False
    pairs = np.unique(np.stack([a, b]), axis=1)
    return len(np.unique(pairs[0])) == pairs.shape[1] == len(np.unique(pairs[1]))


def test_generator_says_what_is_missing_and_its_inputs_are_tie_rich():
    """Without waterz the generator ends with a clear message and writes nothing; its numpy-made inputs are deterministic and do
    what they are for: the oracle finds many edges with EQUAL scores in them."""
    from oracle import seg_ref as S
    gen = _gen()
    try:
        import waterz  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        before = os.path.exists(GOLD)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_goldens_waterz.py")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "waterz is not installed" in r.stdout and os.path.exists(GOLD) == before
    tied = 0
    for seed, shape, n_seeds, levels in gen.CASES:
        affs, frags = gen.make_case(seed, shape, n_seeds, levels)
        affs2, frags2 = gen.make_case(seed, shape, n_seeds, levels)
        assert np.array_equal(affs, affs2) and np.array_equal(frags, frags2)
        edges, scores, merges, mscores = S.rag_merge_scores_u8(affs, frags, 1.0, 256)
        assert len(edges) > 50 and len(merges) > 10
        segs = S.agglomerate_mean_u8(affs, frags, gen.THRESHOLDS)
        assert len(np.unique(segs[0])) >= len(np.unique(segs[-1]))
        if levels <= 8:
            _, counts = np.unique(mscores, return_counts=True)
            tied += int((counts > 1).sum())
    assert tied > 20   # merges at equal scores: the cases decide a tie rule


def test_oracle_against_waterz_goldens():
    if not os.path.exists(GOLD):
        pytest.skip(UNPINNED)
    from oracle import seg_ref as S
    g = np.load(GOLD)
    thresholds = [float(t) for t in g["thresholds"]]
    names = sorted({k.split("/")[0] for k in g.files if k.startswith("case")})
    assert names
    for name in names:
        affs, frags = g[name + "/affs"], g[name + "/frags"]
        segs = S.agglomerate_mean_u8(affs, frags, thresholds)
        for t in range(len(thresholds)):
            assert _same_partition(segs[t], g[name + "/exact_mean/segs"][t]), (name, "exact queue, mean", thresholds[t])
        hist = S.agglomerate_hist_u8(affs, frags, thresholds, 50, False)
        for t in range(len(thresholds)):
            assert _same_partition(hist[t], g[name + "/exact_hist50/segs"][t]), (name, "exact queue, 50 % quantile", thresholds[t])
        # the blockwise call: initial region graph and the merge sequence of the 256-bin queue
        edges, scores, merges, mscores = S.rag_merge_scores_u8(affs, frags, 1.0, 256)
        rag0 = g[name + "/blockwise/initial_rag"]
        want = {(int(min(u, v)), int(max(u, v))) for u, v, _ in rag0}
        assert {(int(u), int(v)) for u, v in edges} == want, name
        mh = g[name + "/blockwise/merge_history"]
        assert len(mh) == len(merges), name
        assert np.array_equal(np.sort(mh[:, :2].astype(np.uint64), axis=1), np.sort(merges, axis=1)), (name, "merge order of the 256-bin queue")
        assert np.allclose(mh[:, 3].astype(np.float32), mscores, rtol=0, atol=1e-6), name


def test_connected_components_inclusivity_against_funlib_goldens():
    if not os.path.exists(GOLD):
        pytest.skip(UNPINNED)
    g = np.load(GOLD)
    if "cc/nodes" not in g.files:
        pytest.skip("parity UNPINNED for the global connected components: the golden file was made without funlib.segment")
    from oracle import seg_ref as S
    for thr in (0.35, 0.5):
        got = S.connected_components(g["cc/nodes"], g["cc/edges"], g["cc/scores"], thr)
        assert _same_partition(got, g[f"cc/components_{thr}"]), f"threshold {thr}: the specified rule (score <= threshold joins) is not funlib's; set cc_inclusive = false"
