#!/usr/bin/env python3
"""One-command pin for what waterz / funlib.segment decide (missing 3): run this WHERE `waterz` AND
`funlib.segment` ARE INSTALLED (they are absent from the build container, the GPU boxes and /root/reference: unpinned git
dependencies, reference pyproject.toml:52-56) and commit the file it writes, tests/golden/waterz_cases.npz.  From then on
tests/test_waterz_pin.py holds oracle/seg_ref.c -- and through it the HIP kernels and the host loops, which are held bit-exact
to the oracle -- to real waterz on tie-rich graphs; until then that test reports "parity unpinned".

    python tools/gen_goldens_waterz.py            # -> tests/golden/waterz_cases.npz, or a clear "not installed" message

What the cases decide (the *specified* choices):
  * queue tie order of the exact queue (discretize_queue=0; reference post/watershed.py:333-338): affinities quantised to 2-8
    levels, so that many edges carry EQUAL scores and the merge order is the tie rule;
  * `discretize_queue=256` binning (reference post/blockwise/waterz_agglom.py:131-139): bin = (int)(score * (N-1)) or (int)(score * N);
  * which of two parallel edges survives a merge, and how stale edges are re-queued (both visible in merge_history);
  * `funlib.segment.graphs.impl.connected_components` (reference post/watershed.py:182): is an edge whose score EQUALS the threshold merged?
Inputs are made by numpy alone (seeded), so this script needs nothing of this repository; the arrays it stores are inputs and
waterz's outputs -- data, not source.
"""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "waterz_cases.npz")
MEAN = "OneMinus<MeanAffinity<RegionGraphType, ScoreValue>>"
HIST = "OneMinus<HistogramQuantileAffinity<RegionGraphType, 50, ScoreValue, 256, false>>"


def make_case(seed, shape, n_seeds, levels):
    """(affs u8 [3][D][H][W], fragments u64 [D][H][W]): fragments = Voronoi cells of random points (L1 metric, ties to the lower
    id), affinities = smooth noise, saturated and quantised to `levels` values"""
    rng = np.random.default_rng(seed)
    D, H, W = shape
    pts = np.stack([rng.integers(0, s, n_seeds) for s in shape], axis=1)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    dist = np.abs(zz[None] - pts[:, 0, None, None, None]) * 3 + np.abs(yy[None] - pts[:, 1, None, None, None]) + np.abs(xx[None] - pts[:, 2, None, None, None])
    frags = (np.argmin(dist, axis=0) + 1).astype(np.uint64)
    frags[rng.random(shape) < 0.03] = 0                         # some background
    a = rng.random((3,) + shape)
    for ax in (1, 2, 3):                                          # cheap smoothing without scipy
        a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax)) / 3
    a = (a - a.min()) / (a.max() - a.min())
    a = np.clip((a - 0.5) * 3 + 0.5, 0, 1)
    a = np.round(a * (levels - 1)) / (levels - 1)
    return (a * 255).astype(np.uint8), frags


CASES = [(1, (6, 24, 24), 40, 2), (2, (6, 24, 24), 40, 3), (3, (8, 32, 32), 90, 4), (4, (4, 40, 40), 120, 8), (5, (8, 32, 32), 60, 256)]
THRESHOLDS = [0.2, 0.35, 0.5, 0.8]


def main():
    try:
        import waterz
    except ImportError as exc:
        print(f"waterz is not installed here ({exc}): nothing written.  Run this script on a machine that has the reference's "
              "environment (pip install of ucsdmanorlab/bootstrapper pulls waterz and funlib.segment from git) and commit "
              f"{os.path.relpath(OUT)}; until then tests/test_waterz_pin.py reports the agglomeration parity as unpinned.")
        return 2
    try:
        from funlib.segment.graphs.impl import connected_components
    except ImportError as exc:
        connected_components = None
        print(f"funlib.segment is not installed here ({exc}): the connected-components case is left out")
    out = {"thresholds": np.array(THRESHOLDS, np.float32), "waterz_version": np.bytes_(getattr(waterz, "__version__", "unknown"))}
    for seed, shape, n_seeds, levels in CASES:
        affs_u8, frags = make_case(seed, shape, n_seeds, levels)
        affs = (affs_u8 / 255.0).astype(np.float32)
        name = f"case{seed}"
        out[name + "/affs"], out[name + "/frags"] = affs_u8, frags
        for tag, fn, dq in (("exact_mean", MEAN, 0), ("bins256_mean", MEAN, 256), ("exact_hist50", HIST, 0)):
            segs, hist, graphs = [], [], []
            kw = dict(discretize_queue=dq) if dq else {}
            for item in waterz.agglomerate(affs=np.ascontiguousarray(affs), thresholds=list(THRESHOLDS), fragments=frags.copy(),
                                           scoring_function=fn, return_merge_history=True, return_region_graph=True, **kw):
                seg, mh, rg = item
                segs.append(seg.copy())
                hist.append(np.array([(m["a"], m["b"], m["c"], m["score"]) for m in mh], dtype=np.float64).reshape(-1, 4))
                graphs.append(np.array([(e["u"], e["v"], e["score"]) for e in rg], dtype=np.float64).reshape(-1, 3))
            out[f"{name}/{tag}/segs"] = np.stack(segs)
            for t in range(len(THRESHOLDS)):
                out[f"{name}/{tag}/merge_history{t}"] = hist[t]
                out[f"{name}/{tag}/region_graph{t}"] = graphs[t]
        # the blockwise call exactly as waterz_agglom.py:131-139 makes it
        gen = waterz.agglomerate(affs=np.ascontiguousarray(affs), thresholds=[0, 1.0], fragments=frags.copy(), scoring_function=MEAN,
                                 discretize_queue=256, return_merge_history=True, return_region_graph=True)
        _, _, rag0 = next(gen)
        _, mh, _ = next(gen)
        out[name + "/blockwise/initial_rag"] = np.array([(e["u"], e["v"], e["score"]) for e in rag0], dtype=np.float64).reshape(-1, 3)
        out[name + "/blockwise/merge_history"] = np.array([(m["a"], m["b"], m["c"], m["score"]) for m in mh], dtype=np.float64).reshape(-1, 4)
    if connected_components is not None:
        nodes = np.arange(1, 9, dtype=np.uint64)
        edges = np.array([[1, 2], [2, 3], [4, 5], [5, 6], [7, 8]], dtype=np.uint64)
        scores = np.array([0.5, 0.25, 0.5000001, 0.49999997, 0.35], dtype=np.float32)
        for thr in (0.35, 0.5):
            out[f"cc/components_{thr}"] = np.asarray(connected_components(nodes, edges, scores, thr), dtype=np.uint64)
        out["cc/nodes"], out["cc/edges"], out["cc/scores"] = nodes, edges, scores
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(CASES)} graphs x 3 queue / scorer settings + the blockwise call" + (", connected components" if connected_components else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
