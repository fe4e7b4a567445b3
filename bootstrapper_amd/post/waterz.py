"""Mirror of the `waterz.agglomerate` calls the reference makes, on the device:
  /root/reference/bootstrapper/post/watershed.py:333-338        thresholds, fragments, scoring function (exact queue)
  /root/reference/bootstrapper/post/blockwise/waterz_agglom.py:131-160   thresholds [0, 1.0], discretize_queue=256,
                                                                 return_merge_history, return_region_graph
  /root/reference/bootstrapper/post/blockwise/watershed_frags.py:165-176 epsilon agglomeration (one threshold, drained)
A generator, one item per threshold (ascending): the segmentation, or (segmentation, merge_history, region_graph) as
requested -- merge_history = [{a, b, c, score}] of the merges since the previous threshold (c = a: the surviving id),
region_graph = [{u, v, score}] of the current graph.  Like waterz, `fragments` is updated IN PLACE and the same array
comes back at every step.  Scoring functions: the mean (post/blockwise/waterz_agglom.py:25, the only one of the blockwise
path) and, for the exact-queue call of post/watershed.py:333-338, the ten histogram-quantile scorers the reference lists
(post/watershed.py:230-243); anything else raises.
"""
import re
import numpy as np
import torch

from .engine import lut_relabel
from .ws import _engine, as_u8_affinities

MEAN = "OneMinus<MeanAffinity<RegionGraphType, ScoreValue>>"
_HIST = re.compile(r"OneMinus<HistogramQuantileAffinity<RegionGraphType,\s*(\d+),\s*ScoreValue,\s*256,\s*(true|false)>>$")

# merge_function names of the segment config -> waterz scoring functions (reference post/watershed.py:230-243)
MERGE_FUNCTIONS = {"mean": MEAN}
for _q in (10, 25, 50, 75, 90):
    MERGE_FUNCTIONS[f"hist_quant_{_q}"] = f"OneMinus<HistogramQuantileAffinity<RegionGraphType, {_q}, ScoreValue, 256, false>>"
    MERGE_FUNCTIONS[f"hist_quant_{_q}_initmax"] = f"OneMinus<HistogramQuantileAffinity<RegionGraphType, {_q}, ScoreValue, 256, true>>"


def _score(sums, counts):
    """OneMinus<MeanAffinity> of (sum in uint8 units, count), as oracle/seg_ref.c edge_score"""
    return (np.float32(1.0) - (sums.astype(np.float64) / (255.0 * counts.astype(np.float64))).astype(np.float32)).astype(np.float32)


def agglomerate(affs, thresholds, fragments=None, scoring_function=MEAN, discretize_queue=0,
                return_merge_history=False, return_region_graph=False, engine=None):
    hist = _HIST.match(scoring_function.replace(" ", "").replace(",", ", ")) if scoring_function != MEAN else None
    if scoring_function != MEAN and hist is None:
        raise NotImplementedError(f"scoring function {scoring_function!r} is not implemented")
    if hist is not None and (discretize_queue or return_merge_history or return_region_graph):
        raise NotImplementedError("the histogram-quantile scorers serve the exact-queue call of post/watershed.py:333-338 only")
    if fragments is None:
        raise NotImplementedError("waterz's own fragment extraction is not used by the reference path")
    thresholds = [float(t) for t in thresholds]
    if any(b < a for a, b in zip(thresholds, thresholds[1:])):
        raise ValueError("thresholds must be ascending")
    a = as_u8_affinities(affs, None)[:3]
    host_out = isinstance(fragments, np.ndarray)
    frag = torch.from_numpy(fragments.view(np.int64)).to(a.device) if host_out else fragments
    if frag.dtype != torch.int64 or tuple(frag.shape) != tuple(a.shape[1:]):
        raise ValueError("fragments must hold 64-bit ids of shape (D, H, W)")
    eng = engine or _engine(a.shape[1:], a.device.index or 0)

    def publish(seg):
        if host_out:
            fragments[...] = seg.cpu().numpy().view(fragments.dtype)
            return fragments
        fragments.copy_(seg)
        return fragments

    if not (discretize_queue or return_merge_history or return_region_graph):
        if hist is not None:
            segs = eng.agglomerate_hist(a, frag.contiguous(), thresholds, int(hist.group(1)), hist.group(2) == "true")
        else:
            segs = eng.agglomerate_mean(a, frag.contiguous(), thresholds)
        eng.status()
        for i in range(len(thresholds)):
            yield publish(segs[i])
        return
    if not discretize_queue:
        raise NotImplementedError("merge history / region graph are produced by the discretized queue (the reference passes "
                                  "discretize_queue=256); the exact queue only yields segmentations")

    frag0 = frag.contiguous().clone()          # every threshold is agglomerated from the fragments (see below)
    done = 0
    for t in thresholds:
        # mergeUntil(t) stops BEFORE it pops anything, so running from scratch up to t gives exactly the merges a
        # continued run would have made by then: history since the previous threshold = the new tail
        if t > 0:
            edges, scores, merges, mscores = eng.rag_merge_scores(a, frag0, t, int(discretize_queue), return_merges=True)
        else:
            edges, scores = eng.rag_merge_scores(a, frag0, 1e-30, int(discretize_queue))
            merges = torch.zeros((0, 2), dtype=torch.int64, device=a.device)
            mscores = torch.zeros(0, dtype=torch.float32, device=a.device)
        e = edges.cpu().numpy().view(np.uint64)
        m = merges.cpu().numpy().view(np.uint64)
        ms = mscores.cpu().numpy()
        # survivors: every absorbed id points at its survivor; chase to the root
        parent = {int(b): int(s) for s, b in m}

        def root(x):
            while x in parent:
                x = parent[x]
            return x
        if len(m):
            keys = np.unique(m[:, 1])
            vals = np.array([root(int(k)) for k in keys], dtype=np.uint64)
            seg = lut_relabel(frag0, torch.from_numpy(keys.view(np.int64)), torch.from_numpy(vals.view(np.int64)))
        else:
            seg = frag0.clone()
        item = [publish(seg)]
        if return_merge_history:
            item.append([{"a": int(s), "b": int(b), "c": int(s), "score": float(sc)} for (s, b), sc in zip(m[done:], ms[done:])])
        done = len(m)
        if return_region_graph:
            sums, counts = eng.rag_edge_stats(len(e))
            acc = {}
            for (u, v), su, c in zip(e.tolist(), sums.tolist(), counts.tolist()):
                ru, rv = root(u), root(v)
                if ru == rv:
                    continue
                k = (min(ru, rv), max(ru, rv))
                p = acc.get(k)
                acc[k] = (su, c) if p is None else (p[0] + su, p[1] + c)
            ks = sorted(acc)
            sc = _score(np.array([acc[k][0] for k in ks], np.uint64), np.array([acc[k][1] for k in ks], np.uint64)) if ks else []
            item.append([{"u": k[0], "v": k[1], "score": float(s)} for k, s in zip(ks, sc)])
        yield item[0] if len(item) == 1 else tuple(item)
