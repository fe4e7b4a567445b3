"""Mirror of the `waterz.agglomerate` call the reference makes
(/root/reference/bootstrapper/post/watershed.py:333-338), on the device.

Only the scoring function the reference enables is implemented
(post/blockwise/waterz_agglom.py:25 "mean"); anything else raises.
"""
import torch

from .ws import _engine

MEAN = "OneMinus<MeanAffinity<RegionGraphType, ScoreValue>>"


def agglomerate(affs, thresholds, fragments=None, scoring_function=MEAN, discretize_queue=0,
                return_merge_history=False, return_region_graph=False, engine=None):
    """Generator: one int64 CUDA segmentation (uint64 ids) per threshold, ascending."""
    if scoring_function != MEAN:
        raise NotImplementedError(f"scoring function {scoring_function!r} is not implemented")
    if discretize_queue:
        raise NotImplementedError("discretize_queue != 0 is not implemented yet")
    if return_merge_history or return_region_graph:
        raise NotImplementedError("merge history / region graph output is not implemented yet")
    if fragments is None:
        raise NotImplementedError("waterz's own fragment extraction is not used by the reference path")
    if affs.dtype != torch.uint8:
        raise TypeError("the device path takes uint8 affinities")
    a = affs[:3]
    eng = engine or _engine(a.shape[1:], a.device.index or 0)
    segs = eng.agglomerate_mean(a, fragments, list(thresholds))
    eng.status()
    for i in range(len(thresholds)):
        yield segs[i]
