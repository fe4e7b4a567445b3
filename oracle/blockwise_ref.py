"""ORACLE (test infrastructure): the blockwise segmentation pipeline of the reference (post/watershed.py:8-203,
post/blockwise/watershed_frags.py:196-246, post/blockwise/waterz_agglom.py:106-170) composed from the pieces of
oracle/seg_ref.c -- the checker of the drivers and of bootstrapper_amd.volume, and the CPU leg bench.py times.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import numpy as np


def pad_read(vol, begin, end, lead=False):
    """zeros outside the volume, like `to_ndarray(roi, fill_value=0)`"""
    shape = vol.shape[1:] if lead else vol.shape
    out = np.zeros((vol.shape[:1] if lead else ()) + tuple(e - b for b, e in zip(begin, end)), vol.dtype)
    src = tuple(slice(max(b, 0), min(e, n)) for b, e, n in zip(begin, end, shape))
    dst = tuple(slice(s.start - b, s.stop - b) for s, b in zip(src, begin))
    if lead:
        out[(slice(None),) + dst] = vol[(slice(None),) + src]
    else:
        out[dst] = vol[src]
    return out


def _boxes(total, block):
    grid = [range(0, t, b) for t, b in zip(total, block)]
    return [((z, y, x), (min(z + block[0], total[0]), min(y + block[1], total[1]), min(x + block[2], total[2])))
            for z in grid[0] for y in grid[1] for x in grid[2]]


def shifted_mask_affinities(a, fragments_in_xy=True, sigma=None, bias=None, dtype=np.float64, seed_eps=None, min_seed_distance=10):
    """watershed_frags.py:116-131 / post/watershed.py:285-303 with scipy, then ws.py's threshold (ws.py:64-77,100): the mask
    of the shifted affinities as 0 / 255 pseudo-affinities (post/ws.py reads nothing else from them)."""
    from scipy.ndimage import gaussian_filter
    x = a[:3].astype(dtype) / dtype(255.0)
    shift = np.zeros_like(x)
    if sigma is not None:
        shift += gaussian_filter(x, sigma=(0, *sigma)) - x
    if bias is not None:
        b = list(bias) if isinstance(bias, (list, tuple)) else [bias] * x.shape[0]
        shift += np.array([b]).reshape((-1, 1, 1, 1)).astype(dtype)
    if seed_eps is not None:   # watershed_frags.py:133-141, the scipy calls of the reference
        from scipy.ndimage import distance_transform_edt, label, maximum_filter
        boundary_mask = np.mean(x, axis=0) > 0.5
        boundary_distances = distance_transform_edt(boundary_mask)
        max_filtered = maximum_filter(boundary_distances, min_seed_distance)
        seeds, _ = label(max_filtered == boundary_distances)
        seeds[~boundary_mask] = 0
        shift -= (seed_eps * distance_transform_edt(seeds == 0)).astype(dtype)
    x = x + shift
    mask = (0.5 * (x[-1] + x[-2]) > 0.5) if fragments_in_xy else (np.mean(x, axis=0) > 0.5)
    return np.repeat((mask.astype(np.uint8) * 255)[None], 3, axis=0)


def epsilon_agglomerate(a, frags, epsilon, bins=256):
    """watershed_frags.py:158-177: waterz.agglomerate(thresholds=[epsilon], discretize_queue=256) written back into the fragments"""
    from oracle import seg_ref as S
    _, _, merges, _ = S.rag_merge_scores_u8(a, frags, epsilon, bins)
    parent = {int(b): int(s) for s, b in merges}
    out = frags.copy()
    for b in parent:
        r = b
        while r in parent:
            r = parent[r]
        out[frags == np.uint64(b)] = np.uint64(r)
    return out


def cpu_blockwise(affs, block, ctx, msd, ff, rd, thresholds, bins=256, workers=1, epsilon=0.0, sigma=None, bias=None, seed_eps=None):
    """-> (fragments u64, nodes, edges, scores, [segmentation per threshold]).  workers > 1: the blocks of a stage run on
    a thread pool (the C calls release the GIL), as the reference's stages run on daisy workers."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import seg_ref as S
    total = affs.shape[1:]
    boxes = _boxes(total, block)
    nv = int(np.prod(block))
    frags = np.zeros(total, np.uint64)

    def frag_block(bi):
        b, e = boxes[bi]
        rb, re = tuple(v - c for v, c in zip(b, ctx)), tuple(v + c for v, c in zip(e, ctx))
        a = pad_read(affs, rb, re, lead=True)
        if a.max() == 0:
            return
        src = a if sigma is None and bias is None and seed_eps is None else shifted_mask_affinities(a, True, sigma, bias, seed_eps=seed_eps,
                                                                                                     min_seed_distance=msd)
        fr, _ = S.ws_fragments_u8(src, True, msd)
        if epsilon > 0:
            fr = epsilon_agglomerate(a, fr, epsilon)
        fr = S.filter_fragments_u8(a, fr, ff, rd)
        crop = np.ascontiguousarray(fr[tuple(slice(ctx[d], ctx[d] + e[d] - b[d]) for d in range(3))])
        lab, n = S.label26(crop)
        assert n < nv
        frags[tuple(slice(b[d], e[d]) for d in range(3))] = np.where(lab > 0, lab.astype(np.uint64) + np.uint64(bi * nv), np.uint64(0))

    def score_block(bi):
        b, e = boxes[bi]
        rb, re = tuple(v - c for v, c in zip(b, ctx)), tuple(v + c for v, c in zip(e, ctx))
        f = pad_read(frags, rb, re)
        if not f.any():
            return None
        ed, sc, _, _ = S.rag_merge_scores_u8(pad_read(affs, rb, re, lead=True), f, 1.0, bins)
        own = (ed[:, 0] - np.uint64(1)) // np.uint64(nv) == np.uint64(bi)
        return ed[own], sc[own]

    if workers > 1:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(frag_block, range(len(boxes))))
            scored = list(ex.map(score_block, range(len(boxes))))
    else:
        for bi in range(len(boxes)):
            frag_block(bi)
        scored = [score_block(bi) for bi in range(len(boxes))]
    scored = [r for r in scored if r is not None]
    E = np.concatenate([r[0] for r in scored]) if scored else np.zeros((0, 2), np.uint64)
    Sc = np.concatenate([r[1] for r in scored]) if scored else np.zeros(0, np.float32)
    nodes = np.unique(frags)
    nodes = nodes[nodes > 0]
    keep = ~np.isnan(Sc)
    segs = []
    for thr in thresholds:
        comp = S.connected_components(nodes, E[keep], Sc[keep], thr) if len(nodes) else nodes
        idx = np.searchsorted(nodes, frags)
        idx[idx >= len(nodes)] = 0
        segs.append(np.where(frags > 0, comp[idx], np.uint64(0)) if len(nodes) else frags.copy())
    return frags, nodes, E, Sc, segs
