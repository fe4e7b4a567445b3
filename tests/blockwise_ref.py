"""The blockwise segmentation pipeline (reference post/watershed.py:8-203) composed from the CPU oracle's pieces: the
checker of the drivers and of bootstrapper_amd.volume.  Test infrastructure only."""
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


def cpu_blockwise(affs, block, ctx, msd, ff, rd, thresholds, bins=256):
    """The blockwise pipeline (reference post/watershed.py:8-203) composed from the oracle's pieces."""
    from oracle import seg_ref as S
    total = affs.shape[1:]
    grid = [range(0, t, b) for t, b in zip(total, block)]
    boxes = [((z, y, x), (min(z + block[0], total[0]), min(y + block[1], total[1]), min(x + block[2], total[2])))
             for z in grid[0] for y in grid[1] for x in grid[2]]
    nv = int(np.prod(block))
    frags = np.zeros(total, np.uint64)
    sizes = {}
    for bi, (b, e) in enumerate(boxes):
        rb, re = tuple(v - c for v, c in zip(b, ctx)), tuple(v + c for v, c in zip(e, ctx))
        a = pad_read(affs, rb, re, lead=True)
        if a.max() == 0:
            continue
        fr, _ = S.ws_fragments_u8(a, True, msd)
        fr = S.filter_fragments_u8(a, fr, ff, rd)
        crop = np.ascontiguousarray(fr[tuple(slice(ctx[d], ctx[d] + e[d] - b[d]) for d in range(3))])
        lab, n = S.label26(crop)
        assert n < nv
        frags[tuple(slice(b[d], e[d]) for d in range(3))] = np.where(lab > 0, lab.astype(np.uint64) + np.uint64(bi * nv), np.uint64(0))
    E, Sc = [], []
    for bi, (b, e) in enumerate(boxes):
        rb, re = tuple(v - c for v, c in zip(b, ctx)), tuple(v + c for v, c in zip(e, ctx))
        f = pad_read(frags, rb, re)
        if not f.any():
            continue
        ed, sc, _, _ = S.rag_merge_scores_u8(pad_read(affs, rb, re, lead=True), f, 1.0, bins)
        own = (ed[:, 0] - np.uint64(1)) // np.uint64(nv) == np.uint64(bi)
        E.append(ed[own])
        Sc.append(sc[own])
    E, Sc = np.concatenate(E), np.concatenate(Sc)
    nodes = np.unique(frags)
    nodes = nodes[nodes > 0]
    keep = ~np.isnan(Sc)
    segs = []
    for thr in thresholds:
        comp = S.connected_components(nodes, E[keep], Sc[keep], thr)
        idx = np.searchsorted(nodes, frags)
        idx[idx >= len(nodes)] = 0
        segs.append(np.where(frags > 0, comp[idx], np.uint64(0)))
    return frags, nodes, E, Sc, segs


