"""ORACLE (test infrastructure): numpy / scipy restatement of the 3-D local shape descriptors the reference trains
against (models/3d_mtlsd/train.py:134-141, AddLocalShapeDescriptor -> lsd.train.LsdExtractor.get_descriptors).

**Parity unpinned**: the lsd package (funkelab/lsd, `lsd.train.local_shape_descriptor`) is a third-party dependency that
is neither under /root/reference nor installed here, and no reference test holds vectors for it.  This follows its
published algorithm: per object, Gaussian-weighted (scipy gaussian_filter, mode "constant", truncate 3.0) count, mean
coordinate and coordinate covariance of the object's mask on the `downsample`-times sub-sampled grid, upsampled by
repetition and written where the object is; then normalisation to [0, 1] and clipping.
Only tests/ may import this."""
import numpy as np
from scipy.ndimage import gaussian_filter


def lsd_targets(labels, roi_offset, roi_shape, sigma, voxel_size, downsample=1, unlabelled=None):
    """labels: int [D][H][W] (with context); -> (descriptors float32 [10][d][h][w], mask float32 [10][d][h][w])"""
    df = int(downsample)
    labels = np.asarray(labels)
    sigma = np.asarray(sigma, np.float64)
    vs = np.asarray(voxel_size, np.float64)
    sub = labels[::df, ::df, ::df]
    sub_vs = vs * df
    sub_sigma = sigma / sub_vs
    grid = np.meshgrid(*[np.arange(n, dtype=np.float64) * sub_vs[d] for d, n in enumerate(sub.shape)], indexing="ij")
    coords = np.array(grid)
    sl = tuple(slice(o, o + s) for o, s in zip(roi_offset, roi_shape))
    sub_sl = tuple(slice(o // df, (o + s) // df) for o, s in zip(roi_offset, roi_shape))
    out = np.zeros((10,) + tuple(roi_shape), np.float64)

    def agg(a):
        return gaussian_filter(a, sigma=sub_sigma, mode="constant", cval=0.0, truncate=3.0)[sub_sl]

    def up(a):
        for ax in range(1, 4):
            a = np.repeat(a, df, axis=ax)
        return a
    for l in np.unique(labels[sl]):
        if l == 0:
            continue
        m = (sub == l).astype(np.float64)
        count = agg(m)
        n = count.copy()
        n[n == 0] = 1
        mc = coords * m
        mean = np.array([agg(mc[d]) for d in range(3)]) / n
        offset = mean - coords[(slice(None),) + sub_sl]
        pairs = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
        cov = np.array([agg(mc[i] * coords[j]) for i, j in pairs]) / n
        cov -= np.array([mean[i] * mean[j] for i, j in pairs])
        var, pe = cov[:3].copy(), cov[3:].copy()
        var[var < 1e-3] = 1e-3
        pe[0] /= np.sqrt(var[0] * var[1])
        pe[1] /= np.sqrt(var[0] * var[2])
        pe[2] /= np.sqrt(var[1] * var[2])
        var /= (sigma ** 2)[:, None, None, None]
        d = up(np.concatenate([offset, var, pe, count[None]]))
        out += d * (labels[sl] == l)
    fg = labels[sl] != 0
    out[[0, 1, 2]] = out[[0, 1, 2]] / sigma[:, None, None, None] * 0.5 + 0.5
    out[[6, 7, 8]] = out[[6, 7, 8]] * 0.5 + 0.5
    out[[0, 1, 2, 6, 7, 8]] *= fg
    np.clip(out, 0.0, 1.0, out=out)
    mask = fg.astype(np.float32)
    if unlabelled is not None:
        mask = mask * (np.asarray(unlabelled)[sl] > 0)
    return out.astype(np.float32), np.repeat(mask[None], 10, axis=0).astype(np.float32)
