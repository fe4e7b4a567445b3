"""Affinity shifts in front of the seeded watershed, on the device.

Reference: /root/reference/bootstrapper/post/watershed.py:285-303 and post/blockwise/watershed_frags.py:116-131 add
  noise_eps * N(0, 1)  +  (gaussian_filter(affs, (0, *sigma)) - affs)  +  bias[c]
to the float affinities before `watershed_from_affinities`.  post/ws.py reads affinities only to threshold their mean
(`0.5 (a_y + a_x) > 0.5` per section, or the mean of the three channels for the 3-D mode, ws.py:64-77,100) -- everything
after that works on the boolean mask -- so the shifted watershed is the unshifted kernel applied to the mask of the
shifted affinities.  This module computes that mask with torch device arithmetic (the Gaussian as scipy.ndimage does it:
separable, truncated at 4 sigma, reflecting borders, one pass per axis) and hands it on as 0 / 255 affinities.
The noise of the reference is unseeded (`np.random.randn`): it is reproduced in distribution, not in value.
`seed_eps` (watershed_frags.py:133-141: the affinities decay with the distance from the seeds of a 3-D boundary distance
transform) uses the exact Euclidean transform and the maximum filter below, which restate the scipy calls of the reference.
"""
import math

import torch


def _gauss_weights(sigma, dtype, device):
    radius = int(4.0 * float(sigma) + 0.5)  # scipy.ndimage.gaussian_filter1d: truncate = 4.0
    x = torch.arange(-radius, radius + 1, dtype=torch.float64, device=device)
    w = torch.exp(-0.5 * (x / float(sigma)) ** 2)
    return (w / w.sum()).to(dtype), radius


def _filter_axis(t, axis, sigma):
    """correlate1d along `axis` with scipy's border mode 'reflect' (d c b a | a b c d | d c b a)"""
    if not sigma:
        return t
    w, r = _gauss_weights(sigma, torch.float64, t.device)
    n = t.shape[axis]
    idx = torch.arange(-r, n + r, device=t.device)
    period = 2 * n
    idx = idx % period
    idx = torch.where(idx >= n, period - 1 - idx, idx)
    padded = t.index_select(axis, idx).to(torch.float64)
    out = torch.zeros_like(t, dtype=torch.float64)
    for k in range(2 * r + 1):
        out += w[k] * padded.narrow(axis, k, n)
    return out.to(t.dtype)  # scipy stores every pass in the output dtype


def gaussian_filter(t, sigma):
    """scipy.ndimage.gaussian_filter(t, sigma) for a sequence `sigma` (one entry per axis, 0 = leave the axis alone)"""
    for axis in range(t.dim()):
        t = _filter_axis(t, axis, sigma[axis])
    return t


def _envelope_axis(f, axis, chunk_bytes=1 << 29):
    """g[i] = min_j f[j] + (i - j)^2 along `axis` (one pass of the separable squared Euclidean distance transform)."""
    f = f.movedim(axis, -1).contiguous()
    n = f.shape[-1]
    idx = torch.arange(n, dtype=f.dtype, device=f.device)
    cost = (idx.view(n, 1) - idx.view(1, n)) ** 2                     # [i][j]
    flat = f.reshape(-1, n)
    out = torch.empty_like(flat)
    rows = max(1, chunk_bytes // (n * n * flat.element_size()))
    for r0 in range(0, flat.shape[0], rows):
        blk = flat[r0:r0 + rows]
        out[r0:r0 + rows] = (blk.unsqueeze(1) + cost.unsqueeze(0)).amin(dim=2)
    return out.reshape(f.shape).movedim(-1, axis)


def distance_transform_edt(mask):
    """scipy.ndimage.distance_transform_edt(mask) for a bool device tensor: the exact Euclidean distance of every True
    element to the nearest False one (float64; squared distances are integers, the root is taken once at the end)."""
    big = float(sum(int(n) ** 2 for n in mask.shape) + 1)           # larger than any squared distance inside the array
    d2 = torch.where(mask, torch.full((), big, dtype=torch.float64, device=mask.device),
                     torch.zeros((), dtype=torch.float64, device=mask.device)).expand(mask.shape).contiguous()
    for axis in range(mask.dim()):
        d2 = _envelope_axis(d2, axis)
    # correctly rounded roots (as the C sqrt of scipy's transform): the squared distances are small integers, so through a
    # table made by numpy on the host rather than through the device's sqrt
    import numpy as np
    roots = torch.from_numpy(np.sqrt(np.arange(int(big) * mask.dim() + 1, dtype=np.float64))).to(mask.device)
    return roots[d2.to(torch.int64)]


def maximum_filter(t, size):
    """scipy.ndimage.maximum_filter(t, size) (cubic window, mode 'reflect', origin 0: offsets -size//2 .. size-size//2-1)."""
    size = int(size)
    lo = size // 2
    for axis in range(t.dim()):
        n = t.shape[axis]
        idx = torch.arange(-lo, n + size - lo - 1, device=t.device)
        period = 2 * n
        idx = idx % period
        idx = torch.where(idx >= n, period - 1 - idx, idx)
        padded = t.index_select(axis, idx)
        out = padded.narrow(axis, 0, n)
        for k in range(1, size):
            out = torch.maximum(out, padded.narrow(axis, k, n))
        t = out
    return t


def seed_distance(a, min_seed_distance):
    """watershed_frags.py:133-141: distance of every voxel from the seeds (the plateau maxima of the 3-D distance
    transform of the boundary mask mean(affs) > 0.5, inside the mask)."""
    boundary_mask = a.mean(dim=0) > 0.5
    dist = distance_transform_edt(boundary_mask)
    seeds = (maximum_filter(dist, min_seed_distance) == dist) & boundary_mask
    return distance_transform_edt(~seeds)


def shifted_affinities(affs_u8, sigma=None, noise_eps=None, bias=None, dtype=torch.float32, generator=None, seed_eps=None,
                       min_seed_distance=10):
    """u8 / 255 affinities (first three channels) plus the shift, as `dtype` (float32: post/watershed.py:259-262,
    float64: watershed_frags.py:198-205)."""
    a = affs_u8[:3].to(dtype) / 255.0
    shift = torch.zeros_like(a)
    if noise_eps is not None:
        shift += torch.randn(a.shape, dtype=torch.float64, device=a.device, generator=generator).to(dtype) * float(noise_eps)
    if sigma is not None:
        shift += gaussian_filter(a, (0, *[float(s) for s in sigma])) - a
    if bias is not None:
        b = [float(bias)] * a.shape[0] if isinstance(bias, (int, float)) else [float(x) for x in bias]
        if len(b) != a.shape[0]:
            raise ValueError(f"bias has {len(b)} entries for {a.shape[0]} affinity channels")
        shift += torch.tensor(b, dtype=dtype, device=a.device).view(-1, 1, 1, 1)
    if seed_eps is not None:
        shift -= (float(seed_eps) * seed_distance(a, min_seed_distance)).to(dtype)
    return a + shift


def boundary_mask_affinities(affs_u8, fragments_in_xy, sigma=None, noise_eps=None, bias=None, dtype=torch.float32, generator=None,
                             seed_eps=None, min_seed_distance=10):
    """uint8 [3][D][H][W] holding 255 where the shifted affinities pass ws.py's threshold and 0 elsewhere: the seeded
    watershed kernels read exactly the same mask from it as ws.py does from the shifted floats."""
    x = shifted_affinities(affs_u8, sigma, noise_eps, bias, dtype, generator, seed_eps, min_seed_distance)
    if fragments_in_xy:
        mask = 0.5 * (x[-1] + x[-2]) > 0.5          # ws.py:64,77 with max_affinity_value = 1.0
    else:
        mask = x.mean(dim=0) > 0.5                  # ws.py:100
    return (mask.to(torch.uint8) * 255)[None].expand(3, -1, -1, -1).contiguous()
