"""Mirror of the reference post/ws.py on the device.

Reference: /root/reference/bootstrapper/post/ws.py:38-112 watershed_from_affinities.
Same argument names and return convention.  The kernels work on the uint8 affinities the predict stage stores
(models/3d_affs/net_config.json "dtype": "uint8"); the reference's drivers hand over exactly those as floats
(u8 / 255, post/watershed.py:259-262, watershed_frags.py:198-205) with max_affinity_value = 1.0, which is accepted and
taken back to uint8 when it is exact; any other input (shifted or smoothed floats, two or more than three channels) is reduced
to ws.py's own boundary mask first, in the input's precision (`mask_affinities`).
"""
import numpy as np
import torch

from .engine import SegEngine

_ENGINES = {}


def _engine(shape, device):
    key = (device, tuple(shape))
    if key not in _ENGINES:
        _ENGINES[key] = SegEngine(shape, device)
    return _ENGINES[key]


def as_u8_affinities(affs, max_affinity_value=None, device=0):
    """uint8 CUDA tensor of `affs`: uint8 input as it is; float input (torch or numpy) only if every value is v / 255
    (* max_affinity_value) for an integer v, compared in the input's own precision."""
    t = torch.from_numpy(np.ascontiguousarray(affs)) if isinstance(affs, np.ndarray) else affs
    if not t.is_cuda:
        t = t.to(torch.device("cuda", device))
    if t.dtype == torch.uint8:
        if max_affinity_value not in (None, 255):
            raise ValueError("uint8 affinities imply max_affinity_value=255")
        return t
    if not t.dtype.is_floating_point:
        raise TypeError(f"affinities of dtype {t.dtype} are not supported")
    scale = 1.0 if max_affinity_value is None else float(max_affinity_value)
    q = torch.round(t.to(torch.float64) * (255.0 / scale)).clamp_(0, 255)
    # u8 / 255 as the reference computes it, in float32 (post/watershed.py:259-262) or float64 (watershed_frags.py:198-205),
    # is within one float32 ulp of q / 255; shifted or smoothed affinities are nowhere near that
    exact = bool(((t.to(torch.float64) * (255.0 / scale) - q).abs().max() <= 255.0 * 1.2e-7).item()) if t.numel() else True
    if not exact:
        raise ValueError("float affinities must be exactly uint8 / 255 (the values `bs predict` stores); shifted, smoothed or "
                         "otherwise continuous affinities are not supported by the device kernels")
    return q.to(torch.uint8)


def mask_affinities(affs, max_affinity_value, fragments_in_xy):
    """The boundary mask exactly as ws.py computes it from arbitrary affinities (ws.py:64,77: `0.5 * (affs[-1] + affs[-2]) >
    0.5 * max_affinity_value` per section; ws.py:100: `np.mean(affs, axis=0) > 0.5 * max_affinity_value` over ALL channels),
    in the input's own precision and numpy's summation order (channel after channel, one division), handed on as 0 / 255
    affinities: everything after the threshold works on the mask, so the kernels read the same mask from it."""
    t = affs
    if t.dtype == torch.uint8 or not t.dtype.is_floating_point:
        s = t.to(torch.int64).sum(dim=0) if not fragments_in_xy else t[-1].to(torch.int64) + t[-2].to(torch.int64)
        n = t.shape[0] if not fragments_in_xy else 2
        mask = 2 * s > int(round(float(max_affinity_value))) * n       # mean > max / 2, in exact integers
    elif fragments_in_xy:
        mask = 0.5 * (t[-1] + t[-2]) > 0.5 * max_affinity_value
    else:
        acc = t[0]
        for c in range(1, t.shape[0]):
            acc = acc + t[c]
        mask = acc / t.shape[0] > 0.5 * max_affinity_value
    return (mask.to(torch.uint8) * 255)[None].expand(3, -1, -1, -1).contiguous()


def watershed_from_affinities(affs, max_affinity_value=1.0, fragments_in_xy=False, return_seeds=False,
                              min_seed_distance=10, engine=None):
    """-> (fragments, max_id[, seeds]) like the reference.  fragments / seeds: int64 CUDA tensors holding the uint64
    ids; max_id: python int (synchronises).  uint8 affinities may be passed with the default max_affinity_value.
    Three channels of uint8 values (or of exactly uint8 / 255 floats, what the reference's drivers hand over) go to the kernels
    as they are; anything else -- more or fewer channels, shifted / smoothed / continuous floats -- through `mask_affinities`."""
    t = torch.from_numpy(np.ascontiguousarray(affs)) if isinstance(affs, np.ndarray) else affs
    if not t.is_cuda:
        t = t.to(torch.device("cuda", 0))
    if t.dim() != 4:
        raise ValueError("affinities must have shape (C, D, H, W)")
    is_u8 = t.dtype == torch.uint8
    if is_u8 and max_affinity_value not in (1.0, 255):
        raise ValueError("uint8 affinities imply max_affinity_value=255")
    a = None
    if t.shape[0] == 3 or (fragments_in_xy and t.shape[0] > 3):
        try:
            a = as_u8_affinities(t[-3:], None if is_u8 else max_affinity_value)
        except ValueError:
            a = None
    if a is None:
        a = mask_affinities(t, 255 if is_u8 else max_affinity_value, fragments_in_xy)
    eng = engine or _engine(a.shape[1:], a.device.index or 0)
    if return_seeds:
        frags, max_id, seeds = eng.ws_fragments(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance, return_seeds=True)
        return frags, int(max_id.item()), seeds
    frags, max_id = eng.ws_fragments(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance)
    return frags, int(max_id.item())
