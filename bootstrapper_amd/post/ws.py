"""Mirror of the reference post/ws.py on the device.

Reference: /root/reference/bootstrapper/post/ws.py:38-112 watershed_from_affinities.
Same argument names and return convention.  The kernels work on the uint8 affinities the predict stage stores
(models/3d_affs/net_config.json "dtype": "uint8"); the reference's drivers hand over exactly those as floats
(u8 / 255, post/watershed.py:259-262, watershed_frags.py:198-205) with max_affinity_value = 1.0, which is accepted and
taken back to uint8 when it is exact -- anything else (shifted or smoothed affinities) is refused, not rounded.
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


def watershed_from_affinities(affs, max_affinity_value=1.0, fragments_in_xy=False, return_seeds=False,
                              min_seed_distance=10, engine=None):
    """-> (fragments, max_id[, seeds]) like the reference.  fragments / seeds: int64 CUDA tensors holding the uint64
    ids; max_id: python int (synchronises).  uint8 affinities may be passed with the default max_affinity_value."""
    if isinstance(affs, torch.Tensor) and affs.dtype == torch.uint8 or isinstance(affs, np.ndarray) and affs.dtype == np.uint8:
        if max_affinity_value not in (1.0, 255):
            raise ValueError("uint8 affinities imply max_affinity_value=255")
        a = as_u8_affinities(affs)
    else:
        a = as_u8_affinities(affs, max_affinity_value)
    a = a[-3:] if a.shape[0] > 3 else a
    eng = engine or _engine(a.shape[1:], a.device.index or 0)
    if return_seeds:
        frags, max_id, seeds = eng.ws_fragments(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance, return_seeds=True)
        return frags, int(max_id.item()), seeds
    frags, max_id = eng.ws_fragments(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance)
    return frags, int(max_id.item())
