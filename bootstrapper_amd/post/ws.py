"""Mirror of the reference post/ws.py on the device.

Reference: /root/reference/bootstrapper/post/ws.py:38-112 watershed_from_affinities.
Same argument names and return convention; inputs are uint8 CUDA affinities (the dtype
the predict stage stores, models/3d_affs/net_config.json "dtype": "uint8").
"""
import torch

from .engine import SegEngine

_ENGINES = {}


def _engine(shape, device):
    key = (device, tuple(shape))
    if key not in _ENGINES:
        _ENGINES[key] = SegEngine(shape, device)
    return _ENGINES[key]


def watershed_from_affinities(affs, max_affinity_value=255, fragments_in_xy=False, return_seeds=False,
                              min_seed_distance=10, engine=None):
    """-> (fragments, max_id) like the reference.  fragments: int64 CUDA tensor holding the
    uint64 ids; max_id: python int (synchronises)."""
    if affs.dtype != torch.uint8:
        raise TypeError("the device path takes the uint8 affinities the predict stage stores; "
                        "float affinities (max_affinity_value=1.0) are not implemented")
    if max_affinity_value != 255:
        raise ValueError("uint8 affinities imply max_affinity_value=255")
    if return_seeds:
        raise NotImplementedError("return_seeds=True is not implemented on the device")
    a = affs[-3:] if affs.shape[0] > 3 else affs
    eng = engine or _engine(a.shape[1:], a.device.index or 0)
    frags, max_id = eng.ws_fragments(a, fragments_in_xy=fragments_in_xy, min_seed_distance=min_seed_distance)
    return frags, int(max_id.item())
