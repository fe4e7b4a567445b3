"""Mutex-watershed fragments on the device: behavioural mirror of /root/reference/bootstrapper/post/mws.py:12-59.

`mwatershed_from_affinities(affs, neighborhood, bias, sigma, noise_eps, strides, randomized_strides)` adds the same
shift to the float affinities (noise_eps * N(0, 1) + (gaussian_filter(affs, (0, *sigma)) - affs) + bias[c], mws.py:36-49)
and clusters them with libbsmi's `bsmi_mws_agglom_f64` in place of `mwatershed.agglom` (mws.py:51-56; the package is
third party and absent: parity unpinned, oracle/mws_ref.py states the restated rule).  The noise of the reference is
unseeded (`np.random.randn`): reproduced in distribution, not in value; `seed` makes a run repeatable.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .shifts import gaussian_filter


def mws_agglom(affs_f64, offsets, strides=None, randomized_strides=False, seed=0):
    """affs_f64: float64 device tensor [K][D][H][W] (sign = attractive / repulsive) -> int64 device tensor [D][H][W] holding
    the uint64 labels (1 + smallest voxel index of the cluster)."""
    if affs_f64.dtype != torch.float64 or affs_f64.dim() != 4 or not affs_f64.is_cuda:
        raise ValueError("mws_agglom takes a float64 CUDA tensor [K][D][H][W]")
    K = affs_f64.shape[0]
    offs = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 3))
    if offs.shape[0] != K:
        raise ValueError(f"{offs.shape[0]} offsets for {K} affinity channels")
    st = None
    if strides is not None:
        st = np.ascontiguousarray(np.asarray(strides, dtype=np.int32).reshape(-1, 3))
        if st.shape[0] != K:
            raise ValueError(f"{st.shape[0]} strides for {K} affinity channels")
    a = affs_f64.contiguous()
    shape = (C.c_int64 * 3)(*a.shape[1:])
    out = torch.empty(tuple(a.shape[1:]), dtype=torch.int64, device=a.device)
    rs = 0
    if randomized_strides and st is not None:
        rs = (int(seed) % 0xfffffffe) + 1
    stream = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(_lib.lib.bsmi_mws_agglom_f64(a.device.index or 0, a.data_ptr(), K, offs.ctypes.data, st.ctypes.data if st is not None else None,
                                           rs, shape, out.data_ptr(), stream))
    return out


def mws_cluster(n_nodes, edges, scores):
    """Mutex watershed of a graph on the host (volara GraphMWS): edges [m][2] of node indices, scores [m] -> uint64 [n_nodes]."""
    e = np.ascontiguousarray(np.asarray(edges, dtype=np.uint64).reshape(-1, 2))
    s = np.ascontiguousarray(np.asarray(scores, dtype=np.float64).reshape(-1))
    if len(e) != len(s):
        raise ValueError("one score per edge")
    out = np.zeros(int(n_nodes), dtype=np.uint64)
    _lib.check(_lib.lib.bsmi_mws_cluster(int(n_nodes), e.ctypes.data if len(e) else None, s.ctypes.data if len(s) else None, len(e),
                                        out.ctypes.data if n_nodes else None))
    return out


def shifted(affs, bias, sigma=None, noise_eps=None, generator=None):
    """affs (float64 device tensor [K][...]) + the shift of mws.py:36-49."""
    shift = torch.zeros_like(affs)
    if noise_eps is not None:
        shift += torch.randn(affs.shape, dtype=torch.float64, device=affs.device, generator=generator) * float(noise_eps)
    if sigma is not None:
        shift += gaussian_filter(affs, (0, *[float(s) for s in sigma])) - affs
    shift += torch.tensor([float(b) for b in bias], dtype=torch.float64, device=affs.device).view(-1, *([1] * (affs.dim() - 1)))
    return affs + shift


def mwatershed_from_affinities(affs, neighborhood, bias, sigma=None, noise_eps=None, strides=None, randomized_strides=False,
                               device=0, seed=None):
    """reference post/mws.py:12-59.  affs: float array / tensor [K][D][H][W]; returns the fragments as a uint64 numpy array
    (numpy in, numpy out, like the reference) or an int64 device tensor when a device tensor was passed."""
    as_numpy = not torch.is_tensor(affs)
    dev = torch.device("cuda", device) if as_numpy else affs.device
    a = (torch.from_numpy(np.ascontiguousarray(affs)) if as_numpy else affs).to(dev, torch.float64)
    if len(neighborhood) != a.shape[0] or len(bias) != a.shape[0]:
        raise ValueError("one offset and one bias per affinity channel")
    gen = None
    if seed is not None:
        gen = torch.Generator(device=dev).manual_seed(int(seed))
    x = shifted(a, bias, sigma, noise_eps, gen)
    frags = mws_agglom(x, neighborhood, strides, randomized_strides, seed=seed or 0)
    return frags.cpu().numpy().view(np.uint64) if as_numpy else frags
