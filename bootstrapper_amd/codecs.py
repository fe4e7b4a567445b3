"""numcodecs-compatible chunk codecs on top of libbsmi's volume I/O entry points (include/bsmi_io.h).

The reference never names a codec: it inherits zarr-python's default (Blosc lz4, clevel 5, byte
shuffle) through funlib.persistence.prepare_ds (/root/reference/bootstrapper/predict.py:169-178,
post/watershed.py:319-330), and reads whatever the user's volumes were written with
(data/volumes.py:14-19).  `from_config` maps the `compressor` entry of a `.zarray` to a bsmi_codec.
"""
import ctypes as C

import numpy as np

from . import _lib

_BLOSC_CNAMES = {"lz4": _lib.BLOSC_LZ4, "lz4hc": _lib.BLOSC_LZ4, "zlib": _lib.BLOSC_ZLIB, "zstd": _lib.BLOSC_ZSTD}

# zarr-python's default for new arrays (zarr/storage.py default_compressor)
DEFAULT_COMPRESSOR = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}


def from_config(comp, itemsize):
    """`.zarray["compressor"]` (dict or None) -> _lib.Codec.  Unknown codecs raise NotImplementedError."""
    if comp is None:
        return _lib.Codec(_lib.CODEC_RAW, 0, 0, 0, itemsize, 0)
    cid = comp.get("id")
    if cid == "zlib":
        return _lib.Codec(_lib.CODEC_ZLIB, int(comp.get("level", 1)), 0, 0, itemsize, 0)
    if cid == "gzip":
        return _lib.Codec(_lib.CODEC_GZIP, int(comp.get("level", 1)), 0, 0, itemsize, 0)
    if cid == "zstd":
        return _lib.Codec(_lib.CODEC_ZSTD, int(comp.get("level", 1)), 0, 0, itemsize, 0)
    if cid == "lz4":
        return _lib.Codec(_lib.CODEC_LZ4, int(comp.get("acceleration", 1)), 0, 0, itemsize, 0)
    if cid == "blosc":
        cname = comp.get("cname", "lz4")
        if cname not in _BLOSC_CNAMES and cname != "blosclz":
            raise NotImplementedError(f"blosc inner codec {cname!r} is not supported (lz4, lz4hc, zlib, zstd; blosclz read-only)")
        shuffle = int(comp.get("shuffle", 1))
        if shuffle == -1:  # numcodecs AUTOSHUFFLE
            shuffle = 2 if itemsize == 1 else 1
        # blosclz frames can be read; new chunks of such an array are written as lz4 frames, which every
        # Blosc reader accepts (the inner codec is recorded per frame, not per array)
        return _lib.Codec(_lib.CODEC_BLOSC, int(comp.get("clevel", 5)), _BLOSC_CNAMES.get(cname, _lib.BLOSC_LZ4), shuffle,
                          itemsize, int(comp.get("blocksize", 0)))
    raise NotImplementedError(
        f"zarr compressor {cid!r} is not supported (null, zlib, gzip, zstd, lz4, blosc)")


def decode(codec, buf, nbytes):
    """bytes-like -> uint8 numpy array of at most `nbytes` decoded bytes."""
    src = np.frombuffer(buf, dtype=np.uint8)
    dst = np.empty(nbytes, dtype=np.uint8)
    n = C.c_size_t()
    _lib.check(_lib.lib.bsmi_codec_decode(C.byref(codec), src.ctypes.data, src.size, dst.ctypes.data, dst.size, C.byref(n)))
    return dst[: n.value]


def encode(codec, buf):
    src = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    cap = _lib.lib.bsmi_codec_bound(C.byref(codec), src.size)
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_size_t()
    _lib.check(_lib.lib.bsmi_codec_encode(C.byref(codec), src.ctypes.data, src.size, dst.ctypes.data, dst.size, C.byref(n)))
    return dst[: n.value].tobytes()


def read_chunks(codec, paths, chunk_nbytes, threads=8):
    """Read + decode chunk files side by side.  Returns [uint8 array of chunk_nbytes, or None where the file is missing]."""
    n = len(paths)
    if n == 0:
        return []
    bufs = [np.empty(chunk_nbytes, dtype=np.uint8) for _ in range(n)]
    cpaths = (C.c_char_p * n)(*[p.encode() for p in paths])
    dsts = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    caps = (C.c_size_t * n)(*([chunk_nbytes] * n))
    lens = (C.c_size_t * n)()
    status = (C.c_int * n)()
    _lib.check(_lib.lib.bsmi_chunks_read(C.byref(codec), n, cpaths, dsts, caps, lens, status, int(threads)))
    out = []
    for i in range(n):
        if status[i] == _lib.CHUNK_MISSING:
            out.append(None)
        elif lens[i] != chunk_nbytes:
            raise ValueError(f"{paths[i]}: chunk decodes to {lens[i]} bytes, expected {chunk_nbytes}")
        else:
            out.append(bufs[i])
    return out


def write_chunks(codec, paths, arrays, threads=8):
    """Encode + write whole chunks (C-contiguous numpy arrays) side by side."""
    n = len(paths)
    if n == 0:
        return
    arrays = [np.ascontiguousarray(a) for a in arrays]
    cpaths = (C.c_char_p * n)(*[p.encode() for p in paths])
    srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
    sizes = (C.c_size_t * n)(*[a.nbytes for a in arrays])
    status = (C.c_int * n)()
    _lib.check(_lib.lib.bsmi_chunks_write(C.byref(codec), n, cpaths, srcs, sizes, status, int(threads)))
