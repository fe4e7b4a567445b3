"""Blosc frames made on the device (csrc/blosc_dev.hip: `bsmi_blosc_encode_dev_u64`, what `bs segment` writes its label volumes
with) decode -- through this repository's host codec AND through c-blosc itself where the image has it -- to the chunks they
were made from: label volumes (long runs), noise (planes stored verbatim, whole frames stored), chunks that overhang the array
(fill 0), strided sources.  Needs an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REAL_BLOSC = "/opt/conda/lib/libblosc.so.1"


def _encode(vol, chunk, view=None):
    """frames of every chunk of the device tensor `vol` (or of the strided view `view` of it) -> [(grid index, bytes)]"""
    from bootstrapper_amd import _lib
    src = vol if view is None else view
    cz, cy, cx = chunk
    grid = [(z, y, x) for z in range(0, src.shape[0], cz) for y in range(0, src.shape[1], cy) for x in range(0, src.shape[2], cx)]
    n = len(grid)
    cb = cz * cy * cx * 8
    origins = (C.c_int64 * (3 * n))(*[v for g in grid for v in g])
    extents = (C.c_int64 * (3 * n))(*[min(c, s - o) for g in grid for o, c, s in zip(g, chunk, src.shape)])
    slot = (int(_lib.lib.bsmi_blosc_dev_frame_bound(cb)) + 255) // 256 * 256
    ns = int(_lib.lib.bsmi_blosc_dev_scratch_bytes(n, cb))
    scratch = torch.empty(ns, dtype=torch.uint8, device="cuda")
    frames = torch.zeros(n * slot, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert src.stride(2) == 1
    _lib.check(_lib.lib.bsmi_blosc_encode_dev_u64(0, C.c_void_p(src.data_ptr()), src.stride(0), src.stride(1), n, origins, extents,
                                                   (C.c_int64 * 3)(*chunk), C.c_void_p(scratch.data_ptr()), ns, C.c_void_p(frames.data_ptr()), slot,
                                                   C.c_void_p(sizes.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy()
    fr = frames.cpu().numpy()
    return [(g, fr[i * slot:i * slot + int(sz[i])].tobytes()) for i, g in enumerate(grid)]


def _expected(host, g, chunk):
    out = np.zeros(chunk, np.uint64)
    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(g, chunk, host.shape))
    part = host[sl]
    out[:part.shape[0], :part.shape[1], :part.shape[2]] = part
    return out


def _labels(rng, shape, run):
    """label volume: blocks of equal ids of about `run` voxels along x, ids with bytes set in every plane of the low 5"""
    z, y, x = shape
    ids = rng.integers(1, 1 << 38, size=(z, y, max(1, x // run + 2)), dtype=np.uint64)
    rep = np.repeat(ids, run, axis=2)[:, :, :x]
    rep[rng.random(shape) < 0.02] = 0
    return np.ascontiguousarray(rep)


@pytest.mark.parametrize("shape,chunk,kind", [
    ((16, 128, 128), (8, 64, 64), "labels"),          # one 256 KiB block per chunk
    ((40, 100, 150), (16, 64, 64), "labels"),         # two blocks per chunk, chunks that overhang the array in all three axes
    ((128, 128, 128), (128, 128, 128), "labels"),     # the benchmark's chunk: 64 blocks
    ((16, 64, 64), (8, 64, 64), "noise"),             # nothing compresses: the frame is stored
    ((16, 64, 128), (8, 64, 64), "mixed"),            # low planes noise, high planes constant: planes stored verbatim inside an lz4 frame
    ((8, 64, 64), (8, 64, 64), "zeros"),
    ((8, 64, 64), (8, 64, 64), "runs_at_the_end"),    # a run that reaches the plane's last bytes (LZ4's end-of-block rules)
])
def test_device_frames_decode_to_the_chunks(shape, chunk, kind):
    from bootstrapper_amd import _lib, codecs
    rng = np.random.default_rng(len(kind) + shape[0])
    if kind == "labels":
        host = _labels(rng, shape, 11)
    elif kind == "noise":
        host = rng.integers(0, 1 << 63, size=shape, dtype=np.uint64)
    elif kind == "mixed":
        host = rng.integers(0, 1 << 16, size=shape, dtype=np.uint64) | np.uint64(0x0000123400000000)
    elif kind == "zeros":
        host = np.zeros(shape, np.uint64)
    else:
        host = _labels(rng, shape, 200)
        host.reshape(-1)[-5000:] = 77
    vol = torch.from_numpy(host.view(np.int64)).cuda()
    codec = _lib.Codec(_lib.CODEC_BLOSC, 5, _lib.BLOSC_LZ4, 1, 8, 0)
    real = C.CDLL(REAL_BLOSC) if os.path.exists(REAL_BLOSC) else None
    total = 0
    for g, frame in _encode(vol, chunk):
        want = _expected(host, g, chunk)
        got = codecs.decode(codec, frame, want.nbytes)
        assert got.size == want.nbytes and np.array_equal(got.view(np.uint64).reshape(chunk), want), (kind, g)
        if real is not None:     # c-blosc 1.21 reads the same frame
            out = np.empty(want.nbytes, np.uint8)
            n = real.blosc_decompress(frame, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size))
            assert n == want.nbytes and np.array_equal(out.view(np.uint64).reshape(chunk), want), (kind, g, n)
        total += len(frame)
    raw = int(np.prod(chunk)) * 8 * len(_encode(vol, chunk))
    if kind in ("labels", "zeros", "runs_at_the_end"):
        assert total < raw / 2.5, (total, raw)          # label volumes shrink: runs of 11 bytes cost 4 bytes each in the planes that
                                                        # change with the id and next to nothing in the constant ones (measured 3.2 : 1;
                                                        # real fragments, ~20 voxels wide and equal row after row: 15-25 : 1)
    if kind == "noise":
        assert total <= raw + 16 * (raw // (int(np.prod(chunk)) * 8))   # stored frames: 16 bytes of header each


def test_device_frames_of_a_strided_view_and_through_the_layer_writer(tmp_path):
    """a view with row and section strides (a slab's interior) encodes like its contiguous copy; `_LayerWriter` writes a dataset
    with the device frames that reads back exactly, partial chunks at the array's end included"""
    from bootstrapper_amd.post.watershed import _LayerWriter
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    rng = np.random.default_rng(4)
    host = _labels(rng, (24, 140, 128), 9)
    big = torch.zeros((30, 150, 128), dtype=torch.int64, device="cuda")
    big[3:27, 5:145] = torch.from_numpy(host.view(np.int64)).cuda()
    view = big[3:27, 5:145]
    a = _encode(big, (8, 64, 64), view=view)
    b = _encode(torch.from_numpy(host.view(np.int64)).cuda(), (8, 64, 64))
    assert [f for _, f in a] == [f for _, f in b]
    ds = prepare_ds(str(tmp_path / "v.zarr") + "/labels", host.shape, chunk_shape=(8, 64, 64), dtype=np.uint64, voxel_size=(1, 1, 1), offset=(0, 0, 0))
    w = _LayerWriter(torch.device("cuda", 0), 8)
    assert w._device_frames_ok(ds, 0, 0, view[:8])
    w.submit(ds, view, 0, 0)
    w.drain()
    w.close()
    assert np.array_equal(open_ds(ds.path)[:], host)
    assert w.devbuf                                          # the device path ran
