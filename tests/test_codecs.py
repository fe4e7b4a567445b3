"""Volume I/O codecs (include/bsmi_io.h) on the CPU: golden frames written by
c-blosc 1.21.0 / liblz4 / libzstd (tools/gen_goldens_codecs.py), round trips, frames checked by
the third-party decoders where this machine has them, and malformed input."""
import ctypes as C
import os

import numpy as np
import pytest

from bootstrapper_amd import _lib, codecs
from bootstrapper_amd._lib import BsmiError

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "codec_cases.npz")
REAL_BLOSC = "/opt/conda/lib/libblosc.so.1"


def _goldens():
    g = np.load(GOLD)
    plain = {k[len("plain__"):]: g[k] for k in g.files if k.startswith("plain__")}
    frames = {k: g[k] for k in g.files if not k.startswith("plain__")}
    return plain, frames


def _codec_for(kind):
    return {"blosc": _lib.Codec(_lib.CODEC_BLOSC, 5, _lib.BLOSC_LZ4, 1, 1, 0),
            "zstd": _lib.Codec(_lib.CODEC_ZSTD, 1, 0, 0, 1, 0),
            "lz4": _lib.Codec(_lib.CODEC_LZ4, 1, 0, 0, 1, 0)}[kind]


def test_decode_matches_third_party_frames():
    plain, frames = _goldens()
    assert len(frames) > 200
    kinds = set()
    for key, frame in frames.items():
        kind, pname = key.split("__")[:2]
        kinds.add(key.split("__")[2] if kind == "blosc" else kind)
        out = codecs.decode(_codec_for(kind), frame.tobytes(), plain[pname].size + 16)
        assert out.size == plain[pname].size and np.array_equal(out, plain[pname]), key
    assert kinds == {"lz4", "lz4hc", "zstd", "zlib", "blosclz"}


def _payloads():
    rng = np.random.default_rng(5)
    lab = np.repeat(rng.integers(1, 2 ** 50, 300, dtype=np.uint64), rng.integers(1, 400, 300))
    return {
        "empty": (np.zeros(0, np.uint8), 1),
        "one": (np.array([7], np.uint8), 1),
        "twelve": (np.arange(12, dtype=np.uint8), 1),
        "labels": (lab, 8),
        "raw": (np.clip(rng.normal(120, 20, 300001), 0, 255).astype(np.uint8), 1),
        "f32": (np.round(rng.random(40013), 2).astype(np.float32), 4),
        "noise": (rng.integers(0, 256, 70001, dtype=np.uint8), 1),
        "zeros": (np.zeros(1 << 20, np.uint16), 2),
    }


ALL_CODECS = [
    {"id": None},
    {"id": "zlib", "level": 3}, {"id": "gzip", "level": 1}, {"id": "zstd", "level": 3}, {"id": "lz4"},
    {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0},
    {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 2, "blocksize": 8192},
    {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": -1, "blocksize": 0},
    {"id": "blosc", "cname": "zlib", "clevel": 9, "shuffle": 0, "blocksize": 1000},
    {"id": "blosc", "cname": "lz4hc", "clevel": 0, "shuffle": 1, "blocksize": 0},
]


@pytest.mark.parametrize("conf", ALL_CODECS, ids=lambda c: "-".join(str(v) for v in c.values()))
def test_round_trip(conf):
    for name, (arr, ts) in _payloads().items():
        cd = codecs.from_config(None if conf["id"] is None else conf, ts)
        enc = codecs.encode(cd, arr)
        out = codecs.decode(cd, enc, arr.nbytes + 5)
        assert out.tobytes() == arr.tobytes(), (conf, name)
        if conf["id"] in ("zstd", "lz4") or (conf["id"] == "blosc" and conf["clevel"]):
            if name in ("labels", "zeros"):
                assert len(enc) < arr.nbytes, (conf, name, len(enc))  # it does compress


def test_frames_are_accepted_by_the_third_party_decoders():
    pa = pytest.importorskip("pyarrow")
    import gzip
    import zlib
    for name, (arr, ts) in _payloads().items():
        plain = arr.tobytes()
        enc = codecs.encode(codecs.from_config({"id": "lz4"}, ts), arr)
        assert int.from_bytes(enc[:4], "little") == len(plain)
        if plain:
            assert pa.decompress(enc[4:], decompressed_size=len(plain), codec="lz4_raw", asbytes=True) == plain
        enc = codecs.encode(codecs.from_config({"id": "zstd", "level": 5}, ts), arr)
        assert pa.decompress(enc, decompressed_size=len(plain), codec="zstd", asbytes=True) == plain
        assert zlib.decompress(codecs.encode(codecs.from_config({"id": "zlib", "level": 2}, ts), arr)) == plain
        assert gzip.decompress(codecs.encode(codecs.from_config({"id": "gzip", "level": 2}, ts), arr)) == plain


@pytest.mark.skipif(not os.path.exists(REAL_BLOSC), reason="c-blosc is not installed on this machine")
def test_blosc_frames_are_accepted_by_c_blosc():
    b = C.CDLL(REAL_BLOSC)
    b.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for conf in ALL_CODECS:
        if conf["id"] != "blosc":
            continue
        for name, (arr, ts) in _payloads().items():
            plain = arr.tobytes()
            enc = codecs.encode(codecs.from_config(conf, ts), arr)
            back = C.create_string_buffer(len(plain) + 1)
            assert b.blosc_decompress_ctx(enc, back, len(plain), 1) == len(plain), (conf, name)
            assert back.raw[:len(plain)] == plain, (conf, name)


def test_malformed_frames_fail_cleanly():
    plain, frames = _goldens()
    rng = np.random.default_rng(11)
    keys = sorted(frames)
    n_err = 0
    for key in keys[::3]:
        kind, pname = key.split("__")[:2]
        frame = frames[key].copy()
        want = plain[pname]
        cd = _codec_for(kind)
        for trial in range(6):
            bad = frame.copy()
            if trial % 2 == 0 and bad.size > 20:
                bad = bad[: rng.integers(1, bad.size - 1)]
            else:
                for pos in rng.integers(0, bad.size, 3):
                    bad[pos] ^= 1 << rng.integers(0, 8)
            try:
                out = codecs.decode(cd, bad.tobytes(), want.size + 16)
                assert out.size <= want.size + 16
            except BsmiError:
                n_err += 1
    assert n_err > 50
    with pytest.raises(BsmiError, match="room for"):
        codecs.decode(_codec_for("blosc"), frames[keys[0]].tobytes(), 3)
    with pytest.raises(NotImplementedError):
        codecs.from_config({"id": "bz2"}, 1)
    with pytest.raises(NotImplementedError):
        codecs.from_config({"id": "blosc", "cname": "snappy"}, 1)


def test_threaded_chunk_files(tmp_path):
    rng = np.random.default_rng(3)
    cd = codecs.from_config(codecs.DEFAULT_COMPRESSOR, 8)
    arrays = [np.repeat(rng.integers(0, 2 ** 40, 64, dtype=np.uint64), 512).reshape(32, 32, 32) for _ in range(13)]
    paths = [str(tmp_path / f"0.0.{i}") for i in range(13)]
    codecs.write_chunks(cd, paths, arrays, threads=4)
    assert all(os.path.getsize(p) < a.nbytes for p, a in zip(paths, arrays))
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    got = codecs.read_chunks(cd, paths + [str(tmp_path / "missing")], arrays[0].nbytes, threads=4)
    assert got[-1] is None
    for a, g in zip(arrays, got):
        assert g.tobytes() == a.tobytes()
    with open(paths[2], "r+b") as f:
        f.seek(40)
        f.write(b"\xff" * 64)
    with pytest.raises(BsmiError, match="0.0.2"):
        codecs.read_chunks(cd, paths, arrays[0].nbytes, threads=3)
    with pytest.raises(BsmiError, match="No such file"):
        codecs.write_chunks(cd, [str(tmp_path / "nodir" / "x")], arrays[:1])


@pytest.mark.parametrize("compressor", ["default", None, "zstd", {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 4096}])
def test_region_reads_and_writes_against_numpy(tmp_path, compressor):
    """bsmi_chunks_read_into / bsmi_chunks_write_from (zarr_io.ZarrArray.read_into / write_from, and through them __getitem__ /
    __setitem__): arbitrary boxes over 2-, 3- and 4-axis arrays -- chunks covered partly (read-modify-write), boxes that end inside
    a chunk, missing chunks (fill value), a leading-axis prefix (three of six channels: a Blosc chunk is decoded only so far),
    strided destinations and sources -- against a numpy mirror of the array."""
    from bootstrapper_amd.zarr_io import prepare_ds, open_ds
    rng = np.random.default_rng(3)
    for case, (shape, chunks, dtype) in enumerate([((37, 53), (16, 20), np.uint8), ((20, 33, 41), (8, 16, 16), np.uint64),
                                                    ((6, 12, 30, 28), (6, 8, 16, 16), np.uint8), ((5, 9, 17, 21), (2, 4, 8, 8), np.float32)]):
        path = str(tmp_path / f"v{case}.zarr") + "/a"
        ds = prepare_ds(path, shape, chunk_shape=chunks, dtype=dtype, compressor=compressor,
                        voxel_size=(1,) * min(3, len(shape)), offset=(0,) * min(3, len(shape)))
        mirror = np.zeros(shape, dtype)
        for _ in range(12):
            lo = [int(rng.integers(0, s)) for s in shape]
            hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
            key = tuple(slice(a, b) for a, b in zip(lo, hi))
            ext = [b - a for a, b in zip(lo, hi)]
            if dtype == np.float32:
                val = rng.random(ext).astype(dtype)
            else:
                val = (rng.integers(0, 5, ext) * rng.integers(0, 2, ext)).astype(dtype)   # compressible, with runs
            if _ % 3 == 0:   # a strided source: every other row of a larger array
                big = np.zeros([2 * e for e in ext[:-1]] + [ext[-1] + 3], dtype)
                view = big[tuple(slice(0, 2 * e, 2) for e in ext[:-1]) + (slice(1, 1 + ext[-1]),)]
                view[...] = val
                ds.write_from(key, view)
            else:
                ds[key] = val
            mirror[key] = val
            ds2 = open_ds(path)
            lo = [int(rng.integers(0, s)) for s in shape]
            hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
            key = tuple(slice(a, b) for a, b in zip(lo, hi))
            assert np.array_equal(ds2[key], mirror[key]), (case, key)
            assert np.array_equal(ds2._getitem_python(ds2._norm(key)), mirror[key])
            # into a strided destination
            ext = [b - a for a, b in zip(lo, hi)]
            big = np.full([e + 2 for e in ext], 7, dtype)
            dst = big[tuple(slice(1, 1 + e) for e in ext)]
            ds2.read_into(key, dst)
            assert np.array_equal(dst, mirror[key]) and big.flat[0] == 7 and big.flat[-1] == 7
        assert np.array_equal(open_ds(path)[:], mirror)
        if len(shape) == 4:   # the first channels only
            assert np.array_equal(open_ds(path)[:3], mirror[:3])
    # errors: a destination of the wrong shape, a read-only array
    ds = open_ds(path)
    with pytest.raises(ValueError):
        ds.read_into((slice(0, 2),), np.zeros((3, 9, 17, 21), np.float32))
    with pytest.raises(PermissionError):
        ds[:1] = 0
