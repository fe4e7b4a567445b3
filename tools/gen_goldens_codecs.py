"""Golden chunk-codec vectors for tests/test_codecs.py -> tests/golden/codec_cases.npz.

Frames are produced by third-party implementations that exist in this container only:
  * Blosc-1 frames by c-blosc 1.21.0 (/opt/conda/lib/libblosc.so.1, the library numcodecs wraps),
    through its C API blosc_compress_ctx;
  * numcodecs-style `lz4` (u32 size + LZ4 block) and `zstd` chunks by liblz4 / libzstd through
    pyarrow's codecs.
Each case stores the plain bytes and the encoded frame.  Run:  python tools/gen_goldens_codecs.py
"""
import ctypes as C
import os
import sys

import numpy as np
import pyarrow as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOSC = "/opt/conda/lib/libblosc.so.1"


def blosc_lib():
    b = C.CDLL(BLOSC)
    b.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_char_p, C.c_size_t, C.c_int]
    b.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return b


def blosc_compress(b, data, typesize, cname, clevel, shuffle, blocksize):
    dst = C.create_string_buffer(len(data) + 16)
    n = b.blosc_compress_ctx(clevel, shuffle, typesize, len(data), data, dst, len(dst), cname.encode(), blocksize, 1)
    assert n > 0, (cname, n)
    return dst.raw[:n]


def payloads(rng):
    """name -> (bytes, typesize): the kinds of chunk the path reads and writes."""
    out = {}
    raw = (np.clip(rng.normal(128, 30, (8, 40, 40)), 0, 255)).astype(np.uint8)
    raw[:, 10:20] = 7
    out["raw_u8"] = (raw.tobytes(), 1)
    labels = np.zeros((6, 33, 31), np.uint64)
    for i in range(40):
        z, y, x = rng.integers(0, 6), rng.integers(0, 28), rng.integers(0, 26)
        labels[z:, y:y + 6, x:x + 6] = rng.integers(1, 2 ** 40)
    out["labels_u64"] = (labels.tobytes(), 8)
    affs = rng.random((3, 5, 24, 24)).astype(np.float32)
    affs[:, :, :8] = 0.5
    out["affs_f32"] = (affs.tobytes(), 4)
    out["ramp_u16"] = ((np.arange(20011) // 3).astype(np.uint16).tobytes(), 2)   # not a multiple of 8 elements
    out["noise_u8"] = (rng.integers(0, 256, 5003, dtype=np.uint8).tobytes(), 1)   # incompressible -> stored
    out["tiny_u32"] = (np.arange(9, dtype=np.uint32).tobytes(), 4)                # below blosc's minimum buffer
    out["odd_u64"] = (labels.tobytes()[:8 * 1001 + 3], 8)                         # bytes not a multiple of typesize
    out["zeros_u8"] = (bytes(70000), 1)
    return out


def main():
    rng = np.random.default_rng(20240607)
    b = blosc_lib()
    cases = {}
    names = []
    data = payloads(rng)
    for pname, (plain, ts) in data.items():
        cases[f"plain/{pname}"] = np.frombuffer(plain, np.uint8)
    for pname, (plain, ts) in data.items():
        for cname in ("lz4", "lz4hc", "zstd", "zlib", "blosclz"):
            for shuffle in (0, 1, 2):
                for blocksize in (0, 4096):
                    if blocksize and pname in ("tiny_u32", "noise_u8"):
                        continue
                    clevel = 5 if cname != "zstd" else 3
                    frame = blosc_compress(b, plain, ts, cname, clevel, shuffle, blocksize)
                    # the library itself must take the frame back
                    back = C.create_string_buffer(len(plain) + 1)
                    assert b.blosc_decompress_ctx(frame, back, len(plain), 1) == len(plain)
                    assert back.raw[:len(plain)] == plain
                    key = f"blosc/{pname}/{cname}/s{shuffle}/b{blocksize}"
                    cases[key] = np.frombuffer(frame, np.uint8)
                    names.append(key)
    for pname, (plain, ts) in data.items():
        z = pa.compress(plain, codec="zstd", asbytes=True)
        cases[f"zstd/{pname}"] = np.frombuffer(z, np.uint8)
        l4 = len(plain).to_bytes(4, "little") + pa.compress(plain, codec="lz4_raw", asbytes=True)
        cases[f"lz4/{pname}"] = np.frombuffer(l4, np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "codec_cases.npz")
    np.savez_compressed(out, **{k.replace("/", "__"): v for k, v in cases.items()})
    print(f"{len(cases)} arrays, {os.path.getsize(out) / 1e6:.2f} MB -> {out}")


if __name__ == "__main__":
    sys.exit(main())
