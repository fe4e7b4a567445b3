// Zarr v2 chunk codecs and threaded chunk-file I/O: the host side of the volume path that sits
// before `bs predict` and after `bs segment`.
//
// The reference reads and writes its volumes through zarr-python + numcodecs
// (/root/reference/bootstrapper/predict.py:169-178 prepare_ds, post/watershed.py:319-330,
// data/volumes.py:14-19); zarr-python's default compressor is Blosc(lz4, clevel 5, byte shuffle).
// Neither package exists on the GPU box, so the container formats are implemented here:
//   * the Blosc-1 frame (c-blosc 1.21: 16-byte header, block starts, per-block splits, byte- and
//     bit-shuffle) around lz4 / lz4hc / zstd / zlib / blosclz streams,
//   * numcodecs' `lz4` (u32 size + one LZ4 block), `zstd` (one frame), `zlib`, `gzip`.
// LZ4 blocks and blosclz streams are decoded (and LZ4 encoded) by the code below; zstd frames go
// through the system's libzstd.so.1 (dlopen: the image has the library but not its header), zlib
// and gzip through libz.  Chunks are independent, so a batch of chunk files is decoded / encoded
// by a small thread pool without the Python GIL.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <mutex>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsmi_io.h"

namespace bsmi {
void set_error(const char* fmt, ...);
}

// records the message for the calling thread (bsmi_last_error) and returns BSMI_ERR_INVALID
#define BSMI_IO_FAIL(...)          \
  do {                             \
    bsmi::set_error(__VA_ARGS__);  \
    return BSMI_ERR_INVALID;       \
  } while (0)

namespace {

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void put32(uint8_t* p, uint32_t v) {
  p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24;
}
inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

// ---------------------------------------------------------------------------------------------
// zstd via the system library
struct ZstdApi {
  bool ok = false;
  size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
  size_t (*bound)(size_t) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  unsigned long long (*content_size)(const void*, size_t) = nullptr;
};

const ZstdApi* zstd_api() {
  static ZstdApi api = [] {
    ZstdApi a;
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return a;
    a.compress = (decltype(a.compress))dlsym(h, "ZSTD_compress");
    a.decompress = (decltype(a.decompress))dlsym(h, "ZSTD_decompress");
    a.bound = (decltype(a.bound))dlsym(h, "ZSTD_compressBound");
    a.is_error = (decltype(a.is_error))dlsym(h, "ZSTD_isError");
    a.content_size = (decltype(a.content_size))dlsym(h, "ZSTD_getFrameContentSize");
    a.ok = a.compress && a.decompress && a.bound && a.is_error && a.content_size;
    return a;
  }();
  return api.ok ? &api : nullptr;
}

// dst[0, len) = the bytes `dist` behind dst, which may overlap it (a repeating pattern of period dist)
inline void copy_match(uint8_t* o, const uint8_t* m, size_t dist, size_t len) {
  if (dist >= len) {
    memcpy(o, m, len);
  } else if (dist == 1) {
    memset(o, m[0], len);
  } else if (dist >= 8) {
    size_t i = 0;
    for (; i + 8 <= len; i += 8) memcpy(o + i, m + i, 8);  // each 8-byte piece lies wholly behind its target
    for (; i < len; ++i) o[i] = m[i];
  } else {
    for (size_t i = 0; i < len; ++i) o[i] = m[i];
  }
}

// ---------------------------------------------------------------------------------------------
// LZ4 block format (lz4 Block Format Description 1.6): token = literal run << 4 | match length - 4,
// run lengths >= 15 continue in 255-steps, little-endian 16-bit offsets, last sequence literals only.
// Returns the number of bytes produced, or -1 for a malformed stream.
int64_t lz4_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  size_t ip = 0, op = 0;
  while (ip < n) {
    unsigned token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned s;
      do {
        if (ip >= n) return -1;
        s = src[ip++];
        lit += s;
      } while (s == 255);
    }
    if (lit > n - ip || lit > cap - op) return -1;
    memcpy(dst + op, src + ip, lit);
    ip += lit;
    op += lit;
    if (ip >= n) break;
    if (n - ip < 2) return -1;
    size_t off = src[ip] | (src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return -1;
    size_t ml = token & 15;
    if (ml == 15) {
      unsigned s;
      do {
        if (ip >= n) return -1;
        s = src[ip++];
        ml += s;
      } while (s == 255);
    }
    ml += 4;
    if (ml > cap - op) return -1;
    const uint8_t* m = dst + op - off;
    uint8_t* o = dst + op;
    copy_match(o, m, off, ml);
    op += ml;
  }
  return (int64_t)op;
}

// Greedy single-probe LZ4 block encoder (a valid stream for any LZ4 decoder; not byte-identical to
// liblz4's output, which the format does not require).  Returns 0 when `cap` is too small.
size_t lz4_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  constexpr int kHashBits = 16;
  static thread_local std::vector<uint32_t> table;
  table.assign(size_t(1) << kHashBits, 0u);
  size_t ip = 0, anchor = 0, op = 0;
  auto emit = [&](size_t lit, size_t off, size_t ml) -> bool {  // ml = 0: last literals
    size_t need = 1 + lit / 255 + 1 + lit + (ml ? 2 + ml / 255 + 1 : 0);
    if (need > cap - op) return false;
    uint8_t* tok = dst + op++;
    size_t l = lit;
    if (l >= 15) {
      *tok = 15 << 4;
      for (l -= 15; l >= 255; l -= 255) dst[op++] = 255;
      dst[op++] = (uint8_t)l;
    } else {
      *tok = (uint8_t)(l << 4);
    }
    memcpy(dst + op, src + anchor, lit);
    op += lit;
    if (ml) {
      dst[op++] = off & 255;
      dst[op++] = (uint8_t)(off >> 8);
      size_t m = ml - 4;
      if (m >= 15) {
        *tok |= 15;
        for (m -= 15; m >= 255; m -= 255) dst[op++] = 255;
        dst[op++] = (uint8_t)m;
      } else {
        *tok |= (uint8_t)m;
      }
    }
    return true;
  };
  if (n >= 13 && n < (size_t(1) << 32) - 1) {
    const size_t last_start = n - 12, match_end = n - 5;
    size_t misses = 0;
    while (ip <= last_start) {
      uint32_t seq = load32(src + ip);
      uint32_t h = (seq * 2654435761u) >> (32 - kHashBits);
      uint32_t cand = table[h];
      table[h] = (uint32_t)(ip + 1);
      if (cand && ip - (cand - 1) <= 65535 && load32(src + cand - 1) == seq) {
        size_t ref = cand - 1, ml = 4;
        // eight bytes per step (label volumes are long runs: the byte loop was most of the encoder's time on them)
        while (ip + ml + 8 <= match_end) {
          uint64_t a, b;
          memcpy(&a, src + ref + ml, 8);
          memcpy(&b, src + ip + ml, 8);
          if (a != b) { ml += (size_t)(__builtin_ctzll(a ^ b) >> 3); goto extended; }
          ml += 8;
        }
        while (ip + ml < match_end && src[ref + ml] == src[ip + ml]) ++ml;
      extended:
        while (ip > anchor && ref > 0 && src[ip - 1] == src[ref - 1]) {  // extend backwards
          --ip; --ref; ++ml;
        }
        if (!emit(ip - anchor, ip - ref, ml)) return 0;
        ip += ml;
        anchor = ip;
        misses = 0;
        if (ip >= 2 && ip - 2 + 4 <= n) {
          uint32_t s2 = load32(src + ip - 2);
          table[(s2 * 2654435761u) >> (32 - kHashBits)] = (uint32_t)(ip - 2 + 1);
        }
      } else {
        ip += 1 + (misses++ >> 6);
      }
    }
  }
  if (!emit(n - anchor, 0, 0)) return 0;
  return op;
}

// ---------------------------------------------------------------------------------------------
// BloscLZ stream (c-blosc 1.21 blosclz.c, a FastLZ descendant): control byte < 32 = literal run of
// ctrl + 1 bytes; otherwise a match of length (ctrl >> 5) + 2 (7 = extended by 255-steps) at
// distance ((ctrl & 31) << 8 | next byte) + 1, with 0x1fff followed by a 16-bit far distance.
int64_t blosclz_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  if (n == 0) return 0;
  size_t ip = 0, op = 0;
  unsigned ctrl = src[ip++] & 31u;
  for (;;) {
    if (ctrl >= 32) {
      size_t len = (ctrl >> 5) - 1;
      size_t ofs = (ctrl & 31u) << 8;
      unsigned code;
      if (len == 6) {
        do {
          if (ip >= n) return -1;
          code = src[ip++];
          len += code;
        } while (code == 255);
      }
      if (ip >= n) return -1;
      code = src[ip++];
      len += 3;
      size_t dist = ofs + code;
      if (code == 255 && ofs == (31u << 8)) {
        if (n - ip < 2) return -1;
        dist = ((size_t)src[ip] << 8) + src[ip + 1] + 8191;
        ip += 2;
      }
      dist += 1;
      if (dist > op || len > cap - op) return -1;
      const uint8_t* m = dst + op - dist;
      uint8_t* o = dst + op;
      copy_match(o, m, dist, len);
      op += len;
    } else {
      size_t run = ctrl + 1;
      if (run > n - ip || run > cap - op) return -1;
      memcpy(dst + op, src + ip, run);
      ip += run;
      op += run;
    }
    if (ip >= n) break;
    ctrl = src[ip++];
  }
  return (int64_t)op;
}

// ---------------------------------------------------------------------------------------------
// zlib / gzip
int64_t zlib_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, bool gz) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, gz ? 16 + MAX_WBITS : MAX_WBITS) != Z_OK) return -1;
  int64_t total = 0;
  size_t in_pos = 0;
  int rc = Z_OK;
  // avail_in / avail_out are 32-bit: feed in pieces
  while (rc != Z_STREAM_END) {
    if (zs.avail_in == 0 && in_pos < n) {
      size_t k = n - in_pos < (1u << 30) ? n - in_pos : (1u << 30);
      zs.next_in = const_cast<Bytef*>(src + in_pos);
      zs.avail_in = (uInt)k;
      in_pos += k;
    }
    size_t room = cap - (size_t)total;
    size_t k = room < (1u << 30) ? room : (1u << 30);
    zs.next_out = dst + total;
    zs.avail_out = (uInt)k;
    rc = inflate(&zs, Z_NO_FLUSH);
    total += (int64_t)(k - zs.avail_out);
    if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return -1; }
    if (rc == Z_OK && zs.avail_out != 0 && zs.avail_in == 0 && in_pos >= n) { inflateEnd(&zs); return -1; }  // truncated
    if (rc == Z_OK && (size_t)total == cap && zs.avail_out == 0) {
      // output full: only fine if the stream ends right here
      uint8_t extra;
      zs.next_out = &extra;
      zs.avail_out = 1;
      rc = inflate(&zs, Z_NO_FLUSH);
      if (rc != Z_STREAM_END || zs.avail_out != 1) { inflateEnd(&zs); return -1; }
    }
  }
  inflateEnd(&zs);
  return total;
}

int64_t zlib_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int level, bool gz) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, gz ? 16 + MAX_WBITS : MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
  size_t in_pos = 0, total = 0;
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    if (zs.avail_in == 0 && in_pos < n) {
      size_t k = n - in_pos < (1u << 30) ? n - in_pos : (1u << 30);
      zs.next_in = const_cast<Bytef*>(src + in_pos);
      zs.avail_in = (uInt)k;
      in_pos += k;
    }
    size_t room = cap - total;
    if (room == 0) { deflateEnd(&zs); return -1; }
    size_t k = room < (1u << 30) ? room : (1u << 30);
    zs.next_out = dst + total;
    zs.avail_out = (uInt)k;
    rc = deflate(&zs, in_pos >= n ? Z_FINISH : Z_NO_FLUSH);
    total += k - zs.avail_out;
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { deflateEnd(&zs); return -1; }
  }
  deflateEnd(&zs);
  return (int64_t)total;
}

// ---------------------------------------------------------------------------------------------
// Blosc shuffles over one block of `bsize` bytes with elements of `ts` bytes.
// (element tiles of 256 keep both the gathered and the scattered side inside L1)
void byte_shuffle(const uint8_t* src, uint8_t* dst, size_t bsize, size_t ts) {
  size_t ne = bsize / ts;
  constexpr size_t kTile = 256;
  for (size_t i0 = 0; i0 < ne; i0 += kTile) {
    size_t i1 = i0 + kTile < ne ? i0 + kTile : ne;
    for (size_t j = 0; j < ts; ++j) {
      uint8_t* row = dst + j * ne;
      const uint8_t* s = src + j;
      for (size_t i = i0; i < i1; ++i) row[i] = s[i * ts];
    }
  }
  memcpy(dst + ne * ts, src + ne * ts, bsize - ne * ts);
}
void byte_unshuffle(const uint8_t* src, uint8_t* dst, size_t bsize, size_t ts) {
  size_t ne = bsize / ts;
  constexpr size_t kTile = 256;
  for (size_t i0 = 0; i0 < ne; i0 += kTile) {
    size_t i1 = i0 + kTile < ne ? i0 + kTile : ne;
    for (size_t j = 0; j < ts; ++j) {
      const uint8_t* row = src + j * ne;
      uint8_t* d = dst + j;
      for (size_t i = i0; i < i1; ++i) d[i * ts] = row[i];
    }
  }
  memcpy(dst + ne * ts, src + ne * ts, bsize - ne * ts);
}
// 8x8 bit transpose of the bytes of x: afterwards byte k holds bit k of every input byte, input
// byte i at bit i.
inline uint64_t transpose8(uint64_t x) {
  uint64_t t;
  t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
  t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
  t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
  return x;
}
// Bit shuffle: the block is a matrix of ne elements x (8 ts) bits; the shuffled block holds its
// transpose, bit-row (byte j, bit k) at row 8 j + k, element i of a row at byte i / 8, bit i % 8.
// c-blosc 1.x (shuffle.c blosc_internal_bitshuffle) does this only when ne is a multiple of 8 and
// copies the block unchanged otherwise; trailing bytes beyond ne * ts are copied.
void bit_shuffle(const uint8_t* src, uint8_t* dst, size_t bsize, size_t ts) {
  size_t ne = bsize / ts;
  if (ne % 8) { memcpy(dst, src, bsize); return; }
  size_t nb = ne / 8;
  for (size_t j = 0; j < ts; ++j)
    for (size_t m = 0; m < nb; ++m) {
      uint64_t x = 0;
      for (int i = 0; i < 8; ++i) x |= (uint64_t)src[(8 * m + i) * ts + j] << (8 * i);
      x = transpose8(x);
      for (int k = 0; k < 8; ++k) dst[(8 * j + k) * nb + m] = (uint8_t)(x >> (8 * k));
    }
  memcpy(dst + ne * ts, src + ne * ts, bsize - ne * ts);
}
void bit_unshuffle(const uint8_t* src, uint8_t* dst, size_t bsize, size_t ts) {
  size_t ne = bsize / ts;
  if (ne % 8) { memcpy(dst, src, bsize); return; }
  size_t nb = ne / 8;
  for (size_t j = 0; j < ts; ++j)
    for (size_t m = 0; m < nb; ++m) {
      uint64_t x = 0;
      for (int k = 0; k < 8; ++k) x |= (uint64_t)src[(8 * j + k) * nb + m] << (8 * k);
      x = transpose8(x);  // the transpose is an involution
      for (int i = 0; i < 8; ++i) dst[(8 * m + i) * ts + j] = (uint8_t)(x >> (8 * i));
    }
  memcpy(dst + ne * ts, src + ne * ts, bsize - ne * ts);
}

// ---------------------------------------------------------------------------------------------
// Blosc-1 frame.  Header: [0] format version, [1] codec format version, [2] flags, [3] typesize,
// [4:8] nbytes, [8:12] blocksize, [12:16] cbytes (little endian); flags: 1 byte shuffle, 2 stored
// ("memcpyed"), 4 bit shuffle, 16 blocks are not split, bits 5-7 the inner codec.
enum { kBloscLz = 0, kLz4 = 1, kSnappy = 2, kZlib = 3, kZstd = 4 };
constexpr int kMaxSplits = 16, kMinBuffer = 128;

int64_t inner_decode(int fmt, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  switch (fmt) {
    case kBloscLz: return blosclz_decode(src, n, dst, cap);
    case kLz4: return lz4_decode(src, n, dst, cap);
    case kZlib: return zlib_decode(src, n, dst, cap, false);
    case kZstd: {
      const ZstdApi* z = zstd_api();
      if (!z) return -2;
      size_t r = z->decompress(dst, cap, src, n);
      return z->is_error(r) ? -1 : (int64_t)r;
    }
    default: return -3;
  }
}

// need: only the first `need` bytes of the chunk are wanted (blocks are independent: the others are not decoded)
int blosc_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len, size_t need = ~(size_t)0) {
  if (n < 16) BSMI_IO_FAIL("blosc: buffer shorter than the 16-byte header");
  unsigned flags = src[2];
  size_t ts = src[3] ? src[3] : 1;
  size_t nbytes = le32(src + 4), blocksize = le32(src + 8), cbytes = le32(src + 12);
  if (src[0] != 2) BSMI_IO_FAIL("blosc: unsupported frame format version %d", src[0]);
  if (cbytes > n) BSMI_IO_FAIL("blosc: frame says %zu compressed bytes, buffer has %zu", cbytes, n);
  if (nbytes > cap) BSMI_IO_FAIL("blosc: frame holds %zu bytes, destination has room for %zu", nbytes, cap);
  *out_len = nbytes;
  if (nbytes == 0) return BSMI_OK;
  if (flags & 2) {
    if (16 + nbytes > n) BSMI_IO_FAIL("blosc: stored frame is truncated");
    memcpy(dst, src + 16, nbytes < need ? nbytes : need);
    return BSMI_OK;
  }
  if (blocksize == 0 || blocksize > nbytes) BSMI_IO_FAIL("blosc: block size %zu does not fit a frame of %zu bytes", blocksize, nbytes);
  size_t nblocks = (nbytes + blocksize - 1) / blocksize, leftover = nbytes % blocksize;
  if (16 + 4 * nblocks > n) BSMI_IO_FAIL("blosc: block-start table is truncated");
  int fmt = flags >> 5;
  bool bytesh = (flags & 1) && ts > 1, bitsh = !bytesh && (flags & 4) && blocksize >= ts;
  static thread_local std::vector<uint8_t> tmp;
  if (bytesh || bitsh) tmp.resize(blocksize);
  for (size_t b = 0; b < nblocks && b * blocksize < need; ++b) {
    bool last_short = (b == nblocks - 1) && leftover;
    size_t bsize = last_short ? leftover : blocksize;
    size_t nsplits = (!(flags & 16) && ts <= (size_t)kMaxSplits && blocksize / ts >= (size_t)kMinBuffer && !last_short) ? ts : 1;
    size_t neblock = bsize / nsplits;
    uint8_t* out = dst + b * blocksize;
    uint8_t* target = (bytesh || bitsh) ? tmp.data() : out;
    size_t ip = le32(src + 16 + 4 * b);
    for (size_t s = 0; s < nsplits; ++s) {
      if (ip + 4 > n) BSMI_IO_FAIL("blosc: block %zu runs past the end of the frame", b);
      size_t cb = le32(src + ip);
      ip += 4;
      if (cb > n - ip) BSMI_IO_FAIL("blosc: block %zu runs past the end of the frame", b);
      if (cb == neblock) {
        memcpy(target + s * neblock, src + ip, neblock);
      } else {
        int64_t r = inner_decode(fmt, src + ip, cb, target + s * neblock, neblock);
        if (r == -2) BSMI_IO_FAIL("blosc: libzstd.so.1 is not available for a zstd frame");
        if (r == -3) BSMI_IO_FAIL("blosc: inner codec %d is not supported", fmt);
        if (r != (int64_t)neblock) BSMI_IO_FAIL("blosc: block %zu is corrupt (inner codec %d)", b, fmt);
      }
      ip += cb;
    }
    if (bytesh) byte_unshuffle(target, out, bsize, ts);
    else if (bitsh) bit_unshuffle(target, out, bsize, ts);
  }
  return BSMI_OK;
}

int64_t inner_encode(int fmt, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int clevel) {
  switch (fmt) {
    case kLz4: return (int64_t)lz4_encode(src, n, dst, cap);
    case kZlib: {
      int64_t r = zlib_encode(src, n, dst, cap, clevel, false);
      return r < 0 ? 0 : r;
    }
    case kZstd: {
      const ZstdApi* z = zstd_api();
      if (!z) return -2;
      size_t r = z->compress(dst, cap, src, n, clevel < 9 ? 2 * clevel - 1 : 22);
      return z->is_error(r) ? 0 : (int64_t)r;
    }
    default: return -3;
  }
}

// cname: BSMI_BLOSC_LZ4 / ZLIB / ZSTD.  shuffle: 0 none, 1 byte, 2 bit.
int blosc_encode(const uint8_t* src, size_t n, int cname, int clevel, int shuffle, size_t ts, size_t blocksize,
                 uint8_t* dst, size_t cap, size_t* out_len) {
  if (n > 0x7fffffffu - 16) BSMI_IO_FAIL("blosc: %zu bytes do not fit one frame (2 GiB - 16 limit)", n);
  if (cap < n + 16) BSMI_IO_FAIL("blosc: destination needs %zu bytes", n + 16);
  if (ts == 0 || ts > 255) ts = 1;
  int fmt = cname == BSMI_BLOSC_LZ4 ? kLz4 : cname == BSMI_BLOSC_ZLIB ? kZlib : cname == BSMI_BLOSC_ZSTD ? kZstd : -1;
  if (fmt < 0) BSMI_IO_FAIL("blosc: cannot encode with inner codec id %d (lz4, zlib and zstd are available)", cname);
  if (clevel < 0 || clevel > 9) BSMI_IO_FAIL("blosc: clevel %d outside 0..9", clevel);
  unsigned flags = 16 | (fmt << 5);
  if (shuffle == 1 && ts > 1) flags |= 1;
  else if (shuffle == 2) flags |= 4;
  else if (shuffle == 1) shuffle = 0;
  if (blocksize == 0) blocksize = 256 * 1024;
  blocksize -= blocksize % ts;
  if (blocksize < ts) blocksize = ts;
  if (blocksize > n) blocksize = n ? n : 1;
  dst[0] = 2; dst[1] = 1; dst[2] = (uint8_t)flags; dst[3] = (uint8_t)ts;
  put32(dst + 4, (uint32_t)n);
  put32(dst + 8, (uint32_t)blocksize);
  size_t nblocks = n ? (n + blocksize - 1) / blocksize : 0;
  size_t op = 16 + 4 * nblocks;
  bool stored = clevel == 0 || op >= n + 16;
  static thread_local std::vector<uint8_t> tmp;
  if (!stored && shuffle) tmp.resize(blocksize);
  for (size_t b = 0; b < nblocks && !stored; ++b) {
    size_t bsize = (b == nblocks - 1 && n % blocksize) ? n % blocksize : blocksize;
    const uint8_t* in = src + b * blocksize;
    if (flags & 1) { byte_shuffle(in, tmp.data(), bsize, ts); in = tmp.data(); }
    else if ((flags & 4) && blocksize >= ts) { bit_shuffle(in, tmp.data(), bsize, ts); in = tmp.data(); }
    put32(dst + 16 + 4 * b, (uint32_t)op);
    size_t avail = cap - op;
    if (avail < 5) { stored = true; break; }
    size_t room = bsize - 1 < avail - 4 ? bsize - 1 : avail - 4;
    int64_t cb = room ? inner_encode(fmt, in, bsize, dst + op + 4, room, clevel) : 0;
    if (cb == -2) BSMI_IO_FAIL("blosc: libzstd.so.1 is not available");
    if (cb <= 0) {  // incompressible block: stored raw, marked by cbytes == block size
      if (avail < 4 + bsize) { stored = true; break; }
      memcpy(dst + op + 4, in, bsize);
      cb = (int64_t)bsize;
    }
    put32(dst + op, (uint32_t)cb);
    op += 4 + (size_t)cb;
  }
  if (!stored && op >= n + 16) stored = true;  // no gain over storing: one plain copy, like c-blosc
  if (stored) {
    dst[2] = (uint8_t)(flags | 2);
    memcpy(dst + 16, src, n);
    op = 16 + n;
  }
  put32(dst + 12, (uint32_t)op);
  *out_len = op;
  return BSMI_OK;
}

int decode_any(const bsmi_codec* c, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len, size_t need = ~(size_t)0) {
  switch (c->id) {
    case BSMI_CODEC_RAW:
      if (n > cap) BSMI_IO_FAIL("raw chunk of %zu bytes, destination has room for %zu", n, cap);
      memcpy(dst, src, n < need ? n : need);
      *out_len = n;
      return BSMI_OK;
    case BSMI_CODEC_ZLIB:
    case BSMI_CODEC_GZIP: {
      int64_t r = zlib_decode(src, n, dst, cap, c->id == BSMI_CODEC_GZIP);
      if (r < 0) BSMI_IO_FAIL("%s chunk is corrupt or larger than %zu bytes", c->id == BSMI_CODEC_GZIP ? "gzip" : "zlib", cap);
      *out_len = (size_t)r;
      return BSMI_OK;
    }
    case BSMI_CODEC_ZSTD: {
      const ZstdApi* z = zstd_api();
      if (!z) BSMI_IO_FAIL("zstd chunk but libzstd.so.1 is not available");
      size_t r = z->decompress(dst, cap, src, n);
      if (z->is_error(r)) BSMI_IO_FAIL("zstd chunk is corrupt or larger than %zu bytes", cap);
      *out_len = r;
      return BSMI_OK;
    }
    case BSMI_CODEC_LZ4: {
      if (n < 4) BSMI_IO_FAIL("lz4 chunk shorter than its size prefix");
      size_t want = le32(src);
      if (want > cap) BSMI_IO_FAIL("lz4 chunk holds %zu bytes, destination has room for %zu", want, cap);
      int64_t r = lz4_decode(src + 4, n - 4, dst, want);
      if (r != (int64_t)want) BSMI_IO_FAIL("lz4 chunk is corrupt");
      *out_len = want;
      return BSMI_OK;
    }
    case BSMI_CODEC_BLOSC:
      return blosc_decode(src, n, dst, cap, out_len, need);
    default:
      BSMI_IO_FAIL("unknown chunk codec id %d", c->id);
  }
}

size_t bound_any(const bsmi_codec* c, size_t n) {
  switch (c->id) {
    case BSMI_CODEC_RAW: return n;
    case BSMI_CODEC_ZLIB:
    case BSMI_CODEC_GZIP: return n + n / 1000 + 64;
    case BSMI_CODEC_ZSTD: { const ZstdApi* z = zstd_api(); return z ? z->bound(n) : n + n / 128 + 512; }
    case BSMI_CODEC_LZ4: return 4 + n + n / 255 + 16;
    case BSMI_CODEC_BLOSC: return n + 16;
    default: return 0;
  }
}

int encode_any(const bsmi_codec* c, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len) {
  switch (c->id) {
    case BSMI_CODEC_RAW:
      if (n > cap) BSMI_IO_FAIL("destination too small");
      memcpy(dst, src, n);
      *out_len = n;
      return BSMI_OK;
    case BSMI_CODEC_ZLIB:
    case BSMI_CODEC_GZIP: {
      int64_t r = zlib_encode(src, n, dst, cap, c->level, c->id == BSMI_CODEC_GZIP);
      if (r < 0) BSMI_IO_FAIL("zlib encode failed (level %d, destination %zu bytes)", c->level, cap);
      *out_len = (size_t)r;
      return BSMI_OK;
    }
    case BSMI_CODEC_ZSTD: {
      const ZstdApi* z = zstd_api();
      if (!z) BSMI_IO_FAIL("zstd requested but libzstd.so.1 is not available");
      size_t r = z->compress(dst, cap, src, n, c->level);
      if (z->is_error(r)) BSMI_IO_FAIL("zstd encode failed (destination %zu bytes)", cap);
      *out_len = r;
      return BSMI_OK;
    }
    case BSMI_CODEC_LZ4: {
      if (n > 0x7E000000u) BSMI_IO_FAIL("lz4 chunk too large");
      if (cap < 4) BSMI_IO_FAIL("destination too small");
      put32(dst, (uint32_t)n);
      size_t r = lz4_encode(src, n, dst + 4, cap - 4);
      if (r == 0) BSMI_IO_FAIL("lz4 encode: destination of %zu bytes is too small", cap);
      *out_len = 4 + r;
      return BSMI_OK;
    }
    case BSMI_CODEC_BLOSC:
      return blosc_encode(src, n, c->cname, c->level, c->shuffle, (size_t)c->typesize, (size_t)c->blocksize, dst, cap, out_len);
    default:
      BSMI_IO_FAIL("unknown chunk codec id %d", c->id);
  }
}

bool read_file(const char* path, std::vector<uint8_t>& buf, int* err) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) { *err = errno; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { *err = errno; close(fd); return false; }
  buf.resize((size_t)st.st_size);
  size_t got = 0;
  while (got < buf.size()) {
    ssize_t r = read(fd, buf.data() + got, buf.size() - got);
    if (r < 0) { if (errno == EINTR) continue; *err = errno; close(fd); return false; }
    if (r == 0) break;
    got += (size_t)r;
  }
  close(fd);
  buf.resize(got);
  return true;
}

bool write_file_atomic(const char* path, const uint8_t* data, size_t n, int* err) {
  std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid()) + "." +
                    std::to_string((unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id()) % 100000);
  int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) { *err = errno; return false; }
  size_t put = 0;
  while (put < n) {
    ssize_t r = write(fd, data + put, n - put);
    if (r < 0) { if (errno == EINTR) continue; *err = errno; close(fd); unlink(tmp.c_str()); return false; }
    put += (size_t)r;
  }
  close(fd);
  if (rename(tmp.c_str(), path) != 0) { *err = errno; unlink(tmp.c_str()); return false; }
  return true;
}

// Scratch buffers of the worker threads.  run_pool starts its threads per request, so a `thread_local` vector is a fresh
// allocation in every one of them: 16 MB of chunk image + 16 MB of encoder output per chunk, every page of it faulted in and
// zeroed before use (the 32 GB of fragments + segmentations of a 1024^3 volume went at 125 MB/s per thread that way).  The
// buffers are handed out from a process-wide free list instead and keep their pages.
class ScratchPool {
 public:
  std::vector<uint8_t> take(size_t n) {
    std::vector<uint8_t> v;
    {
      std::lock_guard<std::mutex> g(mu_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].capacity() >= n && (best == free_.size() || free_[i].capacity() < free_[best].capacity())) best = i;
      if (best == free_.size() && !free_.empty()) best = free_.size() - 1;
      if (best < free_.size()) { v = std::move(free_[best]); free_.erase(free_.begin() + (long)best); }
    }
    if (v.size() < n) v.resize(n);
    return v;
  }
  void give(std::vector<uint8_t>&& v) {
    std::lock_guard<std::mutex> g(mu_);
    if (free_.size() < 256) free_.push_back(std::move(v));
  }
 private:
  std::mutex mu_;
  std::vector<std::vector<uint8_t>> free_;
};
ScratchPool g_scratch;
struct Scratch {   // a buffer of at least n bytes for the life of the object
  std::vector<uint8_t> v;
  explicit Scratch(size_t n) : v(g_scratch.take(n)) {}
  ~Scratch() { g_scratch.give(std::move(v)); }
  uint8_t* data() { return v.data(); }
  size_t size() const { return v.size(); }
};

template <class F>
void run_pool(int n, int threads, F&& body) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  if (threads <= 1) {
    for (int i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (int i; (i = next.fetch_add(1)) < n;) body(i);
    });
  for (auto& t : pool) t.join();
}

}  // namespace

extern "C" {

size_t bsmi_codec_bound(const bsmi_codec* codec, size_t n) { return codec ? bound_any(codec, n) : 0; }

int bsmi_codec_decode(const bsmi_codec* codec, const void* src, size_t n, void* dst, size_t cap, size_t* out_len) {
  if (!codec || (!src && n) || (!dst && cap) || !out_len) BSMI_IO_FAIL("bsmi_codec_decode: null argument");
  return decode_any(codec, (const uint8_t*)src, n, (uint8_t*)dst, cap, out_len);
}

int bsmi_codec_encode(const bsmi_codec* codec, const void* src, size_t n, void* dst, size_t cap, size_t* out_len) {
  if (!codec || (!src && n) || !dst || !out_len) BSMI_IO_FAIL("bsmi_codec_encode: null argument");
  return encode_any(codec, (const uint8_t*)src, n, (uint8_t*)dst, cap, out_len);
}

int bsmi_chunks_read(const bsmi_codec* codec, int n, const char* const* paths, void* const* dsts, const size_t* caps,
                     size_t* lens, int* status, int threads) {
  if (!codec || n < 0 || (n && (!paths || !dsts || !caps || !lens || !status))) BSMI_IO_FAIL("bsmi_chunks_read: null argument");
  std::vector<std::string> msgs((size_t)n);
  run_pool(n, threads, [&](int i) {
    static thread_local std::vector<uint8_t> buf;
    int err = 0;
    lens[i] = 0;
    if (!read_file(paths[i], buf, &err)) {
      if (err == ENOENT) { status[i] = BSMI_CHUNK_MISSING; return; }
      status[i] = BSMI_ERR_INVALID;
      msgs[i] = std::string(paths[i]) + ": " + strerror(err);
      return;
    }
    status[i] = decode_any(codec, buf.data(), buf.size(), (uint8_t*)dsts[i], caps[i], &lens[i]);
    if (status[i] != BSMI_OK) msgs[i] = std::string(paths[i]) + ": " + bsmi_last_error();
  });
  for (int i = 0; i < n; ++i)
    if (status[i] < 0) BSMI_IO_FAIL("%s", msgs[i].c_str());
  return BSMI_OK;
}

int bsmi_chunks_write(const bsmi_codec* codec, int n, const char* const* paths, const void* const* srcs,
                      const size_t* sizes, int* status, int threads) {
  if (!codec || n < 0 || (n && (!paths || !srcs || !sizes || !status))) BSMI_IO_FAIL("bsmi_chunks_write: null argument");
  std::vector<std::string> msgs((size_t)n);
  run_pool(n, threads, [&](int i) {
    static thread_local std::vector<uint8_t> buf;
    buf.resize(bound_any(codec, sizes[i]));
    size_t len = 0;
    status[i] = encode_any(codec, (const uint8_t*)srcs[i], sizes[i], buf.data(), buf.size(), &len);
    if (status[i] != BSMI_OK) { msgs[i] = std::string(paths[i]) + ": " + bsmi_last_error(); return; }
    int err = 0;
    if (!write_file_atomic(paths[i], buf.data(), len, &err)) {
      status[i] = BSMI_ERR_INVALID;
      msgs[i] = std::string(paths[i]) + ": " + strerror(err);
    }
  });
  for (int i = 0; i < n; ++i)
    if (status[i] < 0) BSMI_IO_FAIL("%s", msgs[i].c_str());
  return BSMI_OK;
}


// ---- chunks <-> a strided host array, without a chunk-sized detour through the caller ---------------------------------------
extern "C++" {
namespace {
struct Region {
  size_t item, row_bytes, chunk_bytes, first, need;   // first / need: byte offset of the region's first element / end of its last
  int64_t cs[4], cstride[4];                          // chunk shape, byte strides inside the (C-order) chunk
};
bool region_of(const bsmi_chunk_copy& c, const int64_t cs[4], int itemsize, Region* r) {
  r->item = (size_t)itemsize;
  int64_t st = itemsize;
  for (int d = 3; d >= 0; --d) {
    r->cs[d] = cs[d];
    r->cstride[d] = st;
    st *= cs[d];
    if (c.start[d] < 0 || c.extent[d] < 1 || c.start[d] + c.extent[d] > cs[d]) return false;
  }
  if (c.stride[3] != itemsize) return false;
  r->chunk_bytes = (size_t)st;
  r->row_bytes = (size_t)c.extent[3] * r->item;
  r->first = 0;
  size_t last = 0;
  for (int d = 0; d < 4; ++d) {
    r->first += (size_t)(c.start[d] * r->cstride[d]);
    last += (size_t)((c.start[d] + c.extent[d] - 1) * r->cstride[d]);
  }
  r->need = last + r->item;
  return true;
}
template <class F>
void for_rows(const bsmi_chunk_copy& c, const Region& r, F&& f) {   // f(chunk byte offset, host pointer) per row of extent[3] items
  for (int64_t a = 0; a < c.extent[0]; ++a)
    for (int64_t b = 0; b < c.extent[1]; ++b)
      for (int64_t d = 0; d < c.extent[2]; ++d)
        f((size_t)((c.start[0] + a) * r.cstride[0] + (c.start[1] + b) * r.cstride[1] + (c.start[2] + d) * r.cstride[2] + c.start[3] * r.cstride[3]),
          (uint8_t*)c.base + a * c.stride[0] + b * c.stride[1] + d * c.stride[2]);
}
void fill_bytes(uint8_t* p, size_t n, const uint8_t* pat, size_t item) {
  bool zero = true;
  for (size_t i = 0; i < item; ++i) zero &= pat[i] == 0;
  if (zero) { memset(p, 0, n); return; }
  for (size_t i = 0; i + item <= n; i += item) memcpy(p + i, pat, item);
}
}  // namespace
}  // extern "C++"

int bsmi_chunks_read_into(const bsmi_codec* codec, int n, const bsmi_chunk_copy* copies, const int64_t chunk_shape[4], int itemsize,
                          const void* fill_value, int* status, int threads) {
  if (!codec || n < 0 || (n && (!copies || !status)) || !chunk_shape || itemsize < 1 || itemsize > 16) BSMI_IO_FAIL("bsmi_chunks_read_into: bad argument");
  uint8_t fill[16] = {0};
  if (fill_value) memcpy(fill, fill_value, (size_t)itemsize);
  for (int i = 0; i < n; ++i) {
    Region r;
    if (!copies[i].path || !copies[i].base || !region_of(copies[i], chunk_shape, itemsize, &r))
      BSMI_IO_FAIL("bsmi_chunks_read_into: region %d lies outside its chunk, or the host array's last axis is not contiguous", i);
  }
  std::vector<std::string> msgs((size_t)n);
  run_pool(n, threads, [&](int i) {
    const bsmi_chunk_copy& c = copies[i];
    Region r;
    region_of(c, chunk_shape, itemsize, &r);
    Scratch filebuf(r.chunk_bytes + 4096);
    std::vector<uint8_t>& file = filebuf.v;
    int err = 0;
    if (!read_file(c.path, file, &err)) {
      if (err != ENOENT) { status[i] = BSMI_ERR_INVALID; msgs[i] = std::string(c.path) + ": " + strerror(err); return; }
      status[i] = BSMI_CHUNK_MISSING;
      for_rows(c, r, [&](size_t, uint8_t* host) { fill_bytes(host, r.row_bytes, fill, r.item); });
      return;
    }
    Scratch plain(r.chunk_bytes);
    size_t len = 0;
    status[i] = decode_any(codec, file.data(), file.size(), plain.data(), r.chunk_bytes, &len, r.need);
    if (status[i] != BSMI_OK) { msgs[i] = std::string(c.path) + ": " + bsmi_last_error(); return; }
    if (len != r.chunk_bytes) {
      status[i] = BSMI_ERR_INVALID;
      msgs[i] = std::string(c.path) + ": chunk decodes to " + std::to_string(len) + " bytes, expected " + std::to_string(r.chunk_bytes);
      return;
    }
    for_rows(c, r, [&](size_t off, uint8_t* host) { memcpy(host, plain.data() + off, r.row_bytes); });
  });
  for (int i = 0; i < n; ++i)
    if (status[i] < 0) BSMI_IO_FAIL("%s", msgs[i].c_str());
  return BSMI_OK;
}

int bsmi_chunks_write_from(const bsmi_codec* codec, int n, const bsmi_chunk_copy* copies, const int64_t chunk_shape[4], int itemsize,
                           const void* fill_value, int* status, int threads) {
  if (!codec || n < 0 || (n && (!copies || !status)) || !chunk_shape || itemsize < 1 || itemsize > 16) BSMI_IO_FAIL("bsmi_chunks_write_from: bad argument");
  uint8_t fill[16] = {0};
  if (fill_value) memcpy(fill, fill_value, (size_t)itemsize);
  for (int i = 0; i < n; ++i) {
    Region r;
    if (!copies[i].path || !copies[i].base || !region_of(copies[i], chunk_shape, itemsize, &r))
      BSMI_IO_FAIL("bsmi_chunks_write_from: region %d lies outside its chunk, or the host array's last axis is not contiguous", i);
  }
  std::vector<std::string> msgs((size_t)n);
  run_pool(n, threads, [&](int i) {
    std::vector<uint8_t> file;   // (read-modify-write only)
    const bsmi_chunk_copy& c = copies[i];
    Region r;
    region_of(c, chunk_shape, itemsize, &r);
    Scratch plain(r.chunk_bytes);
    bool whole = true;
    for (int d = 0; d < 4; ++d) whole &= c.start[d] == 0 && c.extent[d] == r.cs[d];
    status[i] = BSMI_OK;
    if (!whole) {
      // the rest of the chunk: what the file holds (read_modify_write: the region covers only part of the chunk's valid extent),
      // else the fill value
      bool have = false;
      int err = 0;
      if (c.read_modify_write && read_file(c.path, file, &err)) {
        size_t len = 0;
        int rc = decode_any(codec, file.data(), file.size(), plain.data(), r.chunk_bytes, &len);
        if (rc != BSMI_OK || len != r.chunk_bytes) {
          status[i] = BSMI_ERR_INVALID;
          msgs[i] = std::string(c.path) + ": existing chunk cannot be read back: " + (rc != BSMI_OK ? bsmi_last_error() : "wrong size");
          return;
        }
        have = true;
      } else if (c.read_modify_write && err != ENOENT) {
        status[i] = BSMI_ERR_INVALID;
        msgs[i] = std::string(c.path) + ": " + strerror(err);
        return;
      }
      if (!have) fill_bytes(plain.data(), r.chunk_bytes, fill, r.item);
    }
    for_rows(c, r, [&](size_t off, uint8_t* host) { memcpy(plain.data() + off, host, r.row_bytes); });
    Scratch enc(bound_any(codec, r.chunk_bytes));
    size_t len = 0;
    status[i] = encode_any(codec, plain.data(), r.chunk_bytes, enc.data(), bound_any(codec, r.chunk_bytes), &len);
    if (status[i] != BSMI_OK) { msgs[i] = std::string(c.path) + ": " + bsmi_last_error(); return; }
    int err = 0;
    if (!write_file_atomic(c.path, enc.data(), len, &err)) {
      status[i] = BSMI_ERR_INVALID;
      msgs[i] = std::string(c.path) + ": " + strerror(err);
    }
  });
  for (int i = 0; i < n; ++i)
    if (status[i] < 0) BSMI_IO_FAIL("%s", msgs[i].c_str());
  return BSMI_OK;
}

}  // extern "C"
