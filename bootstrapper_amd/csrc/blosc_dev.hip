// Blosc-1 frames of label volumes, encoded ON THE DEVICE.
//
// `bs segment --ws` ends by writing what it computed: fragments and one segmentation per threshold, uint64 [Z][Y][X], 8 bytes per
// voxel -- 32 GB for the 1024^3 benchmark volume -- into Zarr datasets whose chunks are Blosc frames (lz4, byte shuffle: what
// zarr-python's default compressor writes and what the reference's `prepare_ds` datasets hold, post/watershed.py:319-354).  Taken
// to the host first, those bytes cross PCIe (0.7 s at best) and then go through the host cores' shuffle and LZ4 passes; they are
// label volumes that compress 20-40 : 1.  Here the frames are made where the data lies: HBM-bound byte work, one pass over the
// volume per byte plane out of L2, a few percent of it written back out, and only the frames travel.
//
// Format (c-blosc 1.x, what csrc/chunk_codec.cpp reads and writes on the host): 16-byte header {2, 1, flags, typesize, nbytes,
// blocksize, cbytes}, block-start table, then per block of 256 KiB its `typesize` = 8 byte planes ("split" blocks), each as a
// 4-byte length + an LZ4 block, or the plane's 32 KiB verbatim when LZ4 does not shrink it.  The LZ4 blocks made here use matches
// at offset 1 only (runs of a byte: a plane of a label volume is little else) -- any LZ4 decoder reads them; liblz4 would also
// find the row above and pack 1.5-2x tighter, which is not worth a sequential search per plane.
//
// Kernel 1, one workgroup per (chunk, block, byte plane): gather the plane's 32 768 bytes from the strided volume into LDS; runs
// of at least 8 equal bytes become LZ4 sequences (literals up to and including the run's first byte, then a match of the rest at
// offset 1); three scans over the workgroup's 256 segments of 128 bytes place every sequence; the last 12 bytes stay literals
// (LZ4's end-of-block rules).  Kernel 2, one workgroup per chunk: block-start table, header, planes packed back to back.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "../../include/bsmi_io.h"

namespace bsmi {
namespace {

constexpr int kBlock = 256 * 1024;      // Blosc block size (bytes) of the frames
constexpr int kPlane = kBlock / 8;      // bytes of one byte plane of a block
constexpr int kSeg = kPlane / 256;      // bytes per thread
constexpr int kMinRun = 8;              // shortest run that becomes a match
constexpr int kPlaneSlot = kPlane + 4;  // scratch per plane: length prefix + data
constexpr uint32_t kNone = 0xffffffffu;

struct DevChunks {
  const uint64_t* src;
  long long sz, sy;             // element strides of the volume (x contiguous)
  const long long* origin;      // [n][3] first voxel of every chunk, relative to src
  const long long* extent;      // [n][3] voxels of the chunk that exist (the rest is fill = 0)
  int cz, cy, cx;               // chunk shape
  int nblocks;                  // blocks per chunk
  unsigned total_blocks;        // blocks of the request (the grid is padded to groups of 8)
};

__device__ __forceinline__ int lit_ext(int lit) { return lit >= 15 ? 1 + (lit - 15) / 255 : 0; }

// inclusive scans over the 256 threads' values in LDS (Hillis-Steele); `op`: 0 sum, 1 max, 2 min over the SUFFIX
template <int OP>
__device__ __forceinline__ uint32_t scan256(uint32_t v, uint32_t* buf, int tid) {
  buf[tid] = v;
  __syncthreads();
#pragma unroll
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t o;
    if (OP == 2) o = tid + d < 256 ? buf[tid + d] : kNone;
    else o = tid >= d ? buf[tid - d] : 0u;
    __syncthreads();
    if (OP == 0) v += o;
    else if (OP == 1) v = v > o ? v : o;
    else v = v < o ? v : o;
    buf[tid] = v;
    __syncthreads();
  }
  return v;
}

__global__ __launch_bounds__(256) void blosc_plane_kernel(const DevChunks a, uint8_t* __restrict__ scratch, uint32_t* __restrict__ plane_sizes) {
  __shared__ __attribute__((aligned(16))) uint8_t s[kPlane];
  __shared__ uint32_t buf[256];
  // the 8 planes of a block on ONE XCD (consecutive workgroup ids go round the 8 XCDs): they read the same 256 KiB
  const unsigned id = blockIdx.x;
  const unsigned plane = (id >> 3) & 7u;
  const unsigned blk_all = (id >> 6) * 8u + (id & 7u);
  if (blk_all >= a.total_blocks) return;   // (uniform: padding of the grid)
  const unsigned chunk = blk_all / (unsigned)a.nblocks, blk = blk_all % (unsigned)a.nblocks;
  const int tid = threadIdx.x;
  const long long oz = a.origin[3 * chunk], oy = a.origin[3 * chunk + 1], ox = a.origin[3 * chunk + 2];
  const long long ez = a.extent[3 * chunk], ey = a.extent[3 * chunk + 1], ex = a.extent[3 * chunk + 2];
  // gather: element e of the chunk (C order) -> its byte `plane`
  {
    const unsigned e0 = blk * (unsigned)kPlane;
    const unsigned cxy = (unsigned)(a.cx * a.cy);
    for (int j = 0; j < kPlane / 256; ++j) {
      const unsigned e = e0 + (unsigned)(j * 256 + tid);
      const unsigned z = e / cxy, r = e - z * cxy, y = r / (unsigned)a.cx, x = r - y * (unsigned)a.cx;
      uint64_t v = 0;
      if ((long long)z < ez && (long long)y < ey && (long long)x < ex) v = a.src[(oz + z) * a.sz + (oy + y) * a.sy + (ox + x)];
      s[j * 256 + tid] = (uint8_t)(v >> (8 * plane));
    }
  }
  __syncthreads();
  const int lo = tid * kSeg, hi = lo + kSeg;
  constexpr int n = kPlane;
  auto is_start = [&](int i) { return i == 0 || i >= n - 12 || s[i] != s[i - 1]; };
  // pass A: first run start of the segment; suffix-min -> the first start at or after the segment's end
  uint32_t first = kNone;
  for (int i = lo; i < hi; ++i)
    if (is_start(i)) { first = (uint32_t)i; break; }
  uint32_t after = scan256<2>(first, buf, tid);           // min over segments >= tid
  after = tid + 1 < 256 ? buf[tid + 1] : kNone;           // ... over segments > tid
  if (after == kNone) after = n;
  __syncthreads();
  // pass B: the segment's last long-run end; prefix-max -> the last long-run end before the segment
  auto for_long_runs = [&](auto&& f) {                    // f(start, length) for the long runs that START in this segment
    int p = -1;
    for (int i = lo; i <= hi; ++i) {
      const bool st = i < hi ? is_start(i) : true;
      if (!st) continue;
      const int here = i < hi ? i : (int)after;
      if (p >= 0 && here - p >= kMinRun) f(p, here - p);
      if (i < hi) p = i;
    }
  };
  uint32_t lle = 0;
  for_long_runs([&](int p, int L) { lle = (uint32_t)(p + L); });
  const uint32_t incl = scan256<1>(lle, buf, tid);
  const uint32_t before = tid ? buf[tid - 1] : 0u;
  const uint32_t q_last = buf[255];
  (void)incl;
  __syncthreads();
  // pass C: encoded sizes of the segment's sequences; exclusive sum -> where they go
  uint32_t mine = 0;
  {
    int q = (int)before;
    for_long_runs([&](int p, int L) {
      const int lit = p - q + 1, ml = L - 1;
      mine += 1 + lit_ext(lit) + lit + 2 + lit_ext(ml - 4);
      q = p + L;
    });
  }
  const uint32_t incl_sum = scan256<0>(mine, buf, tid);
  const uint32_t where = incl_sum - mine;
  const uint32_t seq_total = buf[255];
  const int lit_f = n - (int)q_last;
  const uint32_t csize = seq_total + 1 + lit_ext(lit_f) + lit_f;
  uint8_t* slot = scratch + ((size_t)(chunk * (unsigned)a.nblocks + blk) * 8 + plane) * kPlaneSlot;
  uint8_t* out = slot + 4;
  if (csize >= (uint32_t)n) {  // no gain: the plane verbatim, marked by length == plane size
    for (int j = tid; j < n / 4; j += 256) ((uint32_t*)out)[j] = ((const uint32_t*)s)[j];   // (slot + 4 is 4-byte aligned, not more)
    if (tid == 0) { *(uint32_t*)slot = (uint32_t)n; plane_sizes[(chunk * (unsigned)a.nblocks + blk) * 8 + plane] = (uint32_t)n; }
    return;
  }
  auto put_len = [&](uint8_t*& o, int v) {                // the 255-steps of a length >= 15 (v = length - 15)
    for (; v >= 255; v -= 255) *o++ = 255;
    *o++ = (uint8_t)v;
  };
  {
    int q = (int)before;
    uint8_t* o = out + where;
    for_long_runs([&](int p, int L) {
      const int lit = p - q + 1, m4 = L - 1 - 4;
      *o++ = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (m4 < 15 ? m4 : 15));
      if (lit >= 15) put_len(o, lit - 15);
      for (int i = 0; i < lit; ++i) *o++ = s[q + i];
      *o++ = 1;
      *o++ = 0;
      if (m4 >= 15) put_len(o, m4 - 15);
      q = p + L;
    });
  }
  // the closing literals: header by one thread, the bytes by all
  uint8_t* of = out + seq_total;
  const int hdr = 1 + lit_ext(lit_f);
  if (tid == 0) {
    uint8_t* o = of;
    *o++ = (uint8_t)((lit_f < 15 ? lit_f : 15) << 4);
    if (lit_f >= 15) put_len(o, lit_f - 15);
    *(uint32_t*)slot = csize;
    plane_sizes[(chunk * (unsigned)a.nblocks + blk) * 8 + plane] = csize;
  }
  for (int i = tid; i < lit_f; i += 256) of[hdr + i] = s[q_last + i];
}

// one workgroup per chunk: header, block starts, planes back to back.  A frame that would not be smaller than the chunk is
// stored (flag 2: header + the chunk's bytes), as c-blosc does.
__global__ __launch_bounds__(256) void blosc_frame_kernel(const DevChunks a, const uint8_t* __restrict__ scratch, const uint32_t* __restrict__ plane_sizes,
                                                          uint8_t* __restrict__ frames, size_t slot_bytes, uint32_t* __restrict__ frame_sizes) {
  __shared__ uint32_t start[1024 + 1];   // frame offset of every block (nblocks <= 1024)
  const unsigned chunk = blockIdx.x;
  const int tid = threadIdx.x, nb = a.nblocks;
  uint8_t* f = frames + (size_t)chunk * slot_bytes;
  const uint32_t nbytes = (uint32_t)nb * kBlock;
  if (tid == 0) {
    uint32_t off = 16 + 4 * (uint32_t)nb;
    for (int b = 0; b < nb; ++b) {
      start[b] = off;
      for (int p = 0; p < 8; ++p) off += 4 + plane_sizes[((size_t)chunk * nb + b) * 8 + p];
    }
    start[nb] = off;
  }
  __syncthreads();
  const uint32_t cbytes = start[nb];
  const bool stored = cbytes >= nbytes + 16;
  if (tid == 0) {
    f[0] = 2; f[1] = 1; f[2] = (uint8_t)((stored ? 2 : 0) | 1 | (1 << 5)); f[3] = 8;   // byte shuffle, lz4 format
    const uint32_t w[3] = {nbytes, (uint32_t)kBlock, stored ? nbytes + 16 : cbytes};
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 4; ++j) f[4 + 4 * k + j] = (uint8_t)(w[k] >> (8 * j));
    frame_sizes[chunk] = stored ? nbytes + 16 : cbytes;
  }
  if (stored) {  // the chunk's own bytes (not shuffled)
    const long long oz = a.origin[3 * chunk], oy = a.origin[3 * chunk + 1], ox = a.origin[3 * chunk + 2];
    const long long ez = a.extent[3 * chunk], ey = a.extent[3 * chunk + 1], ex = a.extent[3 * chunk + 2];
    const unsigned cxy = (unsigned)(a.cx * a.cy), ne = nbytes / 8;
    for (unsigned e = tid; e < ne; e += 256) {
      const unsigned z = e / cxy, r = e - z * cxy, y = r / (unsigned)a.cx, x = r - y * (unsigned)a.cx;
      uint64_t v = 0;
      if ((long long)z < ez && (long long)y < ey && (long long)x < ex) v = a.src[(oz + z) * a.sz + (oy + y) * a.sy + (ox + x)];
      for (int j = 0; j < 8; ++j) f[16 + (size_t)e * 8 + j] = (uint8_t)(v >> (8 * j));
    }
    return;
  }
  for (int b = tid; b < nb; b += 256)
    for (int j = 0; j < 4; ++j) f[16 + 4 * b + j] = (uint8_t)(start[b] >> (8 * j));
  // planes: a wave per (block, plane) in turn, bytes of a plane by the wave's lanes
  const int wave = tid >> 6, lane = tid & 63;
  for (int bp = wave; bp < nb * 8; bp += 4) {
    const int b = bp >> 3, p = bp & 7;
    uint32_t off = start[b];
    for (int k = 0; k < p; ++k) off += 4 + plane_sizes[((size_t)chunk * nb + b) * 8 + k];
    const uint32_t len = 4 + plane_sizes[((size_t)chunk * nb + b) * 8 + p];
    const uint8_t* src = scratch + (((size_t)chunk * nb + b) * 8 + p) * kPlaneSlot;
    for (uint32_t i = lane; i < len; i += 64) f[off + i] = src[i];
  }
}

}  // namespace
}  // namespace bsmi

using namespace bsmi;

extern "C" {

size_t bsmi_blosc_dev_frame_bound(size_t chunk_bytes) { return chunk_bytes + 16 + 4 * (chunk_bytes / kBlock) + 8 * 4 * (chunk_bytes / kBlock) + 64; }

size_t bsmi_blosc_dev_scratch_bytes(int n_chunks, size_t chunk_bytes) {
  return (size_t)n_chunks * (chunk_bytes / kBlock) * 8 * ((size_t)kPlaneSlot + 4) + (size_t)n_chunks * 6 * sizeof(long long) + 256;
}

int bsmi_blosc_encode_dev_u64(int device, const uint64_t* src_dev, int64_t stride_z, int64_t stride_y, int n_chunks, const int64_t* origins,
                              const int64_t* extents, const int64_t chunk_shape[3], void* scratch_dev, size_t scratch_bytes, void* frames_dev,
                              size_t slot_bytes, uint32_t* frame_sizes_dev, void* stream) {
  if (!src_dev || !origins || !extents || !chunk_shape || !scratch_dev || !frames_dev || !frame_sizes_dev || n_chunks < 1)
    BSMI_FAIL(BSMI_ERR_INVALID, "bsmi_blosc_encode_dev_u64: null argument");
  const size_t chunk_bytes = (size_t)chunk_shape[0] * chunk_shape[1] * chunk_shape[2] * 8;
  if (chunk_shape[0] < 1 || chunk_shape[1] < 1 || chunk_shape[2] < 1 || chunk_bytes % kBlock || chunk_bytes / kBlock > 1024 || chunk_bytes >= ((size_t)1 << 31))
    BSMI_FAIL(BSMI_ERR_INVALID, "device Blosc frames take chunks of whole 256 KiB blocks (at most 1024 of them): a chunk of %zu bytes goes through the host codec", chunk_bytes);
  if (slot_bytes < bsmi_blosc_dev_frame_bound(chunk_bytes) || scratch_bytes < bsmi_blosc_dev_scratch_bytes(n_chunks, chunk_bytes))
    BSMI_FAIL(BSMI_ERR_INVALID, "device Blosc frames: scratch or frame slots too small");
  for (int i = 0; i < n_chunks; ++i)
    for (int d = 0; d < 3; ++d)
      if (origins[3 * i + d] < 0 || extents[3 * i + d] < 1 || extents[3 * i + d] > chunk_shape[d]) BSMI_FAIL(BSMI_ERR_INVALID, "device Blosc frames: chunk %d has a bad extent", i);
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)(chunk_bytes / kBlock);
  // scratch: [planes][kPlaneSlot] | plane sizes | origins, extents
  uint8_t* planes = (uint8_t*)scratch_dev;
  size_t off = (size_t)n_chunks * nb * 8 * kPlaneSlot;
  off = (off + 15) & ~(size_t)15;
  uint32_t* sizes = (uint32_t*)(planes + off);
  off += (size_t)n_chunks * nb * 8 * 4;
  off = (off + 15) & ~(size_t)15;
  long long* geo = (long long*)(planes + off);
  BSMI_HIP(hipMemcpyAsync(geo, origins, (size_t)n_chunks * 3 * sizeof(long long), hipMemcpyHostToDevice, s));
  BSMI_HIP(hipMemcpyAsync(geo + (size_t)n_chunks * 3, extents, (size_t)n_chunks * 3 * sizeof(long long), hipMemcpyHostToDevice, s));
  DevChunks a;
  a.src = src_dev; a.sz = stride_z; a.sy = stride_y; a.origin = geo; a.extent = geo + (size_t)n_chunks * 3;
  a.cz = (int)chunk_shape[0]; a.cy = (int)chunk_shape[1]; a.cx = (int)chunk_shape[2]; a.nblocks = nb;
  const unsigned total_blocks = (unsigned)n_chunks * (unsigned)nb;
  a.total_blocks = total_blocks;
  hipLaunchKernelGGL(blosc_plane_kernel, dim3((total_blocks + 7) / 8 * 64), dim3(256), 0, s, a, planes, sizes);
  hipLaunchKernelGGL(blosc_frame_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, a, (const uint8_t*)planes, (const uint32_t*)sizes, (uint8_t*)frames_dev,
                     slot_bytes, frame_sizes_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

}  // extern "C"
