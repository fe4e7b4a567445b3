// Box-halo 3x3x3 convolution, bf16 operands / f32 accumulate, for layers with at most 16 output channels,
// where the implicit-GEMM kernel is bound by its LDS staging: there every 32-byte activation row is staged once
// per tap (27 times) and meets only 16 output channels.  Here a workgroup owns a 4 x 8 x 16 box of output voxels
// and stages, per 16-channel chunk of the input, the box with its halo (6 x 10 x 18 voxels, 34 KB, LDS-DMA,
// double buffered) ONCE; all 27 taps read it at an offset.
//
// Orientation: OUT^T (16 channels x 16 voxels) = W (16 x 32) . ACT^T (32 x 16) with v_mfma_f32_16x16x32_bf16,
// K-step = 2 taps x 16 channels, 14 per chunk.  The weights are the A operand, the activations the B operand:
// one conflict-free ds_read_b128 per lane and instruction from the [8-channel half][voxel][16 B] image.  A voxel's
// 4 consecutive output channels land in one lane: 8-byte stores, no shuffle.
//
// The four waves split the box (8 voxel groups each) and keep a chunk's 14 weight fragments in registers, the next
// chunk's prefetched.  Dealing the K-steps to the waves instead (3-4 fragments per wave, partial sums reduced through
// the LDS) was measured and is no faster: the kernel is bound by the memory-side traffic of the staging -- a chunk
// uses 32 bytes of every 128-byte line of a 64-channel source, and the lines of one workgroup's 5 chunks do not
// survive in its XCD's L2 -- not by the weight fetches.  The cropped 1x1x1 residual branch follows as CENTER chunks
// (32 channels of the box's own voxels, no halo, one K-step).
#include "conv_box.h"

#include <algorithm>

#include "conv_dev.h"

namespace bsmi {

typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

namespace box {
constexpr int TZ = 4, TY = 8, TX = 16;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int NH = HZ * HY * HX;  // 1080 voxels of a halo tile
// voxels between the two 8-channel halves of the image: a multiple of 16, so that a ds_read_b128 lane group -- 8 lanes
// of one half and 8 of the other, 16 voxels apart in total -- covers 16 different 16-byte slots (all 64 banks)
constexpr int NHP = (NH + 15) / 16 * 16;
constexpr int NV = TZ * TY * TX;  // 512 voxels of the box = 32 groups of 16 x-consecutive voxels, g = (z = g / 8, y = g % 8)
constexpr int NG = NV / 16 / 4;   // groups per wave
constexpr int kThreads = 256;
constexpr int kFullInstr = (2 * NHP + kThreads - 1) / kThreads;  // 16-byte pieces of a halo image / threads
constexpr int kCenterInstr = 4 * NV / kThreads;
constexpr int kBuf = kFullInstr * kThreads * 16;                 // bytes of one staging buffer (>= 4 NV 16)
__device__ __forceinline__ constexpr int tap_bytes(int k) {  // tap k as a byte offset inside one half of the halo image
  return k < 27 ? (((k / 9) * HY + (k / 3) % 3) * HX + k % 3) * 16 : 0;
}
}  // namespace box

__device__ __forceinline__ uint16_t box_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

__global__ __launch_bounds__(box::kThreads, 2) void conv_box_kernel(BoxArgs a, int nty, int ntx) {
  using namespace box;
  __shared__ __attribute__((aligned(16))) char lds[2 * kBuf];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4, hi = q >> 1, half = q & 1;
  const int tile = blockIdx.x;
  const int X0 = (tile % ntx) * TX, Y0 = ((tile / ntx) % nty) * TY, Z0 = (tile / (ntx * nty)) * TZ;
  const lptr_t lbase = (lptr_t)lds;

  const int g0 = wave * NG;
  f32x4_t acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t acur[14], anext[14];

  auto stage_full = [&](int c, int buf) {
    const BoxChunk ck = a.chunks[c];
#pragma unroll
    for (int k = 0; k < kFullInstr; ++k) {
      int i = k * kThreads + tid;
      i = i < 2 * NHP ? i : 2 * NHP - 1;
      const int hf = i >= NHP;
      int v = i - hf * NHP;
      v = v < NH ? v : NH - 1;  // the pad between the halves is filled with a copy of the last voxel
      const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
      const int cz = min(Z0 + hz, ck.D - 1), cy = min(Y0 + hy, ck.H - 1), cx = min(X0 + hx, ck.W - 1);
      const size_t off = (size_t)cz * ck.sz + (size_t)cy * ck.sy + (size_t)cx * ck.sx + hf * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)ck.base + off, lbase + buf * kBuf + (k * kThreads + wave * 64) * 16, 16, 0, 0);
    }
  };
  auto stage_center = [&](int c, int buf) {
    const BoxChunk ck = a.chunks[c];
#pragma unroll
    for (int k = 0; k < kCenterInstr; ++k) {
      const int i = k * kThreads + tid;
      const int qd = i / NV, v = i - qd * NV;
      const int z = v / (TY * TX), y = (v / TX) % TY, x = v % TX;
      const int cz = min(Z0 + z, ck.D - 1), cy = min(Y0 + y, ck.H - 1), cx = min(X0 + x, ck.W - 1);
      const size_t off = (size_t)cz * ck.sz + (size_t)cy * ck.sy + (size_t)cx * ck.sx + qd * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)ck.base + off, lbase + buf * kBuf + (k * kThreads + wave * 64) * 16, 16, 0, 0);
    }
  };
  auto load_full_w = [&](int c, u32x4_t* dst) {
#pragma unroll
    for (int s = 0; s < 14; ++s) dst[s] = *(const u32x4_t*)(a.w + (((size_t)c * 14 + s) * 64 + lane) * 4);
  };
  auto load_center_w = [&](int j, u32x4_t* dst) { dst[0] = *(const u32x4_t*)(a.w + (((size_t)a.n_full * 14 + j) * 64 + lane) * 4); };
  auto fence = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  const int nf = a.n_full, nc = a.n_center;
  stage_full(0, 0);
  load_full_w(0, acur);
  fence();
  for (int c = 0; c < nf; ++c) {
    const int buf = c & 1;
    if (c + 1 < nf) {
      stage_full(c + 1, buf ^ 1);
      load_full_w(c + 1, anext);
    } else if (nc > 0) {
      stage_center(nf, buf ^ 1);
      load_center_w(0, anext);
    }
    const uint32_t rd = (uint32_t)(buf * kBuf + half * (NHP * 16) + n * 16);
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const uint32_t ts = rd + (hi ? tap_bytes(2 * s + 1) : tap_bytes(2 * s));
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int gg = g0 + g;
        const u32x4_t b = *(const u32x4_t*)(lds + ts + (uint32_t)(((gg >> 3) * HY + (gg & 7)) * HX) * 16);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8_t, acur[s]), __builtin_bit_cast(s16x8_t, b), acc[g], 0, 0, 0);
      }
    }
    fence();
#pragma unroll
    for (int s = 0; s < 14; ++s) acur[s] = anext[s];
  }
  for (int j = 0; j < nc; ++j) {
    const int buf = (nf + j) & 1;
    if (j + 1 < nc) {
      stage_center(nf + j + 1, buf ^ 1);
      load_center_w(j + 1, anext);
    }
    const uint32_t rd = (uint32_t)(buf * kBuf + q * (NV * 16) + n * 16);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const u32x4_t b = *(const u32x4_t*)(lds + rd + (uint32_t)(g0 + g) * 256);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8_t, acur[0]), __builtin_bit_cast(s16x8_t, b), acc[g], 0, 0, 0);
    }
    fence();
    acur[0] = anext[0];
  }

  float bias[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bias[i] = a.bias[q * 4 + i];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int gg = g0 + g;
    const int z = Z0 + (gg >> 3), y = Y0 + (gg & 7), x = X0 + n;
    if (z < a.Do && y < a.Ho && x < a.Wo) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[g][e] + bias[e], 0.f);
      const u32x2_t pk = {(uint32_t)box_bf16(v[0]) | ((uint32_t)box_bf16(v[1]) << 16), (uint32_t)box_bf16(v[2]) | ((uint32_t)box_bf16(v[3]) << 16)};
      *(u32x2_t*)(a.out + (((size_t)z * a.Ho + y) * a.Wo + x) * a.Co + q * 4) = pk;
    }
  }
}

int launch_conv_box(const BoxArgs& a, hipStream_t s) {
  using namespace box;
  if (a.n_full < 1) BSMI_FAIL(BSMI_ERR_INVALID, "box conv without a full chunk");
  const int ntz = ceil_div(a.Do, TZ), nty = ceil_div(a.Ho, TY), ntx = ceil_div(a.Wo, TX);
  hipLaunchKernelGGL(conv_box_kernel, dim3((unsigned)(ntz * nty * ntx)), dim3(kThreads), 0, s, a, nty, ntx);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

}  // namespace bsmi
