// First ConvPass of a raw-input U-Net (reference models/3d_affs/unet.py:63-76 with in_channels = 1,
// models/3d_affs/predict.py:147-149 normalisation) as one kernel, bf16 operands / f32 accumulate.
//
// As separate launches this pass is three HBM-bound kernels that cost 1.3 ms per 128^3 block: the 1-channel
// input is padded to a 16-channel tensor (240 MB) so that the implicit GEMM can read it, the first conv then
// multiplies 15 zero channels per tap, and the second conv stages each 32-byte activation row 27 times
// through the LDS.  Here a workgroup owns a 4 x 8 x 32 output tile:
//   phase 0  raw tile (8 x 12 x 36, with the two-voxel halo of both convs) -> LDS, normalised, bf16
//   phase 1  conv 1 on MFMA as  Y1^T = W1 (16 x 32) . X^T (32 x 16 voxels): K = the 27 taps (one 16x16x32
//            instruction per 16 voxels), the B fragment gathered from the raw tile; + bias, ReLU, bf16
//            -> LDS tile of conv-1 activations (6 x 10 x 34 voxels, two 8-channel halves)
//   phase 2  conv 2 as  OUT^T = W2 (16 x 432) . Y1^T: 14 K-steps of (2 taps x 16 channels), A fragments
//            (weights) resident in 56 registers, B fragments one ds_read_b128 per lane straight from the
//            activation tile at the tap's offset; + bias + residual (centre voxel x w_res), ReLU, bf16,
//            8-byte stores: the four lanes of a voxel write its 32-byte row.
// The transposed orientation (channels on the MFMA rows) puts a voxel's 4 consecutive channels in one lane, so
// both the LDS tile and the output are written with 8-byte accesses and no shuffle.
//
// SPLIT (BSMI_PREC_BF16X3): every operand is a (hi, lo) pair of bf16 values and every product hi*hi + lo*hi + hi*lo
// (conv_dev.h): the raw tile, the conv-1 tile and the weight fragments exist twice, a K-step is three MFMAs, the
// residual is an f32 multiply, and the output rows carry (hi, lo) vectors interleaved.  The doubled tiles take 145 KB
// of LDS: one workgroup of 8 waves per CU instead of two of 4.
#include "first_pass.h"

#include <algorithm>
#include <cstring>
#include <vector>

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

namespace bsmi {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ uint16_t to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float from_bf16(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

constexpr int TZ = 4, TY = 8, TX = 32;
constexpr int RZ = TZ + 4, RY = TY + 4, RX = TX + 4;        // raw tile
constexpr int AZ = TZ + 2, AY = TY + 2, AX = TX + 2;        // conv-1 activation tile
constexpr int NA = AZ * AY * AX, NA_PAD = (NA + 15) / 16 * 16;
constexpr int NOUT = TZ * TY * TX;
constexpr int kThreads = 256;

__device__ __forceinline__ constexpr int raw_tap(int k) {  // tap k of a 3x3x3 kernel as an offset in the raw tile
  return k < 27 ? ((k / 9) * RY + (k / 3) % 3) * RX + k % 3 : 0;
}
__device__ __forceinline__ constexpr int act_tap(int k) {
  return k < 27 ? ((k / 9) * AY + (k / 3) % 3) * AX + k % 3 : 0;
}

template <bool SPLIT>
struct FpGeom {
  static constexpr int kT = SPLIT ? 512 : kThreads;
  static constexpr int P = SPLIT ? 2 : 1;                     // planes: hi (and lo)
  static constexpr int XS = RZ * RY * RX;                      // uint16 per plane of the raw tile
  static constexpr int XS_PAD = (XS + 7) / 8 * 8;
  static constexpr int ACT = 2 * NA_PAD * 8;                   // uint16 per plane of the conv-1 tile
  static constexpr int kLds = (P * XS_PAD + P * ACT) * 2;      // bytes
};

__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)to_bf16(a) | ((uint32_t)to_bf16(b) << 16); }
__device__ __forceinline__ float lo_of(float v) { return v - from_bf16(to_bf16(v)); }

template <bool RAW_F32, bool SPLIT>
__global__ __launch_bounds__(FpGeom<SPLIT>::kT, SPLIT ? 1 : 2) void first_pass_kernel(FirstPassArgs a, int ntz, int nty, int ntx) {
  using G = FpGeom<SPLIT>;
  constexpr int kT = G::kT;
  extern __shared__ __attribute__((aligned(16))) uint16_t fp_lds[];
  uint16_t* const xs = fp_lds;                                 // [P][XS_PAD]
  uint16_t* const act = fp_lds + G::P * G::XS_PAD;             // [P][2 halves][NA_PAD][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int Do = a.D - 4, Ho = a.H - 4, Wo = a.W - 4;

  // weight fragments and per-lane constants, once per workgroup
  const u32x4_t w1f = *(const u32x4_t*)(a.w1a + lane * 4);
  u32x4_t w1l = w1f;
  u32x4_t w2f[14], w2l[SPLIT ? 14 : 1];
#pragma unroll
  for (int s = 0; s < 14; ++s) w2f[s] = *(const u32x4_t*)(a.w2a + (s * 64 + lane) * 4);
  if constexpr (SPLIT) {
    w1l = *(const u32x4_t*)(a.w1a + (64 + lane) * 4);
#pragma unroll
    for (int s = 0; s < 14; ++s) w2l[s] = *(const u32x4_t*)(a.w2a + ((14 + s) * 64 + lane) * 4);
  }
  float b1[4], b2[4], wr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b1[i] = a.vec[q * 4 + i];
    b2[i] = a.vec[16 + q * 4 + i];
    wr[i] = a.vec[32 + q * 4 + i];
  }
  int toff1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // k = 8 q + j; the four q variants are compile-time tables selected per lane
    const int t0 = raw_tap(j), t1 = raw_tap(8 + j), t2 = raw_tap(16 + j), t3 = raw_tap(24 + j);
    toff1[j] = q == 0 ? t0 : q == 1 ? t1 : q == 2 ? t2 : t3;
  }
  const int hi_tap = q >> 1, half = q & 1;
  auto mma = [](u32x4_t x, u32x4_t y, f32x4_t c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, y), c, 0, 0, 0);
  };

  const int ntiles = ntz * nty * ntx;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
    const int X0 = tx * TX, Y0 = ty * TY, Z0 = tz * TZ;
    __syncthreads();  // the previous tile's readers are done with xs / act

    // ---- phase 0: raw tile -> xs (normalised, bf16); zeros beyond the input (ragged last tiles).
    // All loads of a thread are issued before the first is used.
    {
      constexpr int kIter = (RZ * RY * RX + kT - 1) / kT;
      float x[kIter];
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int i = tid + it * kT;
        const int rx = i % RX, ry = (i / RX) % RY, rz = i / (RX * RY);
        const int gz = Z0 + rz, gy = Y0 + ry, gx = X0 + rx;
        x[it] = 0.f;
        if (i < RZ * RY * RX && gz < a.D && gy < a.H && gx < a.W) {
          const size_t idx = ((size_t)gz * a.H + gy) * a.W + gx;
          if constexpr (RAW_F32) x[it] = ((const float*)a.raw)[idx];
          else x[it] = (float)((const uint8_t*)a.raw)[idx];
        }
      }
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int i = tid + it * kT;
        float v = x[it];
        if constexpr (!RAW_F32) {
          v = v * (1.0f / 255.0f);
          if (a.raw_dtype == BSMI_RAW_U8) v = v * 2.0f + -1.0f;
        }
        if (i < RZ * RY * RX) {
          xs[i] = to_bf16(v);
          if constexpr (SPLIT) xs[G::XS_PAD + i] = to_bf16(lo_of(v));
        }
      }
    }
    __syncthreads();

    // ---- phase 1: conv 1 -> act
    for (int g = wave; g < NA_PAD / 16; g += kT / 64) {
      const int v = g * 16 + n;
      const int vc = v < NA ? v : NA - 1;
      const int az = vc / (AY * AX), r = vc - az * (AY * AX), ay = r / AX, ax = r - ay * AX;
      const int base = (az * RY + ay) * RX + ax;
      uint32_t p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = (uint32_t)xs[base + toff1[2 * j]] | ((uint32_t)xs[base + toff1[2 * j + 1]] << 16);
      const u32x4_t bfrag = {p[0], p[1], p[2], p[3]};
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mma(w1f, bfrag, acc);
      if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          p[j] = (uint32_t)xs[G::XS_PAD + base + toff1[2 * j]] | ((uint32_t)xs[G::XS_PAD + base + toff1[2 * j + 1]] << 16);
        const u32x4_t blo = {p[0], p[1], p[2], p[3]};
        acc = mma(w1l, bfrag, acc);
        acc = mma(w1f, blo, acc);
      }
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaxf(acc[i] + b1[i], 0.f);
      const u32x2_t pk = {pack2(y[0], y[1]), pack2(y[2], y[3])};
      uint16_t* dst = act + (((q >> 1) * NA_PAD + v) * 8 + (q & 1) * 4);
      *(u32x2_t*)dst = pk;
      if constexpr (SPLIT) {
        const u32x2_t pl = {pack2(lo_of(y[0]), lo_of(y[1])), pack2(lo_of(y[2]), lo_of(y[3]))};
        *(u32x2_t*)(dst + G::ACT) = pl;
      }
    }
    __syncthreads();

    // ---- phase 2: conv 2 + residual -> out
    for (int g = wave; g < NOUT / 16; g += kT / 64) {
      const int o = g * 16 + n;
      const int zo = o / (TY * TX), yo = (o / TX) % TY, xo = o % TX;
      const int base = (zo * AY + yo) * AX + xo;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        const int off = base + (hi_tap ? act_tap(2 * s + 1) : act_tap(2 * s));
        const uint16_t* src = act + (half * NA_PAD + off) * 8;
        const u32x4_t bfrag = *(const u32x4_t*)src;
        acc = mma(w2f[s], bfrag, acc);
        if constexpr (SPLIT) {
          const u32x4_t blo = *(const u32x4_t*)(src + G::ACT);
          acc = mma(w2l[s], bfrag, acc);
          acc = mma(w2f[s], blo, acc);
        }
      }
      const int gz = Z0 + zo, gy = Y0 + yo, gx = X0 + xo;
      if (gz < Do && gy < Ho && gx < Wo) {
        const int ci = ((zo + 2) * RY + yo + 2) * RX + xo + 2;
        float xc = from_bf16(xs[ci]);
        if constexpr (SPLIT) xc += from_bf16(xs[G::XS_PAD + ci]);
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = fmaxf(acc[i] + b2[i] + wr[i] * xc, 0.f);
        const u32x2_t pk = {pack2(y[0], y[1]), pack2(y[2], y[3])};
        const size_t vox = ((size_t)gz * Ho + gy) * Wo + gx;
        if constexpr (SPLIT) {
          // 16 channels per voxel as [8 hi][8 lo][8 hi][8 lo]: this lane's four channels start at 4 q
          uint16_t* dst = a.out + vox * 32 + (q >> 1) * 16 + (q & 1) * 4;
          const u32x2_t pl = {pack2(lo_of(y[0]), lo_of(y[1])), pack2(lo_of(y[2]), lo_of(y[3]))};
          *(u32x2_t*)dst = pk;
          *(u32x2_t*)(dst + 8) = pl;
        } else {
          *(u32x2_t*)(a.out + vox * 16 + q * 4) = pk;
        }
      }
    }
  }
}

uint16_t host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float host_bf16_round(float f) {
  uint32_t u = (uint32_t)host_bf16(f) << 16;
  float r;
  memcpy(&r, &u, 4);
  return r;
}

}  // namespace

void free_first_pass(FirstPassWeights& fw) {
  if (fw.w1a) (void)hipFree(fw.w1a);
  if (fw.w2a) (void)hipFree(fw.w2a);
  if (fw.vec) (void)hipFree(fw.vec);
  fw = FirstPassWeights();
}

int pack_first_pass(FirstPassWeights& fw, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* wres, const float* bres, bool split) {
  if (C < 1 || C > 16) BSMI_FAIL(BSMI_ERR_INVALID, "first pass: %d channels (1..16 supported)", C);
  // split: the lo images (what the bf16 rounding of a weight left over) follow the hi images
  const int P = split ? 2 : 1;
  std::vector<uint32_t> a1((size_t)P * 64 * 4, 0u), a2((size_t)P * 14 * 64 * 4, 0u);
  std::vector<float> vec(48, 0.f);
  auto put = [](std::vector<uint32_t>& img, size_t lane_base, int j, uint16_t h) {
    img[lane_base + j / 2] |= (uint32_t)h << (16 * (j & 1));
  };
  auto part = [](float w, int pl) { return pl == 0 ? host_bf16(w) : host_bf16(w - host_bf16_round(w)); };
  for (int pl = 0; pl < P; ++pl)
    for (int l = 0; l < 64; ++l) {
      const int m = l & 15, q = l >> 4;
      for (int j = 0; j < 8; ++j) {
        const int k = q * 8 + j;
        if (m < C && k < 27) put(a1, ((size_t)pl * 64 + l) * 4, j, part(w1[(size_t)m * 27 + k], pl));
      }
      for (int s = 0; s < 14; ++s) {
        const int tap = 2 * s + (q >> 1);
        for (int j = 0; j < 8; ++j) {
          const int c = (q & 1) * 8 + j;
          if (m < C && c < C && tap < 27)
            put(a2, (((size_t)pl * 14 + s) * 64 + l) * 4, j, part(w2[((size_t)m * C + c) * 27 + tap], pl));
        }
      }
    }
  for (int m = 0; m < C; ++m) {
    vec[m] = b1[m];
    vec[16 + m] = b2[m] + bres[m];
    vec[32 + m] = split ? wres[m] : host_bf16_round(wres[m]);  // split: the residual is an f32 multiply
  }
  if (!fw.w1a) BSMI_HIP(hipMalloc((void**)&fw.w1a, a1.size() * 4));
  if (!fw.w2a) BSMI_HIP(hipMalloc((void**)&fw.w2a, a2.size() * 4));
  if (!fw.vec) BSMI_HIP(hipMalloc((void**)&fw.vec, vec.size() * 4));
  BSMI_HIP(hipMemcpy(fw.w1a, a1.data(), a1.size() * 4, hipMemcpyHostToDevice));
  BSMI_HIP(hipMemcpy(fw.w2a, a2.data(), a2.size() * 4, hipMemcpyHostToDevice));
  BSMI_HIP(hipMemcpy(fw.vec, vec.data(), vec.size() * 4, hipMemcpyHostToDevice));
  fw.ready = true;
  fw.split = split;
  return BSMI_OK;
}

template <bool RAW_F32, bool SPLIT>
static int launch_fp(const FirstPassArgs& a, int grid, int ntz, int nty, int ntx, hipStream_t s) {
  using G = FpGeom<SPLIT>;
  auto kern = first_pass_kernel<RAW_F32, SPLIT>;
  static DeviceOnce once;
  const int rc_once = once.run([&]() -> int {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds));
    return BSMI_OK;
  });
  if (rc_once) return rc_once;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::kT), G::kLds, s, a, ntz, nty, ntx);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int launch_first_pass(const FirstPassArgs& a, int n_cus, hipStream_t s) {
  if (a.D < 5 || a.H < 5 || a.W < 5) BSMI_FAIL(BSMI_ERR_INVALID, "first pass: input (%d,%d,%d) too small", a.D, a.H, a.W);
  const int ntz = ceil_div(a.D - 4, TZ), nty = ceil_div(a.H - 4, TY), ntx = ceil_div(a.W - 4, TX);
  const int ntiles = ntz * nty * ntx;
  const int grid = std::min(ntiles, (a.split ? 1 : 2) * std::max(n_cus, 1));
  const bool f32 = a.raw_dtype == BSMI_RAW_F32;
  if (a.split) return f32 ? launch_fp<true, true>(a, grid, ntz, nty, ntx, s) : launch_fp<false, true>(a, grid, ntz, nty, ntx, s);
  return f32 ? launch_fp<true, false>(a, grid, ntz, nty, ntx, s) : launch_fp<false, false>(a, grid, ntz, nty, ntx, s);
}

}  // namespace bsmi
