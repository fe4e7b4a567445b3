// Implicit-GEMM 3-D "valid" convolution for gfx950 MFMA.
//
// GEMM view of a ConvPass layer (reference models/3d_affs/unet.py:7-76):
//   out[m][n] = act( bias[n] + sum_{step} sum_{k} A_step[m][k] * W_step[n][k] )
// where m runs over output voxels (z,y,x raster order), n over output channels and
// each K-step is four 32-byte units (kernel tap, 16-channel group) of one source tensor.  The cropped
// 1x1x1 residual branch of ConvPass (unet.py:38-41,67-71) and the channel concat of
// Upsample.forward (unet.py:223) are just more K-steps reading other tensors, so a
// whole ConvPass stage is one launch with a fused bias(+ReLU) epilogue.
//
// Data layout: activations channels-last [D][H][W][Cpad]; weights pre-packed on the
// host as [step][Npad][128 B] (k contiguous), zero padded.  Tiles are staged
// HBM/L2 -> LDS by LDS-DMA as [row][128 B] images with a 16-byte-chunk XOR swizzle
// (chunk ^= (row>>1)&7) so that the ds_read_b128 fragment reads of
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 are bank-conflict free.
#include "conv_igemm.h"

#include <cstdlib>

namespace bsmi {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct bf16_elem {
  uint16_t v;
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    // 4 x v_mfma_f32_32x32x2_f32: lane half h holds k = 4h..4h+3 of this 8-wide sub-step;
    // instruction t contracts k in {t, 4+t}.  Same permutation on A and B.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    return acc;
  }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_elem> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void store(bf16_elem* p, float v) {
    __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
  }
};

typedef const __attribute__((address_space(1))) char* gptr_t;
typedef __attribute__((address_space(3))) char* lptr_t;
typedef const __attribute__((address_space(4))) int32_t* cint_ptr_t;  // constant AS: scalar loads

// T: element type; BM x BN block tile; WM x WN waves (4 waves = one per SIMD, so each wave
// may use the whole 512-register file: large register tiles, few LDS reads per MFMA).
//
// Staging is LDS-DMA (global_load_lds_dwordx4): one wave instruction moves 8 tile rows x
// 128 B = 1 KiB; lane l lands at row (l>>3), 16-byte slot (l&7) of that KiB, and FETCHES the
// source chunk (l&7) ^ ((row>>1)&7): the swizzle is applied on the per-lane source address,
// the LDS image stays lane-linear.  Two LDS stages; stage s+1 is in flight while stage s is
// multiplied; one barrier per K-step, placed before the last sub-step's MFMAs so that the
// fragment reads of the next K-step are issued under them.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int NW = WM * WN;
  constexpr int ROWB = 128;  // bytes per tile row per K-step
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // LDS-DMA instructions per wave per K-step
  constexpr int STAGE = (BM + BN) * ROWB;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32");

  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A: BM rows | B: BN rows][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;  // 10 dwords per K-step
  const int nsteps = a.nsteps;

  // XCD-aware tile map: consecutive block ids are dealt round-robin to the 8 XCDs, so give
  // each XCD a contiguous run of tiles (same weight panel, neighbouring row panels -> L2 hits)
  const int mt = (a.M + BM - 1) / BM, ntn = a.Npad / BN;
  const int ntiles = mt * ntn;
  int tile;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_n = tile / mt, tile_m = tile - tile_n * mt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = lane >> 3, lchunk = lane & 7;
  // tile rows staged by this lane: row(i) = (i*NW + wave)*8 + lrow.  NW*8 is a multiple of 16,
  // so the swizzle key (row>>1)&7 -- hence the source chunk, its unit and 16-byte half -- is
  // the same for every i.
  static_assert((NW * 8) % 16 == 0, "swizzle key must not depend on the instruction index");
  const int skey = ((wave * 8 + lrow) >> 1) & 7;
  const int g = lchunk ^ skey;            // source chunk that lands in LDS slot lchunk
  const int unit = g >> 1;                // which of the K-step's 4 units this lane fetches
  const uint32_t hoff = (uint32_t)((g & 1) << 4);
  uint32_t zyx[A_INSTR];                  // output voxel of row(i), packed z:10 | y:11 | x:11
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + lrow;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    const int x = m % a.Wo;
    const int zy = m / a.Wo;
    zyx[i] = ((uint32_t)(zy / a.Ho) << 22) | ((uint32_t)(zy % a.Ho) << 11) | (uint32_t)x;
  }
  const uint32_t offb = (uint32_t)((n0 + wave * 8 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)a.Npad * ROWB;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto issue = [&](int s, int buf) {
    const cint_ptr_t d = steps + s * 10;
    const uint64_t base = (uint64_t)(uint32_t)d[0] | ((uint64_t)(uint32_t)d[1] << 32);
    const int sz = d[2], sy = d[3], sx = d[4];
    const int d0 = d[5], d1 = d[6], d2 = d[7], d3 = d[8];
    const gptr_t abase = (gptr_t)base;
    const gptr_t wbase = (gptr_t)a.w + (size_t)s * wstep;
    const lptr_t la = (lptr_t)(smem + buf * STAGE);
    const lptr_t lb = la + BM * ROWB;
    const int dl = unit == 0 ? d0 : (unit == 1 ? d1 : (unit == 2 ? d2 : d3));
    const uint32_t lofs = (uint32_t)dl + hoff;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int z = (int)(zyx[i] >> 22), y = (int)((zyx[i] >> 11) & 0x7ff), x = (int)(zyx[i] & 0x7ff);
      const uint32_t voff = (uint32_t)(z * sz + y * sy + x * sx) + lofs;
      __builtin_amdgcn_global_load_lds(abase + voff, la + (i * NW + wave) * 1024, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(wbase + (size_t)i * NW * 8 * ROWB + offb, lb + (i * NW + wave) * 1024, 16, 0, 0);
  };

  const int lr = lane & 31, lh = lane >> 5;
  // per-fragment LDS row offsets and swizzle keys are loop invariant
  uint32_t arow[FM], akey[FM], brow[FN], bkey[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm * WTM + i * 32 + lr;
    arow[i] = row * ROWB;
    akey[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * WTN + j * 32 + lr;
    brow[j] = BM * ROWB + row * ROWB;
    bkey[j] = (row >> 1) & 7;
  }

  u32x4_t fa[2][FM], fb[2][FN];  // fragment double buffer, indexed by compile-time constants only
  auto load_frags = [&](const char* st, int sub, u32x4_t* pa, u32x4_t* pb) {
    const uint32_t c = 2 * sub + lh;
#pragma unroll
    for (int i = 0; i < FM; ++i) pa[i] = *(const u32x4_t*)(st + arow[i] + ((c ^ akey[i]) << 4));
#pragma unroll
    for (int j = 0; j < FN; ++j) pb[j] = *(const u32x4_t*)(st + brow[j] + ((c ^ bkey[j]) << 4));
  };
  auto mma = [&](const u32x4_t* pa, const u32x4_t* pb) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = Elem<T>::mfma(pa[i], pb[j], acc[i][j]);
  };
  // K-step boundary: stage s+1 has landed everywhere and everybody is done reading stage s;
  // refill this stage's buffer with stage s+2 and fetch the first fragments of stage s+1.
  auto boundary = [&](int s) {
    if (s + 1 < nsteps) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 2 < nsteps) issue(s + 2, s & 1);
      load_frags(smem + ((s + 1) & 1) * STAGE, 0, fa[0], fb[0]);
    }
  };

  issue(0, 0);
  if (nsteps > 1) {
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR + B_INSTR) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  load_frags(smem, 0, fa[0], fb[0]);

  for (int s = 0; s < nsteps; ++s) {
    const char* st = smem + (s & 1) * STAGE;
    load_frags(st, 1, fa[1], fb[1]);
    mma(fa[0], fb[0]);
    load_frags(st, 2, fa[0], fb[0]);
    mma(fa[1], fb[1]);
    load_frags(st, 3, fa[1], fb[1]);
    mma(fa[0], fb[0]);
    // fa[1]/fb[1] (sub-step 3) must be in registers before anyone may overwrite the stage
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    boundary(s);
    mma(fa[1], fb[1]);
  }

  // epilogue: bias (+ReLU), convert, store channels-last
  T* out = (T*)a.out;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + lr;
    if (n >= a.Co) continue;
    const float bv = a.bias[n];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M) {
          float v = acc[i][j][r] + bv;
          if (a.relu) v = v > 0.f ? v : 0.f;
          Elem<T>::store(out + (size_t)m * a.Co + n, v);
        }
      }
    }
  }
}

int tile_bm(TileCfg) { return 256; }
int tile_bn(TileCfg c) {
  switch (c) {
    case TILE_256x32: return 32;
    case TILE_256x64: return 64;
    case TILE_256x160: return 160;
    case TILE_256x320: return 320;
    case TILE_256x256: return 256;
    default: return 0;
  }
}

TileCfg choose_tile(int cout) {
  // minimise padded N weighted by how well each tile keeps the MFMA pipe fed
  const TileCfg cands[] = {TILE_256x32, TILE_256x64, TILE_256x160, TILE_256x256, TILE_256x320};
  double eff[] = {0.35, 0.6, 0.85, 1.0, 1.0};
  // the 128x160 register tile of TILE_256x320 spills (320 accumulators exceed the 256 AGPRs);
  // keep it out of the choice unless asked for (experiments only)
  if (!getenv("BSMI_USE_320")) eff[4] = 0.01;
  TileCfg best = TILE_256x32;
  double bestc = 1e30;
  for (int i = 0; i < 5; ++i) {
    const int bn = tile_bn(cands[i]);
    const double cost = (double)round_up(cout, bn) / eff[i];
    if (cost < bestc - 1e-9) {
      bestc = cost;
      best = cands[i];
    }
  }
  return best;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_one(const ConvArgs& a, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN>;
  if (!attr_set) {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int grid = ceil_div(a.M, BM) * (a.Npad / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

template <typename T>
static int launch_cfg(const ConvArgs& a, TileCfg cfg, hipStream_t stream) {
  switch (cfg) {
    case TILE_256x32: return launch_one<T, 256, 32, 4, 1>(a, stream);
    case TILE_256x64: return launch_one<T, 256, 64, 4, 1>(a, stream);
    case TILE_256x160: return launch_one<T, 256, 160, 4, 1>(a, stream);
    case TILE_256x320: return launch_one<T, 256, 320, 2, 2>(a, stream);
    case TILE_256x256: return launch_one<T, 256, 256, 2, 2>(a, stream);
    default: BSMI_FAIL(BSMI_ERR_INVALID, "unknown tile config %d", (int)cfg);
  }
}

int launch_conv_igemm(const ConvArgs& a, int precision, TileCfg cfg, hipStream_t stream) {
  if (a.M <= 0 || a.nsteps <= 0 || a.Npad % tile_bn(cfg) != 0)
    BSMI_FAIL(BSMI_ERR_INVALID, "conv launch: bad geometry M=%d nsteps=%d Npad=%d", a.M, a.nsteps, a.Npad);
  if (precision == BSMI_PREC_F32) return launch_cfg<float>(a, cfg, stream);
  if (precision == BSMI_PREC_BF16) return launch_cfg<bf16_elem>(a, cfg, stream);
  BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
}

}  // namespace bsmi
