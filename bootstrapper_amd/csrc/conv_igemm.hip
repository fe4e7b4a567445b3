// Implicit-GEMM 3-D "valid" convolution for gfx950 MFMA.
//
// GEMM view of a ConvPass layer (reference models/3d_affs/unet.py:7-76):
//   out[m][n] = act( bias[n] + sum_{step} sum_{k} A_step[m][k] * W_step[n][k] )
// where m runs over output voxels (z,y,x raster order), n over output channels and
// each K-step is one (source tensor, kernel tap, 128-byte channel chunk).  The cropped
// 1x1x1 residual branch of ConvPass (unet.py:38-41,67-71) and the channel concat of
// Upsample.forward (unet.py:223) are just more K-steps reading other tensors, so a
// whole ConvPass stage is one launch with a fused bias(+ReLU) epilogue.
//
// Data layout: activations channels-last [D][H][W][Cpad]; weights pre-packed on the
// host as [step][Npad][128 B] (k contiguous), zero padded.  Tiles are staged
// HBM/L2 -> registers -> LDS as [row][128 B] images with a 16-byte-chunk XOR swizzle
// (chunk ^= (row>>1)&7) so that the ds_read_b128 fragment reads of
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 are bank-conflict free.
#include "conv_igemm.h"

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

// T: element type; BM x BN block tile; WM x WN waves; each wave owns (BM/WM) x (BN/WN).
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int ROWB = 128;                  // bytes per tile row per K-step
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int RPP = NT / 8;                // rows covered per load pass
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                      // [2][BM][128]
  char* Bs = smem + 2 * BM * ROWB;      // [2][BN][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = a.Npad / BN;
  const int tile_n = blockIdx.x % ntn;
  const int tile_m = blockIdx.x / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int chunk = tid & 7;
  const int lrow = tid >> 3;

  // per-row base element offsets into each source tensor
  int rowbase[kMaxConvTensors][A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int m = m0 + lrow + i * RPP;
    m = m < a.M ? m : a.M - 1;
    const int x = m % a.Wo;
    const int zy = m / a.Wo;
    const int y = zy % a.Ho;
    const int z = zy / a.Ho;
#pragma unroll
    for (int t = 0; t < kMaxConvTensors; ++t)
      rowbase[t][i] = ((z * a.t[t].H + y) * a.t[t].W + x) * a.t[t].C;
  }
  const char* tptr[kMaxConvTensors];
#pragma unroll
  for (int t = 0; t < kMaxConvTensors; ++t) tptr[t] = (const char*)a.t[t].ptr;

  const char* wbase = (const char*)a.w + (size_t)(n0 + lrow) * ROWB + chunk * 16;
  const size_t wstep = (size_t)a.Npad * ROWB;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4_t ra[A_LOADS], rb[B_LOADS];

  auto load_global = [&](int s) {
    const KStep ks = a.steps[s];
    const bool valid = chunk < 2 * ks.nsub;
    const char* base = tptr[0];
    if (ks.tensor == 1) base = tptr[1];
    if (ks.tensor == 2) base = tptr[2];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      int rbv = rowbase[0][i];
      if (ks.tensor == 1) rbv = rowbase[1][i];
      if (ks.tensor == 2) rbv = rowbase[2][i];
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (valid)
        v = *(const u32x4_t*)(base + ((size_t)(uint32_t)(rbv + ks.a_off) * sizeof(T)) + chunk * 16);
      ra[i] = v;
    }
    const char* wp = wbase + (size_t)s * wstep;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) rb[i] = *(const u32x4_t*)(wp + (size_t)i * RPP * ROWB);
  };

  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int row = lrow + i * RPP;
      *(u32x4_t*)(As + buf * BM * ROWB + row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int row = lrow + i * RPP;
      *(u32x4_t*)(Bs + buf * BN * ROWB + row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4)) = rb[i];
    }
  };

  auto compute = [&](int buf, int nsub) {
    const char* Ab = As + buf * BM * ROWB;
    const char* Bb = Bs + buf * BN * ROWB;
    const int lr = lane & 31, lh = lane >> 5;
    for (int sub = 0; sub < nsub; ++sub) {
      const int c = 2 * sub + lh;
      u32x4_t fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 32 + lr;
        fa[i] = *(const u32x4_t*)(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + j * 32 + lr;
        fb[j] = *(const u32x4_t*)(Bb + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = Elem<T>::mfma(fa[i], fb[j], acc[i][j]);
    }
  };

  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < a.nsteps; ++s) {
    const int nsub = a.steps[s].nsub;
    if (s + 1 < a.nsteps) load_global(s + 1);
    compute(s & 1, nsub);
    if (s + 1 < a.nsteps) store_lds((s + 1) & 1);
    __syncthreads();
  }

  // epilogue: bias (+ReLU), convert, store channels-last
  T* out = (T*)a.out;
  const int lr = lane & 31, lh = lane >> 5;
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
  const double eff[] = {0.35, 0.6, 0.85, 1.0, 1.0};
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
    case TILE_256x320: return launch_one<T, 256, 320, 4, 2>(a, stream);
    case TILE_256x256: return launch_one<T, 256, 256, 2, 4>(a, stream);
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
