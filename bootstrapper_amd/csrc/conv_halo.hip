// Halo-tiled implicit-GEMM 3-D convolution on MFMA (gfx950).
//
// Same GEMM as conv_igemm.hip, but the M tile is a 3-D box of output voxels (TZ x TY x TX <= 256)
// and the activation operand is not gathered per kernel tap: for every 16-channel slice of a
// source tensor the box's input halo ((TZ+kz-1) x (TY+ky-1) x (TX+kx-1) voxels x 32 B) is staged
// ONCE into LDS and all kz*ky*kx taps read their A fragments from it at a per-tap row offset.
// This cuts the activation traffic into the CU by ~10x (the kernel is bound by the L2 -> LDS
// fill rate, not by MFMA issue); the weight tiles still stream through a 4-slot LDS-DMA ring.
//
// LDS: [halo buffer 0 | halo buffer 1 | 4 x weight slot (BNL rows x 64 B)].
// A "phase" is one staged halo; its K-steps pair two kernel taps (2 x 16 channels of K).  The
// 1x1x1 residual branch uses SHORT phases: the box's own 256 voxels x 32 channels (64-B rows).
// Halo p+2 is issued at the boundary that ends phase p (its buffer is free from then on); the
// host simulates the in-order vmcnt queue and stores, per boundary, which counted wait is safe.
//
// STATUS (round 1): numerically verified (all U-Net parity tests pass with BSMI_USE_HALO=1) but
// 1.4-2.5x SLOWER than the gather kernel: the A-fragment reads of non-contiguous halo rows hit
// 20-50 % LDS bank conflicts (32-byte rows leave one swizzle bit) and the per-tile set-up divides
// dominate short K loops.  Opt-in only; the fix (64-byte halo rows, row pitches chosen so that the
// 16 lanes of a ds_read_b128 group fall on 16 distinct rows mod 16) is next round's work.
#include "conv_halo.h"

#include <algorithm>
#include <cstdlib>

namespace bsmi {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gptr_t;
typedef __attribute__((address_space(3))) char* lptr_t;
typedef const __attribute__((address_space(4))) int32_t* cint_ptr_t;

struct hbf16 { uint16_t v; };

template <typename T> struct HElem;
template <> struct HElem<float> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    return acc;
  }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct HElem<hbf16> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void store(hbf16* p, float v) {
    __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
  }
};

// wait variants of a K-step boundary (HaloStep::wait); BI = weight LDS-DMA instructions per wave
// per K-step, HL / HS = instructions per wave of a LONG / SHORT halo
enum { W_2B = 0, W_2B_HL = 1, W_2B_HS = 2, W_1B = 3, W_ALL = 4, W_1B_HS = 5, W_1B_HL = 6 };

template <typename T, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 1) void conv_halo_kernel(const HaloArgs a) {
  constexpr int BM = 256, NW = 4;
  static_assert(WM * WN == NW, "one wave per SIMD");
  constexpr int ROWB = kStepRowBytes;
  constexpr int NSLOT = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int BNL = (BN + 63) / 64 * 64;
  constexpr int BI = BNL / 16 / NW;
  constexpr int HL = kHaloLongInstr, HS = kHaloShortInstr;
  constexpr int SLOT = BNL * ROWB;
  constexpr int BRING = 2 * kHaloBufBytes;  // LDS offset of the weight ring
  static_assert(2 * BI + HL <= 63, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;    // 8 dwords per K-step
  const cint_ptr_t phases = (cint_ptr_t)a.phases;  // 4 dwords per phase
  const int nsteps = a.nsteps;

  // tile map (as conv_igemm.hip): XCD-contiguous runs of tiles, weight panel fastest
  const int nbox = a.NBZ * a.NBY * a.NBX, ntn = a.Npad / BN;
  const int ntiles = nbox * ntn;
  int tile;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int box = tile / ntn, tile_n = tile - box * ntn;
  const int n0 = tile_n * BN;
  const int bx = box % a.NBX, byz = box / a.NBX;
  const int by = byz % a.NBY, bz = byz / a.NBY;
  const int z0 = bz * a.TZ, y0 = by * a.TY, x0 = bx * a.TX;
  const int boxrows = a.TZ * a.TY * a.TX;

  // ---- per-lane source offsets of the halo rows this lane stages (computed once per tile) ----
  // LONG: instruction q covers halo rows (q*NW + wave)*32 .. +32, lane -> row (lane>>1), half lane&1
  const int HYX = a.HY * a.HX;
  uint32_t lofs0[HL], lofs1[HL], lofs2[HL];
  {
#pragma unroll
    for (int q = 0; q < HL; ++q) {
      int j = (q * NW + wave) * 32 + (lane >> 1);
      j = j < a.hv_long ? j : a.hv_long - 1;
      const int jz = j / HYX, jr = j - jz * HYX;
      const int jy = jr / a.HX, jx = jr - jy * a.HX;
      auto off = [&](const HaloSrc& t) -> uint32_t {
        int z = z0 + t.oz + jz, y = y0 + t.oy + jy, x = x0 + t.ox + jx;
        z = z < t.D ? z : t.D - 1; y = y < t.H ? y : t.H - 1; x = x < t.W ? x : t.W - 1;
        return (uint32_t)(z * t.sz + y * t.sy + x * t.sx);
      };
      lofs0[q] = off(a.t[0]); lofs1[q] = off(a.t[1]); lofs2[q] = off(a.t[2]);
    }
  }
  // SHORT: instruction q covers box rows (q*NW + wave)*16 .. +16, lane -> row (lane>>2), chunk lane&3
  uint32_t sofs0[HS], sofs1[HS], sofs2[HS];
  {
#pragma unroll
    for (int q = 0; q < HS; ++q) {
      int m = (q * NW + wave) * 16 + (lane >> 2);
      m = m < boxrows ? m : 0;
      const int tz = m / (a.TY * a.TX), mr = m - tz * (a.TY * a.TX);
      const int ty = mr / a.TX, tx = mr - ty * a.TX;
      auto off = [&](const HaloSrc& t) -> uint32_t {
        int z = z0 + t.rz + tz, y = y0 + t.ry + ty, x = x0 + t.rx + tx;
        z = z < t.D ? z : t.D - 1; y = y < t.H ? y : t.H - 1; x = x < t.W ? x : t.W - 1;
        return (uint32_t)(z * t.sz + y * t.sy + x * t.sx);
      };
      sofs0[q] = off(a.t[0]); sofs1[q] = off(a.t[1]); sofs2[q] = off(a.t[2]);
    }
  }
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;
  // source sub-chunk this lane fetches (swizzle on the source side; LDS image lane-linear)
  const uint32_t lsrc = (uint32_t)(((lane & 1) ^ (((lane >> 1) >> 3) & 1)) << 4);        // LONG: half ^ (row>>3)&1
  const uint32_t ssrc = (uint32_t)(((lane & 3) ^ (((lane >> 2) >> 2) & 3)) << 4);        // SHORT: chunk ^ (row>>2)&3
  // (row>>3)&1 / (row>>2)&3 of the staged row only depend on the lane: the per-instruction row
  // offsets (q*NW+wave)*32 resp. *16 are multiples of 16.

  auto issue_halo = [&](int p) {
    const cint_ptr_t d = phases + p * 4;
    const int t = d[0], c0 = d[1], kind = d[2], bufbase = d[3];
    const bool t1 = t == 1, t2 = t == 2;
    const gptr_t hb = (gptr_t)(t1 ? base1 : (t2 ? base2 : base0));
    const lptr_t lh = (lptr_t)(smem + bufbase);
    if (kind == 0) {
#pragma unroll
      for (int q = 0; q < HL; ++q) {
        const uint32_t ro = t1 ? lofs1[q] : (t2 ? lofs2[q] : lofs0[q]);
        __builtin_amdgcn_global_load_lds(hb + (size_t)(ro + (uint32_t)c0 + lsrc), lh + (q * NW + wave) * 1024, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < HS; ++q) {
        const uint32_t ro = t1 ? sofs1[q] : (t2 ? sofs2[q] : sofs0[q]);
        __builtin_amdgcn_global_load_lds(hb + (size_t)(ro + (uint32_t)c0 + ssrc), lh + (q * NW + wave) * 1024, 16, 0, 0);
      }
    }
  };

  // ---- weight ring ------------------------------------------------------------------------------
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = (lane >> 4) & 3;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)a.Npad * ROWB;
  auto issue_b = [&](int h) {
    const gptr_t wbase = (gptr_t)a.w + (size_t)(h < nsteps ? h : nsteps - 1) * wstep;
    const lptr_t lb = (lptr_t)(smem + BRING + (h & (NSLOT - 1)) * SLOT);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      __builtin_amdgcn_global_load_lds(wbase + (size_t)i * NW * 16 * ROWB + offb, lb + (i * NW + wave) * 1024, 16, 0, 0);
  };

  // ---- fragments -------------------------------------------------------------------------------
  const int lr = lane & 31, lh = lane >> 5;
  int hrowL[FM], hrowS[FM];  // halo row of this lane's A-fragment rows in LONG / SHORT phases
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int m = wm * WTM + i * 32 + lr;
    m = m < boxrows ? m : 0;
    const int tz = m / (a.TY * a.TX), mr = m - tz * (a.TY * a.TX);
    const int ty = mr / a.TX, tx = mr - ty * a.TX;
    hrowL[i] = (tz * a.HY + ty) * a.HX + tx;
    hrowS[i] = m;
  }
  uint32_t brow[FN], bkey[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * WTN + j * 32 + lr;
    brow[j] = row * ROWB;
    bkey[j] = (row >> 2) & 3;
  }

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct SDesc { int trow0, trow1, fmt, bufbase, wait, issue; };
  auto fetch = [&](int h) -> SDesc {
    const cint_ptr_t d = steps + (h < nsteps ? h : nsteps - 1) * 8;
    return SDesc{d[0], d[1], d[2], d[3], d[4], d[5]};
  };

  u32x4_t fa[2][FM], fb[2][FN];
  auto load_frags = [&](const SDesc& ds, int h, int sub, u32x4_t* pa, u32x4_t* pb) {
    const int trow = sub ? ds.trow1 : ds.trow0;
    const int cb = (sub ? (ds.fmt >> 8) : ds.fmt) & 0xff;
    const int rsh = (ds.fmt >> 16) & 0xf, ksh = (ds.fmt >> 20) & 0xf, kmask = (ds.fmt >> 24) & 0xf;
    const bool is_short = rsh == 6;
    const char* hbuf = smem + ds.bufbase;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = (is_short ? hrowS[i] : hrowL[i]) + trow;
      const int chunk = (cb + lh) ^ ((row >> ksh) & kmask);
      pa[i] = *(const u32x4_t*)(hbuf + ((uint32_t)row << rsh) + (chunk << 4));
    }
    const char* bs = smem + BRING + (h & (NSLOT - 1)) * SLOT;
    const uint32_t c = 2 * sub + lh;
#pragma unroll
    for (int j = 0; j < FN; ++j) pb[j] = *(const u32x4_t*)(bs + brow[j] + ((c ^ bkey[j]) << 4));
  };
  auto mma = [&](const u32x4_t* pa, const u32x4_t* pb) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = HElem<T>::mfma(pa[i], pb[j], acc[i][j]);
  };

  // prologue: halos of phases 0 and 1, weight K-steps 0..2; everything must land
  issue_halo(0);
  if (a.nphases > 1) issue_halo(1);
  issue_b(0);
  issue_b(1);
  issue_b(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  SDesc dcur = fetch(0), dnext = fetch(1);
  load_frags(dcur, 0, 0, fa[0], fb[0]);
  issue_b(3);

  for (int h = 0; h < nsteps; ++h) {
    load_frags(dcur, h, 1, fa[1], fb[1]);
    mma(fa[0], fb[0]);
    // boundary: fa[1]/fb[1] are in registers; weight K-step h+1 (and a halo first used by step
    // h+1) must have landed everywhere.  The host picked the counted wait that guarantees it.
    switch (dcur.wait) {
      case W_2B:    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * BI) : "memory"); break;
      case W_2B_HL: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * BI + HL) : "memory"); break;
      case W_2B_HS: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * BI + HS) : "memory"); break;
      case W_1B:    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BI) : "memory"); break;
      case W_1B_HS: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BI + HS) : "memory"); break;
      case W_1B_HL: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BI + HL) : "memory"); break;
      default:      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    if (dcur.issue >= 0) issue_halo(dcur.issue);
    issue_b(h + 4);
    const SDesc dn2 = fetch(h + 2);
    load_frags(dnext, h + 1, 0, fa[0], fb[0]);
    mma(fa[1], fb[1]);
#pragma unroll
    for (int k = 0; k < BI; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
    }
#pragma unroll
    for (int k = 0; k < FM + FN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
    }
    dcur = dnext;
    dnext = dn2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: bias (+ReLU), convert, store channels-last; rows are box voxels
  T* out = (T*)a.out;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= boxrows) continue;
      const int tz = m / (a.TY * a.TX), mr = m - tz * (a.TY * a.TX);
      const int ty = mr / a.TX, tx = mr - ty * a.TX;
      const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
      if (z >= a.Do || y >= a.Ho || x >= a.Wo) continue;
      const size_t vox = ((size_t)z * a.Ho + y) * a.Wo + x;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + lr;
        if (n >= a.Co) continue;
        float v = acc[i][j][r] + a.bias[n];
        if (a.relu) v = v > 0.f ? v : 0.f;
        HElem<T>::store(out + vox * a.Co + n, v);
      }
    }
  }
}

bool halo_choose_box(int Do, int Ho, int Wo, const int k[3], int box[3]) {
  long best = -1;
  long best_halo = 0;
  for (int tz = 1; tz <= 16 && tz <= Do; ++tz)
    for (int ty = 1; ty <= 64 && ty <= Ho; ++ty) {
      if (tz * ty > 256) break;
      const int txmax = std::min(256 / (tz * ty), Wo);
      for (int tx = 1; tx <= txmax; ++tx) {
        const long halo = (long)(tz + k[0] - 1) * (ty + k[1] - 1) * (tx + k[2] - 1);
        if (halo > kHaloLongRows) break;
        const long boxes = (long)ceil_div(Do, tz) * ceil_div(Ho, ty) * ceil_div(Wo, tx);
        if (best < 0 || boxes < best || (boxes == best && halo < best_halo)) {
          best = boxes; best_halo = halo;
          box[0] = tz; box[1] = ty; box[2] = tx;
        }
      }
    }
  return best > 0;
}

template <typename T, int BN, int WM, int WN>
static int launch_halo_one(const HaloArgs& a, hipStream_t stream) {
  constexpr int BNL = (BN + 63) / 64 * 64;
  constexpr int smem = 2 * kHaloBufBytes + 4 * BNL * kStepRowBytes;
  static bool attr_set = false;
  auto kern = conv_halo_kernel<T, BN, WM, WN>;
  if (!attr_set) {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int grid = a.NBZ * a.NBY * a.NBX * (a.Npad / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

template <typename T>
static int launch_halo_cfg(const HaloArgs& a, TileCfg cfg, hipStream_t stream) {
  switch (cfg) {
    case TILE_256x32: return launch_halo_one<T, 32, 4, 1>(a, stream);
    case TILE_256x64: return launch_halo_one<T, 64, 4, 1>(a, stream);
    case TILE_256x160: return launch_halo_one<T, 160, 4, 1>(a, stream);
    case TILE_256x256: return launch_halo_one<T, 256, 2, 2>(a, stream);
    default: BSMI_FAIL(BSMI_ERR_INVALID, "halo kernel: unsupported tile config %d", (int)cfg);
  }
}

int launch_conv_halo(const HaloArgs& a, int precision, TileCfg cfg, hipStream_t stream) {
  if (a.nsteps <= 0 || a.nphases <= 0 || a.Npad % tile_bn(cfg) != 0 || a.TZ * a.TY * a.TX > 256 || a.hv_long > kHaloLongRows)
    BSMI_FAIL(BSMI_ERR_INVALID, "halo conv launch: bad geometry");
  if (precision == BSMI_PREC_F32) return launch_halo_cfg<float>(a, cfg, stream);
  if (precision == BSMI_PREC_BF16) return launch_halo_cfg<hbf16>(a, cfg, stream);
  BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
}

}  // namespace bsmi
