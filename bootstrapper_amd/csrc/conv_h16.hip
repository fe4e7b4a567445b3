// Halo-resident implicit-GEMM 3-D "valid" convolution for the stages with at most 64 output channels, split-bf16 mode (gfx950).
//
// Why another form.  The gather kernel (conv_igemm.hip conv_x3_body, 256 x 64 tile, 64 x 64 wave tiles) runs these stages at
// 0.05-0.37 of the split-bf16 ceiling.  A dev build that staged the activation rows of only one K-step in nine
// (-DBSMI_DBG_SKIP_A) still stopped at 0.42: the loop has two barriers per K-step and 48 MFMAs per wave between them, and the
// first fused raster-halo kernel (conv_rh.hip, 32 x 64 wave tiles, 24 MFMAs per barrier) lost to it for the same reason.  What
// the 256-column tiles have and these do not is work per barrier: 96-120 MFMAs per wave and K-step, 0.75 MFMA-busy.
//
// This kernel gives the narrow stages that ratio.  Four waves, each with a 128-row x BN-column accumulator (8 row blocks:
// 96 MFMAs per K-step at BN = 64), two workgroups per CU.  512 rows of staging per K-step do not fit twice into half a CU's
// LDS as a gather; as a halo they do: the M dimension runs over the rows of the stage's INPUT raster (conv_rh.hip), where tap
// (dy, dx) of row q is row q + dy * Win + dx, so the tile's rows plus the 2 Win + 2 behind them are staged ONCE per (source
// tensor, 16-channel chunk, z tap) -- 32 bytes per row and plane, 48 KB for 768 rows -- and every in-plane tap reads its A
// fragments from there.  16-channel chunks because 32 would not leave room for two workgroups; K = 32 of the MFMA is then two
// taps x 16 channels: lanes 0-31 of the A fragment read tap a's rows, lanes 32-63 tap b's (nine taps: four pairs and a K-step
// whose second half meets zero weights, 10 % of the MFMAs).  The halo is single-buffered: while one workgroup of a CU waits
// for its next phase's rows the other multiplies.  Weights: one K-step (BN rows x 64 B per plane) a step ahead, ONE barrier
// per K-step.
//
// LDS: [halo: HR rows x 64 B][weights: 2 slots x (hi, lo) x BN rows x 64 B].  A halo row is the voxel's 64
// bytes as they lie in the tensor -- 16-byte slots (hi c0-7, lo c0-7, hi c8-15, lo c8-15) -- so one LDS-DMA instruction
// stages 16 whole rows, one L1 access per row (two planes of 32-byte rows, the first version, took two accesses per row and
// 2.0-2.3 us per phase: the halo loads run at the L1's access rate, the same plateau the gather kernels sit on).  Slot s of
// row r lands at slot s ^ ((r >> 2) & 1): a ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) takes rows {0-3, 12-15} of one
// slot and rows {4-11} of the slot two further, and with that key its 16 lanes hit 16 different slots of the 256-byte bank
// row at ANY row offset (found by search over the 4-periodic keys; the plain layout is 2-way).  The weight slots keep the
// chunk swizzle of conv_igemm.hip.  (Tried: the weight fragments straight from global memory into registers, a K-step ahead,
// no ring and no barrier inside a phase -- slower at 16 columns (72 -> 12 channels 0.56 -> 0.61 ms) and at 64 (360 -> 60:
// 2.19 -> 2.55): whatever goes through the L1 competes with the halo loads.  Also tried: one dword per row of the NEXT phase requested a phase ahead, to have its lines in
// L2 when the LDS-DMA loads go out -- slower, 360 -> 60 channels 2.23 -> 2.38 ms, 72 -> 12 0.71 -> 0.97: the extra L1 accesses
// cost more than the shorter round trip saves.)
#include "conv_h16.h"

#include <algorithm>

#include "conv_dev.h"

namespace bsmi {

#ifdef BSMI_STAMP  // dev build: where a tile's time goes, 100 MHz ticks of wave 0 summed over tiles:
// [0] halo issue + wait + barrier, [1] K-steps up to their last MFMA issue, [2] end-of-step wait + barrier, [3] tiles, [4] phases,
// [5] K-steps, [6] prologue, [7] epilogue
__device__ unsigned long long g_h16_stamp[8];
extern "C" int bsmi_debug_stamps_h16(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h16_stamp), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_h16_stamp), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#define H16_STAMP(var_) const unsigned long long var_ = wall_clock64()
#else
#define H16_STAMP(var_)
#endif

namespace {

constexpr int kNW = 4;  // waves per workgroup
constexpr int kR = 8;   // 16-row blocks per wave

template <int FNB, int HR>
__global__ __launch_bounds__(64 * kNW, 2) void conv_h16_kernel(const H16Args a) {
  using T = bf16f_elem;
  constexpr int BN = 16 * FNB, BM = kH16TileRows;
  static_assert(BM == kNW * kR * 16, "tile rows");
  constexpr int HALO = HR * 64;        // bytes of the halo
  constexpr int OFF_B = HALO;
  constexpr int EB = BN * 64;          // one weight plane of one K-step
  constexpr int SLOT = 2 * EB;         // hi, lo
  constexpr int JW = HR / 16 / kNW;    // halo LDS-DMA instructions per wave (16 rows each)
  static_assert(HR % (16 * kNW) == 0, "the waves share the halo rows evenly");
  constexpr int G = FNB >= 4 ? 1 : (FNB == 2 ? 2 : 8);  // row blocks multiplied side by side: >= 4 MFMAs between two on one accumulator
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous tile order (conv_igemm_kernel): neighbouring tiles share halo rows through one L2
  const int ntiles = a.ntiles;
  const int qn = ntiles >> 3, rn = ntiles & 7;
  const int xcd = blockIdx.x & 7, jn = blockIdx.x >> 3;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jn;
  // Balanced tiles (h16_tiling): the first n_big tiles have 8 row blocks per wave, the others r_small, chosen so that the
  // launch is a whole number of rounds of the resident workgroups
  const int R = tile < a.n_big ? kR : a.r_small;
  const int q0 = tile < a.n_big ? tile * BM : a.n_big * BM + (tile - a.n_big) * a.r_small * (kNW * 16);
  const int cut = (kR - R) * (kNW * 16);  // rows this tile is short of 512: as many halo rows less
  H16_STAMP(st_begin);
#ifdef BSMI_STAMP
  unsigned long long acc_halo = 0, acc_k = 0, acc_w = 0;
#endif
  const cint_ptr_t phases = (cint_ptr_t)a.phases;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;
  const int nsteps = a.nsteps, nphases = a.nphases;

  // the halo rows this lane stages (row 16 j + lane / 4 of the wave's j-th instruction), as (z, y, x) of the input raster
  uint32_t zyx[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    int q = q0 + 16 * (jj * kNW + wave) + (lane >> 2);
    q = q < a.Q ? q : a.Q - 1;
    const int xx = q % a.Win;
    const int zy = q / a.Win;
    const int yy = zy % a.Hin, z = zy / a.Hin;
    zyx[jj] = ((uint32_t)z << 22) | ((uint32_t)yy << 11) | (uint32_t)xx;
  }
  // lane l lands at row l >> 2, slot l & 3 of its KiB and fetches the source slot (l & 3) ^ key(row); 16 rows = 4 key periods
  const uint32_t half_src = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 1)) << 4);
  static_assert(kMaxConvTensors == 3, "three source slots");
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;

  // weights: lane l lands at row l >> 2, 16-byte slot l & 3 of its KiB and fetches the source chunk (l & 3) ^ key(row)
  const uint32_t bsrc = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ swz16((lane >> 4) & 3)) << 4));
  const gptr_t w_hi = (gptr_t)a.w, w_lo = (gptr_t)a.w_lo;
  const size_t wstep = (size_t)a.Npad * 64;

#define H16_ISSUE_B(s_)                                                                                              \
  do {                                                                                                                \
    _Pragma("unroll") for (int pp_ = 0; pp_ < 2; ++pp_) {                                                             \
      const int p_ = wave + 4 * pp_;                                                                                  \
      if (p_ < 2 * FNB) {                                                                                             \
        const int plane_ = p_ / FNB, blk_ = p_ - plane_ * FNB;                                                        \
        const gptr_t src_ = (plane_ ? w_lo : w_hi) + (size_t)(s_) * wstep + (size_t)(blk_ * 1024) + bsrc;            \
        const lptr_t dst_ = (lptr_t)(smem + OFF_B + ((s_) & 1) * SLOT + plane_ * EB + blk_ * 1024);                   \
        __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                                                       \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)

  f32x4_t acc[kR][FNB];
#pragma unroll
  for (int i = 0; i < kR; ++i)
#pragma unroll
    for (int j = 0; j < FNB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const uint32_t arow0 = (uint32_t)(wave * (R * 16) + lr);
  const uint32_t aslot = (uint32_t)((lq & 1) << 1);  // logical slot of the hi vector: channels 0-7 or 8-15
  const bool tap_b = (lq >> 1) != 0;
  uint32_t boff[FNB];
#pragma unroll
  for (int j = 0; j < FNB; ++j) boff[j] = (uint32_t)((j * 16 + lr) * 64 + ((lq ^ swz16((lr >> 2) & 3)) << 4));

  H16_ISSUE_B(0);
  H16_STAMP(st_loop);
  int s = 0;
  for (int p = 0; p < nphases; ++p) {
    H16_STAMP(st_p0);
    const cint_ptr_t d = phases + p * 4;
    const int t = d[0], delta = d[1], rows = d[2], ns = d[3];
    // the phase's halo: everybody left the previous one at the last barrier
    {
      const bool t1 = t == 1, t2 = t == 2;
      const gptr_t hb = (gptr_t)(t1 ? base1 : (t2 ? base2 : base0));
      const int sz = t1 ? a.t[1].sz : (t2 ? a.t[2].sz : a.t[0].sz);
      const int sy = t1 ? a.t[1].sy : (t2 ? a.t[2].sy : a.t[0].sy);
      const int sx = t1 ? a.t[1].sx : (t2 ? a.t[2].sx : a.t[0].sx);
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) {
        const uint32_t v = zyx[jj];
        const uint32_t src = (v >> 22) * (uint32_t)sz + ((v >> 11) & 2047u) * (uint32_t)sy + (v & 2047u) * (uint32_t)sx + (uint32_t)delta + half_src;
        if (16 * (jj * kNW + wave) < rows - cut)  // (rows behind the last tap's reach are never read)
          __builtin_amdgcn_global_load_lds(hb + (size_t)src, (lptr_t)(smem + (jj * kNW + wave) * 1024), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef BSMI_STAMP
    acc_halo += wall_clock64() - st_p0;
#endif
    for (int k = 0; k < ns; ++k, ++s) {
      H16_STAMP(st_k0);
      if (s + 1 < nsteps) H16_ISSUE_B(s + 1);
      const cint_ptr_t ds = steps + s * 4;
      const int offa = ds[0], offb = ds[1];
      const char* slot = smem + OFF_B + (s & 1) * SLOT;
      u32x4_t BH[FNB], BL[FNB];
#pragma unroll
      for (int j = 0; j < FNB; ++j) {
        BH[j] = *(const u32x4_t*)(slot + boff[j]);
        BL[j] = *(const u32x4_t*)(slot + EB + boff[j]);
      }
      const uint32_t ar = arow0 + (uint32_t)(tap_b ? offb : offa);
      const char* ap = smem + ar * 64 + ((aslot ^ ((ar >> 2) & 1)) << 4);  // the hi vector; its lo vector is the slot beside it (^ 16)
      const char* apl = smem + ar * 64 + ((aslot ^ ((ar >> 2) & 1) ^ 1) << 4);
#pragma unroll
      for (int i0 = 0; i0 < kR; i0 += G) {
        if (i0 >= R) break;  // (a short tile; with G > 1 its last group multiplies a few rows nobody stores)
        u32x4_t AH[G], AL[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          AH[g] = *(const u32x4_t*)(ap + (i0 + g) * 1024);
          AL[g] = *(const u32x4_t*)(apl + (i0 + g) * 1024);
        }
        // A lo x B hi, A hi x B hi, A hi x B lo (the order of conv_x3_body)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < FNB; ++j)
            acc[i0 + g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, AL[g]), __builtin_bit_cast(bf16x8_t, BH[j]), acc[i0 + g][j], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < FNB; ++j)
            acc[i0 + g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, AH[g]), __builtin_bit_cast(bf16x8_t, BH[j]), acc[i0 + g][j], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < FNB; ++j)
            acc[i0 + g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, AH[g]), __builtin_bit_cast(bf16x8_t, BL[j]), acc[i0 + g][j], 0, 0, 0);
      }
      // the next K-step's weights have landed; nobody still reads this step's slot or (at a phase's end) the halo
      H16_STAMP(st_k1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef BSMI_STAMP
      {
        const unsigned long long st_k2 = wall_clock64();
        acc_k += st_k1 - st_k0;
        acc_w += st_k2 - st_k1;
      }
#endif
    }
  }
#undef H16_ISSUE_B
  H16_STAMP(st_epi);

  // epilogue: bias (+ReLU), (hi, lo) through per-wave LDS strips (in the halo area), 16-byte streaming stores; raster rows that
  // are no output voxel (xx >= Wo, yy >= Ho) are dropped
  constexpr int PITCH = BN * 2 + 16;
  constexpr int CPR = BN * 2 / 16;
  constexpr int NCH = 16 * CPR;
  constexpr int LO_STRIPS = kNW * 16 * PITCH;
  static_assert(2 * LO_STRIPS <= HALO, "strips fit in the halo area");
  char* strip = smem + wave * (16 * PITCH);
  T* out = (T*)a.out;
  float bv[FNB];
#pragma unroll
  for (int j = 0; j < FNB; ++j) bv[j] = a.bias[j * 16 + lr];
#pragma unroll
  for (int i = 0; i < kR; ++i) {
    if (i >= R) break;
#pragma unroll
    for (int j = 0; j < FNB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[j];  // 16x16: register r holds row 4 (lane >> 4) + r, column lane & 15
        if (a.relu) v = v > 0.f ? v : 0.f;
        Elem<T>::store((T*)(strip + (4 * lq + r) * PITCH) + j * 16 + lr, v);
        Elem<T>::store((T*)(strip + LO_STRIPS + (4 * lq + r) * PITCH) + j * 16 + lr, split_lo(v));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < (NCH + 63) / 64; ++k) {
      const int c = lane + 64 * k;
      if (c >= NCH) break;
      const int row = c / CPR, cc = c - row * CPR;
      const int q = q0 + wave * (R * 16) + i * 16 + row;
      const int n = cc * 8;
      if (q < a.Q && n < a.Co) {
        const int xx = q % a.Win;
        const int zy = q / a.Win;
        const int yy = zy % a.Hin, z = zy / a.Hin;
        if (xx < a.Wo && yy < a.Ho) {
          T* dst = out + act_index<T>(((size_t)(z * a.Ho + yy) * a.Wo + xx) * a.Co, n);
          store_stream16(dst, *(const u32x4_t*)(strip + row * PITCH + cc * 16));
          store_stream16(dst + kSplitLoElems, *(const u32x4_t*)(strip + LO_STRIPS + row * PITCH + cc * 16));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#ifdef BSMI_STAMP
  if (tid == 0) {
    const unsigned long long st_end = wall_clock64();
    atomicAdd(&g_h16_stamp[0], acc_halo);
    atomicAdd(&g_h16_stamp[1], acc_k);
    atomicAdd(&g_h16_stamp[2], acc_w);
    atomicAdd(&g_h16_stamp[3], 1ull);
    atomicAdd(&g_h16_stamp[4], (unsigned long long)nphases);
    atomicAdd(&g_h16_stamp[5], (unsigned long long)nsteps);
    atomicAdd(&g_h16_stamp[6], st_loop - st_begin);
    atomicAdd(&g_h16_stamp[7], st_end - st_epi);
  }
#endif
}

template <int FNB, int HR>
int launch_t(const H16Args& a, hipStream_t stream) {
  constexpr int smem = HR * 64 + 4 * 16 * FNB * 64;  // halo, two weight slots
  static_assert(2 * smem <= 160 * 1024, "two workgroups per CU (three at 16 columns and 768 rows)");
  auto kern = conv_h16_kernel<FNB, HR>;
  static DeviceOnce once;
  const int rc_once = once.run([&]() -> int {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipFuncAttributes fa;
    BSMI_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    // scratch traffic counts on vmcnt like the LDS-DMA loads the loop waits for: refuse a kernel that spills
    if (fa.localSizeBytes != 0) BSMI_FAIL(BSMI_ERR_STATE, "halo conv kernel (%d columns) was compiled with %zu bytes of scratch", 16 * FNB, (size_t)fa.localSizeBytes);
    return BSMI_OK;
  });
  if (rc_once) return rc_once;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * kNW), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

}  // namespace

int h16_halo_rows(int Win, int ky, int kx, int npad, int* max_r) {
  if (npad != 16 && npad != 64) return 0;
  const int reach = (ky - 1) * Win + (kx - 1);
  *max_r = kR;
  if (kH16TileRows + reach <= 768) return 768;
  // 16 columns: 768 rows of halo leave room for three workgroups per CU (52 KB each), 1024 rows for two -- tiles of 7 row
  // blocks per wave where those fit the smaller buffer (Win = 130: 448 + 262 rows)
  if (npad == 16 && (kR - 1) * kNW * 16 + reach <= 768) {
    *max_r = kR - 1;
    return 768;
  }
  if (kH16TileRows + reach <= 1024 && npad == 16) return 1024;  // (at 64 columns: 64 KB of halo + 16 KB of weights, more than half a CU's LDS)
  return 0;
}

void h16_tiling(int64_t Q, int npad, int halo_rows, int max_r, int n_cus, int* ntiles, int* n_big, int* r_small) {
  const int64_t big = kH16TileRows, unit = kNW * 16;
  if (max_r < kR) {  // every tile max_r row blocks per wave
    *ntiles = (int)((Q + max_r * unit - 1) / (max_r * unit));
    *n_big = 0;
    *r_small = max_r;
    return;
  }
  const int64_t nt = (Q + big - 1) / big;
  *ntiles = (int)nt;
  *n_big = (int)nt;
  *r_small = kR;
  // workgroups resident per CU: by LDS (launch_t's byte count; 160 KB per CU), at most two waves per SIMD's worth at 64 columns
  const int64_t lds = (int64_t)halo_rows * 64 + 4 * npad * 64;
  const int64_t per_cu = std::min<int64_t>(npad == 64 ? 2 : 3, (160 * 1024) / lds);
  const int64_t slots = per_cu * n_cus;
  if (slots <= 0 || nt % slots == 0) return;
  // The launch would run ceil(nt / slots) rounds with the last one partly empty (the 360 -> 60 channel stage: 1 357 tiles on
  // 512 slots, three rounds for 2.65 rounds of work).  The same number of rounds, all full, of shorter tiles: n_big of 512
  // rows, the rest r_small row blocks per wave, r_small + 1 = 8 -- only if that is at least 6 (a shorter tile has fewer MFMAs
  // per barrier and a larger share of halo rows).
  const int64_t rounds = (nt + slots - 1) / slots, total = rounds * slots;
  // (measured: with seven rounds or more the shorter tiles cost more than the part-empty last round -- 60 -> 60 channels,
  // 6.3 rounds: 0.90 -> 0.96 ms; 360 -> 60, 2.65 rounds: 2.27 -> 2.13 ms)
  if (rounds > 4) return;
  const int64_t r = (Q + total * unit - 1) / (total * unit);  // row blocks per wave if `total` equal tiles shared the rows
  if (r < 6 || r > kR) return;
  if (r < kR) {  // equal tiles of r blocks: at most `total` of them
    const int64_t rows = r * unit;
    *ntiles = (int)((Q + rows - 1) / rows);
    *n_big = 0;
    *r_small = (int)r;
    return;
  }
  // between 7 and 8 blocks: n_big * 512 + (total - n_big) * 448 >= Q
  const int64_t small = big - unit;
  int64_t nb = (Q - total * small + unit - 1) / unit;
  if (nb < 0) nb = 0;
  *ntiles = (int)total;
  *n_big = (int)nb;
  *r_small = kR - 1;
}

int launch_conv_h16(const H16Args& a, int halo_rows, hipStream_t stream) {
  if (a.Q <= 0 || a.nsteps <= 0 || a.nphases <= 0 || a.Hin <= 0 || a.Win <= 0 || a.Hin > 2047 || a.Win > 2047 || a.Do > 1023)
    BSMI_FAIL(BSMI_ERR_INVALID, "halo conv: bad geometry Q=%d steps=%d phases=%d raster %d x %d", a.Q, a.nsteps, a.nphases, a.Hin, a.Win);
  if (a.ntiles <= 0 || a.n_big < 0 || a.n_big > a.ntiles || a.r_small < 1 || a.r_small > kR ||
      (int64_t)a.n_big * kH16TileRows + (int64_t)(a.ntiles - a.n_big) * a.r_small * (kNW * 16) < a.Q)
    BSMI_FAIL(BSMI_ERR_INVALID, "halo conv: tiling %d tiles (%d of 512 rows, the others %d blocks per wave) does not cover %d rows", a.ntiles, a.n_big,
              a.r_small, a.Q);
  if (a.Npad == 64 && halo_rows == 768) return launch_t<4, 768>(a, stream);
  if (a.Npad == 16 && halo_rows == 768) return launch_t<1, 768>(a, stream);
  if (a.Npad == 16 && halo_rows == 1024) return launch_t<1, 1024>(a, stream);
  BSMI_FAIL(BSMI_ERR_INVALID, "halo conv: no kernel for %d columns with %d halo rows", a.Npad, halo_rows);
}

}  // namespace bsmi
