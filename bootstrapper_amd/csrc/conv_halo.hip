// Halo-tiled implicit-GEMM 3-D convolution on MFMA (gfx950).
//
// Same GEMM as conv_igemm.hip, but the M tile is a 3-D box of output voxels (TZ x TY x TX <= 256)
// and the activation operand is not gathered per kernel tap: for every 32-channel chunk of a
// source tensor the box's input halo ((TZ+kz-1) x (TY+ky-1) x (TX+kx-1) voxels x 32 B) is staged
// ONCE into LDS and all kz*ky*kx taps read their A fragments from it at a per-tap row offset.
// This cuts the activation traffic into the CU by ~10x (the kernel is bound by the L2 -> LDS
// fill rate, not by MFMA issue); the weight tiles still stream through a 4-slot LDS-DMA ring.
//
// LDS: [halo buffer 0 | halo buffer 1 | 3-4 x weight slot (BNL rows x 64 B)].
// A "phase" is one staged halo; a K-step is one kernel tap x 32 channels.  The 1x1x1 residual
// branch uses SHORT phases: the box's own 256 voxels x 32 channels.
// Halo p+2 is issued at the boundary that ends phase p (its buffer is free from then on); the
// host simulates the in-order vmcnt queue and stores, per boundary, which counted wait is safe.
//
// Halo rows are 64 bytes (32 channels) at LDS row jz*PZ + jy*PY + jx; the host picks the pitches
// so that the 16 lanes of a ds_read_b128 lane group fall on rows that are distinct mod 16, which
// makes the (row>>2)&3 chunk swizzle bank-conflict free (halo_choose_geometry).
//
// STATUS (round 1): opt-in (BSMI_USE_HALO=1).  Numerically verified -- every U-Net parity test
// passes on this path -- but not yet faster than the gather kernel: measured on the 128^3 block,
// 7.3 ms vs 6.4 ms on the two largest layers and ~2.4x slower on the short-K layers, where the
// per-tile set-up (halo address tables, full drain before the first K-step) is not amortised.
// Next steps: several boxes per workgroup with the next halo prefetched, cheaper per-read
// addressing, and box shapes that are both coalescing- and bank-friendly.
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

// wait variants of a K-step boundary (HaloStep::wait).  D = NSLOT - 2 weight K-steps may stay in
// flight across a boundary; BI = weight LDS-DMA instructions per wave per K-step, HL / HS =
// instructions per wave of a LONG / SHORT halo.
enum { W_D = 0, W_D_HL = 1, W_D_HS = 2, W_DM1 = 3, W_ALL = 4, W_DM1_HS = 5, W_DM1_HL = 6 };

// x / d for small x, d with magic = ceil(2^32 / d) (0 encodes d == 1)
__device__ __forceinline__ uint32_t udiv_small(uint32_t x, uint32_t magic) { return magic ? __umulhi(x, magic) : x; }

// Stage the halo of phase p: HL (LONG) or HS (SHORT) LDS-DMA instructions per wave.  A free
// force-inlined function rather than a lambda: a lambda with several call sites is kept as an
// out-of-line call whose closure (and every captured array) lives in scratch memory.
template <int HL, int HS, int NW>
__device__ __forceinline__ void halo_issue(cint_ptr_t phases, int p_in, const uint32_t (&l0)[HL], const uint32_t (&l1)[HL],
                                           const uint32_t (&l2)[HL], const uint32_t (&s0)[HS], const uint32_t (&s1)[HS],
                                           const uint32_t (&s2)[HS], uint64_t base0, uint64_t base1, uint64_t base2,
                                           uint32_t hsrc, char* smem, int wave) {
  const int p = __builtin_amdgcn_readfirstlane(p_in);
  const cint_ptr_t d = phases + p * 4;
  const int t = d[0], c0 = d[1], kind = d[2], bufbase = d[3];
  const bool t1 = t == 1, t2 = t == 2;
  const gptr_t hb = (gptr_t)(t1 ? base1 : (t2 ? base2 : base0));
  const lptr_t lh = (lptr_t)(smem + bufbase);
  const uint32_t co = (uint32_t)c0 + hsrc;
  if (kind == 0) {
#pragma unroll
    for (int q = 0; q < HL; ++q) {
      const uint32_t ro = t1 ? l1[q] : (t2 ? l2[q] : l0[q]);
      __builtin_amdgcn_global_load_lds(hb + (size_t)(ro + co), lh + (q * NW + wave) * 1024, 16, 0, 0);
    }
  } else {
#pragma unroll
    for (int q = 0; q < HS; ++q) {
      const uint32_t ro = t1 ? s1[q] : (t2 ? s2[q] : s0[q]);
      __builtin_amdgcn_global_load_lds(hb + (size_t)(ro + co), lh + (q * NW + wave) * 1024, 16, 0, 0);
    }
  }
}

template <typename T, int BN, int WM, int WN, int NSLOT>
__global__ __launch_bounds__(256, 1) void conv_halo_kernel(const HaloArgs a) {
  constexpr int BM = 256, NW = 4;
  static_assert(WM * WN == NW, "one wave per SIMD");
  constexpr int ROWB = kStepRowBytes;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int BNL = (BN + 63) / 64 * 64;
  constexpr int BI = BNL / 16 / NW;
  constexpr int HL = kHaloLongInstr, HS = kHaloShortInstr;
  constexpr int D = NSLOT - 2;
  constexpr int SLOT = BNL * ROWB;
  constexpr int BRING = 2 * kHaloBufBytes;  // LDS offset of the weight ring
  static_assert(D * BI + HL <= 63, "vmcnt range");
  static_assert(D >= 1, "ring depth");

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
  const int TYX = a.TY * a.TX;

  // box row m -> (tz, ty, tx); rows past the box map to row 0 (their results are never stored)
  auto box_coords = [&](int m, int& tz, int& ty, int& tx) __attribute__((always_inline)) {
    m = m < boxrows ? m : 0;
    tz = (int)udiv_small((uint32_t)m, a.mTYX);
    const int mr = m - tz * TYX;
    ty = (int)udiv_small((uint32_t)mr, a.mTX);
    tx = mr - ty * a.TX;
  };

  // ---- per-lane source offsets of the halo rows this lane stages (once per tile) ----------------
  // An LDS-DMA instruction covers 16 halo rows x 64 B: lane -> row (lane>>2), 16-byte slot lane&3,
  // fetching source chunk (lane&3) ^ ((row>>2)&3) (swizzle on the source side).
  // (no references / pointers into the kernel-argument struct: they would force it, and the
  //  arrays below, into scratch memory)
#define BSMI_SRC_OFF(T, Z, Y, X)                                                            \
  ((uint32_t)(min((Z), a.t[T].D - 1) * a.t[T].sz + min((Y), a.t[T].H - 1) * a.t[T].sy + \
              min((X), a.t[T].W - 1) * a.t[T].sx))
  uint32_t lofs0[HL], lofs1[HL], lofs2[HL];
  {
    const int hv = a.HZ * a.PZ;
#pragma unroll
    for (int q = 0; q < HL; ++q) {
      int j = (q * NW + wave) * 16 + (lane >> 2);
      j = j < hv ? j : hv - 1;
      const int jz = (int)udiv_small((uint32_t)j, a.mPZ);
      const int jr = j - jz * a.PZ;
      int jy = (int)udiv_small((uint32_t)jr, a.mPY);
      int jx = jr - jy * a.PY;
      jy = jy < a.HY ? jy : a.HY - 1;  // pitch padding: any valid voxel, never read
      jx = jx < a.HX ? jx : a.HX - 1;
      lofs0[q] = BSMI_SRC_OFF(0, z0 + a.t[0].oz + jz, y0 + a.t[0].oy + jy, x0 + a.t[0].ox + jx);
      lofs1[q] = BSMI_SRC_OFF(1, z0 + a.t[1].oz + jz, y0 + a.t[1].oy + jy, x0 + a.t[1].ox + jx);
      lofs2[q] = BSMI_SRC_OFF(2, z0 + a.t[2].oz + jz, y0 + a.t[2].oy + jy, x0 + a.t[2].ox + jx);
    }
  }
  uint32_t sofs0[HS], sofs1[HS], sofs2[HS];
  {
#pragma unroll
    for (int q = 0; q < HS; ++q) {
      int tz, ty, tx;
      box_coords((q * NW + wave) * 16 + (lane >> 2), tz, ty, tx);
      sofs0[q] = BSMI_SRC_OFF(0, z0 + a.t[0].rz + tz, y0 + a.t[0].ry + ty, x0 + a.t[0].rx + tx);
      sofs1[q] = BSMI_SRC_OFF(1, z0 + a.t[1].rz + tz, y0 + a.t[1].ry + ty, x0 + a.t[1].rx + tx);
      sofs2[q] = BSMI_SRC_OFF(2, z0 + a.t[2].rz + tz, y0 + a.t[2].ry + ty, x0 + a.t[2].rx + tx);
    }
  }
#undef BSMI_SRC_OFF
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;
  const uint32_t hsrc = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);

  auto issue_halo = [&](int p_in) __attribute__((always_inline)) {
    halo_issue<HL, HS, NW>(phases, p_in, lofs0, lofs1, lofs2, sofs0, sofs1, sofs2, base0, base1, base2, hsrc, smem, wave);
  };

  // ---- weight ring ------------------------------------------------------------------------------
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = (lane >> 4) & 3;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)a.Npad * ROWB;
  auto issue_b = [&](int h) __attribute__((always_inline)) {
    const gptr_t wbase = (gptr_t)a.w + (size_t)(h < nsteps ? h : nsteps - 1) * wstep;
    const lptr_t lb = (lptr_t)(smem + BRING + (h % NSLOT) * SLOT);
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
    int tz, ty, tx;
    box_coords(m, tz, ty, tx);
    hrowL[i] = tz * a.PZ + ty * a.PY + tx;
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

  // K-step descriptors are kept as plain wave-uniform scalars (a struct passed through lambdas
  // ends up in scratch memory, whose reloads drain the vmcnt queue of the LDS-DMA loads)
  u32x4_t fa[2][FM], fb[2][FN];
  auto load_frags = [&](int trow, int cbu, bool is_short, int bufbase, int h, int sub, u32x4_t* pa, u32x4_t* pb) __attribute__((always_inline)) {
    const int cb = cbu + lh;
    const char* hbuf = smem + bufbase;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = is_short ? hrowS[i] : hrowL[i] + trow;
      const int chunk = cb ^ ((row >> 2) & 3);
      pa[i] = *(const u32x4_t*)(hbuf + ((uint32_t)row << 6) + (chunk << 4));
    }
    const char* bs = smem + BRING + (h % NSLOT) * SLOT;
    const uint32_t c = 2 * sub + lh;
#pragma unroll
    for (int j = 0; j < FN; ++j) pb[j] = *(const u32x4_t*)(bs + brow[j] + ((c ^ bkey[j]) << 4));
  };
  auto mma = [&](const u32x4_t* pa, const u32x4_t* pb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = HElem<T>::mfma(pa[i], pb[j], acc[i][j]);
  };

  // prologue: halos of phases 0 and 1, weight K-steps 0 .. NSLOT-2; everything must land
  issue_halo(0);
  if (a.nphases > 1) issue_halo(1);
#pragma unroll
  for (int h = 0; h < NSLOT - 1; ++h) issue_b(h);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // descriptor of the current (c_*) and the next (n_*) K-step
  auto clamp_step = [&](int h) __attribute__((always_inline)) { return (h < nsteps ? h : nsteps - 1) * 8; };
  int c_trow0 = steps[0], c_trow1 = steps[1], c_cb = steps[2], c_buf = steps[3], c_wait = steps[4], c_issue = steps[5];
  int n_trow0, n_trow1, n_cb, n_buf, n_wait, n_issue;
  {
    const cint_ptr_t d = steps + clamp_step(1);
    n_trow0 = d[0]; n_trow1 = d[1]; n_cb = d[2]; n_buf = d[3]; n_wait = d[4]; n_issue = d[5];
  }
  load_frags(c_trow0, c_cb & 0xff, (c_cb >> 16) & 1, c_buf, 0, 0, fa[0], fb[0]);
  issue_b(NSLOT - 1);

  for (int h = 0; h < nsteps; ++h) {
    load_frags(c_trow1, (c_cb >> 8) & 0xff, (c_cb >> 16) & 1, c_buf, h, 1, fa[1], fb[1]);
    mma(fa[0], fb[0]);
    // boundary: fa[1]/fb[1] are in registers; weight K-step h+1 (and a halo first used by step
    // h+1) must have landed everywhere.  The host picked the counted wait that guarantees it.
    if (c_wait == W_D)           asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * BI) : "memory");
    else if (c_wait == W_D_HL)   asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * BI + HL) : "memory");
    else if (c_wait == W_D_HS)   asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * BI + HS) : "memory");
    else if (c_wait == W_DM1)    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * BI) : "memory");
    else if (c_wait == W_DM1_HS) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * BI + HS) : "memory");
    else if (c_wait == W_DM1_HL) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * BI + HL) : "memory");
    else                         asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c_issue >= 0) issue_halo(c_issue);
    issue_b(h + NSLOT);
    const cint_ptr_t d2 = steps + clamp_step(h + 2);
    const int f_trow0 = d2[0], f_trow1 = d2[1], f_cb = d2[2], f_buf = d2[3], f_wait = d2[4], f_issue = d2[5];
    load_frags(n_trow0, n_cb & 0xff, (n_cb >> 16) & 1, n_buf, h + 1, 0, fa[0], fb[0]);
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
    c_trow0 = n_trow0; c_trow1 = n_trow1; c_cb = n_cb; c_buf = n_buf; c_wait = n_wait; c_issue = n_issue;
    n_trow0 = f_trow0; n_trow1 = f_trow1; n_cb = f_cb; n_buf = f_buf; n_wait = f_wait; n_issue = f_issue;
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
      int tz, ty, tx;
      box_coords(m, tz, ty, tx);
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

// number of extra LDS cycles (bank conflicts) of the A-fragment reads of one K-step for a given
// box / pitch: for every 32-row fragment and every ds_read_b128 lane group count how many of the
// 16 rows coincide mod 16 (64-byte rows, chunk swizzle (row>>2)&3)
static int halo_conflicts(const int box[3], int PZ, int PY, int wm_count) {
  static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  const int boxrows = box[0] * box[1] * box[2];
  int cost = 0;
  for (int f = 0; f < 256 / 32; ++f)
    for (int g = 0; g < 2; ++g) {
      int seen[16] = {0};
      for (int l = 0; l < 16; ++l) {
        int m = f * 32 + groups[g][l];
        if (m >= boxrows) m = 0;
        const int tz = m / (box[1] * box[2]), mr = m % (box[1] * box[2]);
        const int row = tz * PZ + (mr / box[2]) * PY + mr % box[2];
        cost += seen[row & 15]++;
      }
    }
  (void)wm_count;
  return cost;
}

bool halo_choose_geometry(int Do, int Ho, int Wo, const int k[3], int wm, int box[3], int pitch[2]) {
  // score = number of boxes (tile padding waste) inflated by the LDS bank-conflict cost of the
  // A-fragment reads: a conflict-free box that needs a few more tiles beats a conflicting one
  double best_score = -1;
  for (int tz = 1; tz <= 16 && tz <= Do; ++tz)
    for (int ty = 1; ty <= 64 && ty <= Ho; ++ty) {
      if (tz * ty > 256) break;
      const int txmax = std::min(256 / (tz * ty), Wo);
      for (int tx = 1; tx <= txmax; ++tx) {
        const int hz = tz + k[0] - 1, hy = ty + k[1] - 1, hx = tx + k[2] - 1;
        if ((long)hz * hy * hx > kHaloRows) break;
        const long boxes = (long)ceil_div(Do, tz) * ceil_div(Ho, ty) * ceil_div(Wo, tx);
        if (best_score >= 0 && (double)boxes > best_score) continue;  // cannot win even conflict free
        const int b[3] = {tz, ty, tx};
        int c_best = -1, py_best = 0, pz_best = 0;
        for (int py = hx; py < hx + 16; ++py)
          for (int pz = hy * py; pz < hy * py + 16; ++pz) {
            if ((long)hz * pz > kHaloRows) break;
            const int c = halo_conflicts(b, pz, py, wm);
            if (c_best < 0 || c < c_best) { c_best = c; py_best = py; pz_best = pz; }
          }
        if (c_best < 0) continue;
        // 256 lane reads per K-step half; every extra conflict cycle costs about as much as a read
        const double score = (double)boxes * (1.0 + 0.5 * c_best / 128.0);
        if (best_score < 0 || score < best_score) {
          best_score = score;
          box[0] = tz; box[1] = ty; box[2] = tx;
          pitch[0] = pz_best; pitch[1] = py_best;
        }
      }
    }
  return best_score > 0;
}

template <typename T, int BN, int WM, int WN, int NSLOT>
static int launch_halo_one(const HaloArgs& a, hipStream_t stream) {
  constexpr int BNL = (BN + 63) / 64 * 64;
  constexpr int smem = 2 * kHaloBufBytes + NSLOT * BNL * kStepRowBytes;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = conv_halo_kernel<T, BN, WM, WN, NSLOT>;
  if (!attr_set) {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int grid = a.NBZ * a.NBY * a.NBX * (a.Npad / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int halo_ring_slots(TileCfg cfg) { return cfg == TILE_256x256 ? 3 : 4; }

template <typename T>
static int launch_halo_cfg(const HaloArgs& a, TileCfg cfg, hipStream_t stream) {
  switch (cfg) {
    case TILE_256x32: return launch_halo_one<T, 32, 4, 1, 4>(a, stream);
    case TILE_256x64: return launch_halo_one<T, 64, 4, 1, 4>(a, stream);
    case TILE_256x160: return launch_halo_one<T, 160, 4, 1, 4>(a, stream);
    case TILE_256x256: return launch_halo_one<T, 256, 2, 2, 3>(a, stream);
    default: BSMI_FAIL(BSMI_ERR_INVALID, "halo kernel: unsupported tile config %d", (int)cfg);
  }
}

int launch_conv_halo(const HaloArgs& a, int precision, TileCfg cfg, hipStream_t stream) {
  if (a.nsteps <= 0 || a.nphases <= 0 || a.Npad % tile_bn(cfg) != 0 || a.TZ * a.TY * a.TX > 256 || a.HZ * a.PZ > kHaloRows)
    BSMI_FAIL(BSMI_ERR_INVALID, "halo conv launch: bad geometry");
  if (precision == BSMI_PREC_F32) return launch_halo_cfg<float>(a, cfg, stream);
  if (precision == BSMI_PREC_BF16) return launch_halo_cfg<hbf16>(a, cfg, stream);
  BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
}

}  // namespace bsmi
