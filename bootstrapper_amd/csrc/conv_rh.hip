// Raster-halo implicit-GEMM 3-D "valid" convolution for gfx950 MFMA.
//
// Same GEMM as conv_igemm.hip (one launch per ConvPass stage, residual and concat folded in as
// K-steps, bias + ReLU epilogue), but the activation operand is not gathered tap by tap.  The M
// dimension runs over the rows q = (z * Hin + yy) * Win + xx of the stage's INPUT raster (Hin =
// Ho + ky - 1, Win = Wo + kx - 1; rows with yy >= Ho or xx >= Wo are computed and dropped:
// 1 - Ho*Wo / (Hin*Win) of the work, 2-6 %).  In that raster the in-plane tap (dy, dx) of row q
// is simply row q + dy * Win + dx, for every row of the tile.  So for each (source tensor,
// 32-channel chunk, z-tap) -- a "phase" -- the tile's 256 rows and the (ky-1)*Win + (kx-1) rows
// after them are staged ONCE into an LDS halo buffer (64 bytes per row, same chunk swizzle as
// the weight tiles, which is invariant under a row shift), and the ky*kx taps of the phase read
// their A fragments from it at a row offset.  Against the tap-by-tap gather this divides the
// activation bytes that enter the CU -- and the LDS-DMA instructions the waves must issue, which
// is what bounds the small-Cout layers -- by up to ky*kx * 256 / (256 + 2*Win + 2).
//
// LDS: [halo 0 | halo 1 | ring of 4 weight slots (BN rows x 64 B)].  The halo of the next phase
// is staged into the other buffer while the current phase is multiplied; the host simulates the
// in-order vmcnt queue and stores with every K-step how many weight groups and halos may still
// be in flight at its barrier (RhStep::wait).  Eight waves (two per SIMD), early / late
// staggered issue as in conv_igemm.hip; persistent form with per-XCD work queues and split-K
// tail for the big tiles.
#include "conv_rh.h"

#include <cstdlib>

#include "conv_dev.h"

namespace bsmi {

constexpr int kRhNW = 8;  // waves per workgroup

template <int HP>
__device__ __forceinline__ void rh_issue_halo(cint_ptr_t phases, int p_in, const uint32_t (&ro0)[HP], const uint32_t (&ro1)[HP],
                                              const uint32_t (&ro2)[HP], uint64_t base0, uint64_t base1, uint64_t base2,
                                              uint32_t hsrc, char* smem, int wave) {
  constexpr int HBYTES = HP * kRhNW * 16 * kStepRowBytes;
  const int p = __builtin_amdgcn_readfirstlane(p_in);
  const cint_ptr_t d = phases + p * 4;
  const int t = d[0], delta = d[1], buf = d[2];
  const bool t1 = t == 1, t2 = t == 2;
  const gptr_t hb = (gptr_t)(t1 ? base1 : (t2 ? base2 : base0));
  const lptr_t lh = (lptr_t)(smem + buf * HBYTES);
  const uint32_t co = (uint32_t)delta + hsrc;
#pragma unroll
  for (int q = 0; q < HP; ++q) {
    const uint32_t ro = t1 ? ro1[q] : (t2 ? ro2[q] : ro0[q]);
    __builtin_amdgcn_global_load_lds(hb + (size_t)(ro + co), lh + (q * kRhNW + wave) * 1024, 16, 0, 0);
  }
}

// counted wait of a K-step barrier: `code` = a * 3 + b -> a weight groups (G pieces each) and b
// halos (HP pieces each) may stay in flight
template <int G, int HP>
__device__ __forceinline__ void rh_wait(int code) {
  static_assert(2 * G + 2 * HP <= 63, "vmcnt range");
#define RH_W(c_, n_) case c_: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n_) : "memory"); break;
  switch (code) {
    RH_W(0, 0) RH_W(1, HP) RH_W(2, 2 * HP)
    RH_W(3, G) RH_W(4, G + HP) RH_W(5, G + 2 * HP)
    RH_W(6, 2 * G) RH_W(7, 2 * G + HP)
    default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * G + 2 * HP) : "memory"); break;
  }
#undef RH_W
}

template <typename T, int BN, int WM, int WN, int HP, int B_INSTR, bool LATE>
__device__ __forceinline__ void conv_rh_body(const RhArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  constexpr int BM = 256, NW = kRhNW;
  static_assert(WM * WN == NW, "two waves per SIMD");
  constexpr int ROWB = kStepRowBytes;
  constexpr int NSLOT = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int G = B_INSTR;
  constexpr int HBYTES = HP * NW * 16 * ROWB;
  constexpr int RING = 2 * HBYTES;
  constexpr int SLOT = BN * ROWB;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;
  const cint_ptr_t phases = (cint_ptr_t)a.phases;
  const int nsteps = a.nsteps;
  const int nloc = s1 - s0;
  const int ntn = a.Npad / BN;
  const int tile_m = tile / ntn, tile_n = tile - tile_m * ntn;
  const int q0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = (lane >> 4) & 3;
  // source chunk that lands in LDS slot lchunk; split tensors hold (hi, lo) vectors interleaved: the plane's chunks are 32
  // bytes apart and the phase's byte offset (RhPhase::delta) selects hi (+0) or lo (+16)
  const uint32_t hsrc = (uint32_t)((lchunk ^ skey) << (IsSplit<T>::value ? 5 : 4));
  static_assert(kMaxConvTensors == 3, "three source slots");
  uint32_t ro0[HP], ro1[HP], ro2[HP];
#pragma unroll
  for (int i = 0; i < HP; ++i) {
    int q = q0 + (i * NW + wave) * 16 + lrow;
    q = q < a.Q ? q : a.Q - 1;
    const int xx = q % a.Win;
    const int zy = q / a.Win;
    const int yy = zy % a.Hin, z = zy / a.Hin;
    ro0[i] = (uint32_t)(z * a.t[0].sz + yy * a.t[0].sy + xx * a.t[0].sx);
    ro1[i] = (uint32_t)(z * a.t[1].sz + yy * a.t[1].sy + xx * a.t[1].sx);
    ro2[i] = (uint32_t)(z * a.t[2].sz + yy * a.t[2].sy + xx * a.t[2].sx);
  }
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB) + (uint32_t)((lchunk ^ skey) << 4);
  const size_t wstep = (size_t)a.Npad * ROWB;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct Desc { int rowoff, bufphase, wait, issue; };
  auto fetch = [&](int h) -> Desc {  // h relative to s0; past the end: the last K-step again
    const int ha = s0 + h;
    const cint_ptr_t d = steps + (ha < nsteps ? ha : nsteps - 1) * 4;
    return Desc{d[0], d[1], d[2], d[3]};
  };
  auto issue_w = [&](int h) {
    const gptr_t wbase = (gptr_t)a.w + (size_t)(s0 + h < nsteps ? s0 + h : nsteps - 1) * wstep;
    const lptr_t lb = (lptr_t)(smem + RING + (h & (NSLOT - 1)) * SLOT);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(wbase + (size_t)i * NW * 16 * ROWB + offb, lb + (i * NW + wave) * 1024, 16, 0, 0);
  };
#define RH_HALO(p_) rh_issue_halo<HP>(phases, (p_), ro0, ro1, ro2, base0, base1, base2, hsrc, smem, wave)

  const int lr = lane & 31, lh = lane >> 5;
  uint32_t arow[FM], brow[FN], bkey[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) arow[i] = wm * WTM + i * 32 + lr;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * WTN + j * 32 + lr;
    brow[j] = RING + row * ROWB;
    bkey[j] = (row >> 2) & 3;
  }

  u32x4_t fa[2][FM], fb[2][FN];
  auto load_frags = [&](int slot, const Desc& d, int sub, u32x4_t* pa, u32x4_t* pb) {
    const uint32_t c = 2 * sub + lh;
    const char* hb = smem + (d.bufphase & 1) * HBYTES;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const uint32_t t = arow[i] + (uint32_t)d.rowoff;
      pa[i] = *(const u32x4_t*)(hb + (t << 6) + ((c ^ ((t >> 2) & 3)) << 4));
    }
    const char* st = smem + slot * SLOT;
#pragma unroll
    for (int j = 0; j < FN; ++j) pb[j] = *(const u32x4_t*)(st + brow[j] + ((c ^ bkey[j]) << 4));
  };
  auto mma = [&](const u32x4_t* pa, const u32x4_t* pb) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = Elem<T>::mfma(pa[i], pb[j], acc[i][j]);
  };

  // prologue: the halo of the first phase (and of the next one if its staging slot lies before
  // s0), weight K-steps 0..2.  Everything older than the last two weight groups has landed
  // after the wait, halos included (in-order return).
  Desc d = fetch(0);
  {
    const int p0 = d.bufphase >> 8;
    RH_HALO(p0);
    if (p0 + 1 < a.nphases && phases[(p0 + 1) * 4 + 3] < s0) RH_HALO(p0 + 1);
  }
  issue_w(0);
  issue_w(1);
  issue_w(2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
  __builtin_amdgcn_s_barrier();
  load_frags(0, d, 0, fa[0], fb[0]);
  if constexpr (!LATE) issue_w(3);
  Desc dn = fetch(1);
  int prev_issue = -1;  // RhStep::issue of the previous K-step (LATE waves stage half a K-step later)
  const bool cold = s0 != 0;  // split-K part: the steady-state wait codes hold from local K-step 4 on

  for (int h = 0; h < nloc; ++h) {
    if constexpr (LATE) {
      if (prev_issue >= 0) RH_HALO(prev_issue);
      issue_w(h + 3);
    }
    load_frags(h & (NSLOT - 1), d, 1, fa[1], fb[1]);
    mma(fa[0], fb[0]);
#ifndef BSMI_NO_SCHED_HINTS
    if constexpr (LATE) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < FM + FN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#endif
    rh_wait<G, HP>((cold && h < 4) ? 0 : d.wait);
    __builtin_amdgcn_s_barrier();
    const int my_issue = (s0 + h < s1) ? d.issue : -1;
    if constexpr (!LATE) {
      if (my_issue >= 0) RH_HALO(my_issue);
      issue_w(h + 4);
    }
    prev_issue = my_issue;
    d = dn;
    dn = fetch(h + 2);
    load_frags((h + 1) & (NSLOT - 1), d, 0, fa[0], fb[0]);
    mma(fa[1], fb[1]);
#ifndef BSMI_NO_SCHED_HINTS
    if constexpr (!LATE) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
      }
    }
#pragma unroll
    for (int k = 0; k < FM + FN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
    }
#endif
  }
#undef RH_HALO
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // nobody still reads LDS when the next tile's prologue stages into it

  if (part) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) *(f32x16_t*)(part + ((size_t)(i * FN + j) * (64 * NW) + tid) * 16) = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // epilogue as in conv_igemm.hip: 16 rows at a time through a per-wave LDS strip, 16-byte streaming stores;
  // the rows of the input raster that are no output voxel (xx >= Wo, yy >= Ho) are dropped here
  constexpr int ESZ = (int)sizeof(T);
  constexpr int PITCH = WTN * ESZ + 16;
  constexpr int CPR = WTN * ESZ / 16;
  constexpr int NCH = 16 * CPR;
  constexpr bool SPLIT = IsSplit<T>::value;  // (hi, lo) output vectors: the lo values go through a second set of strips
  constexpr int LO_STRIPS = NW * 16 * PITCH;
  static_assert((SPLIT ? 2 : 1) * NW * 16 * PITCH <= RING + NSLOT * SLOT, "epilogue strips fit in LDS");
  char* strip = smem + wave * (16 * PITCH);
  T* out = (T*)a.out;
  float bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + lr;
    bv[j] = n < a.Npad ? a.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
          float v = acc[i][j][hf * 8 + rr] + bv[j];
          if (a.relu) v = v > 0.f ? v : 0.f;
          Elem<T>::store((T*)(strip + row * PITCH) + j * 32 + lr, v);
          if constexpr (SPLIT) Elem<T>::store((T*)(strip + LO_STRIPS + row * PITCH) + j * 32 + lr, split_lo(v));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < (NCH + 63) / 64; ++k) {
        const int c = lane + 64 * k;
        if (c >= NCH) break;
        const int row = c / CPR, cc = c - row * CPR;
        const u32x4_t v = *(const u32x4_t*)(strip + row * PITCH + cc * 16);
        const int q = q0 + wm * WTM + i * 32 + hf * 16 + row;
        const int n = n0 + wn * WTN + cc * (16 / ESZ);
        if (q < a.Q && n < a.Co) {
          const int xx = q % a.Win;
          const int zy = q / a.Win;
          const int yy = zy % a.Hin, z = zy / a.Hin;
          if (xx < a.Wo && yy < a.Ho) {
            T* dst = out + act_index<T>(((size_t)(z * a.Ho + yy) * a.Wo + xx) * a.Co, n);
            store_stream16(dst, v);
            if constexpr (SPLIT) store_stream16(dst + kSplitLoElems, *(const u32x4_t*)(strip + LO_STRIPS + row * PITCH + cc * 16));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// wave -> (weight pieces per K-step, early / late) instantiation
template <typename T, int BN, int WM, int WN, int HP>
__device__ __forceinline__ void conv_rh_dispatch(const RhArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  constexpr int NW = kRhNW, NBP = BN / 16;
  constexpr int HI = (NBP + NW - 1) / NW, LO = NBP / NW, R = NBP % NW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (R == 0) {
    if (wave < 4) conv_rh_body<T, BN, WM, WN, HP, HI, false>(a, smem, tile, s0, s1, part);
    else conv_rh_body<T, BN, WM, WN, HP, HI, true>(a, smem, tile, s0, s1, part);
  } else if constexpr (R == 4) {
    if (wave < 4) conv_rh_body<T, BN, WM, WN, HP, HI, false>(a, smem, tile, s0, s1, part);
    else conv_rh_body<T, BN, WM, WN, HP, LO, true>(a, smem, tile, s0, s1, part);
  } else {
    static_assert(R < 4, "weight pieces: the uneven part must fall on the early waves");
    if (wave < R) conv_rh_body<T, BN, WM, WN, HP, HI, false>(a, smem, tile, s0, s1, part);
    else if (wave < 4) conv_rh_body<T, BN, WM, WN, HP, LO, false>(a, smem, tile, s0, s1, part);
    else conv_rh_body<T, BN, WM, WN, HP, LO, true>(a, smem, tile, s0, s1, part);
  }
}

template <typename T, int BN, int WM, int WN, int HP>
__global__ __launch_bounds__(64 * kRhNW, 1) void conv_rh_kernel(const RhArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = ((a.Q + 255) / 256) * (a.Npad / BN);
  const int q = ntiles >> 3, r = ntiles & 7;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  conv_rh_dispatch<T, BN, WM, WN, HP>(a, smem, tile, 0, a.nsteps, nullptr);
}

// persistent form: per-XCD work queues of full tiles followed by split-K tail parts (same scheme
// and geometry as conv_igemm_sk_kernel)
struct RhSkGeom {
  int base, count, per, rounds, rem, P;
};
__device__ __forceinline__ RhSkGeom rh_sk_geom(int ntiles, int xcd, int G) {
  RhSkGeom g;
  const int q = ntiles >> 3, r = ntiles & 7;
  g.base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  g.count = q + (xcd < r ? 1 : 0);
  g.per = G >> 3;
  g.rounds = g.count / g.per;
  g.rem = g.count - g.rounds * g.per;
  g.P = g.rem ? (g.per / g.rem < 16 ? g.per / g.rem : 16) : 1;
  return g;
}

template <typename T, int BN, int WM, int WN, int HP>
__global__ __launch_bounds__(64 * kRhNW, 1) void conv_rh_sk_kernel(const RhArgs a, float* ws, int* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int sh_item;
  const int ntiles = ((a.Q + 255) / 256) * (a.Npad / BN);
  const int S = a.nsteps;
  for (int k = 0; k < 8; ++k) {
    const int xcd = (blockIdx.x + k) & 7;
    const RhSkGeom g = rh_sk_geom(ntiles, xcd, gridDim.x);
    const int nfull = g.rounds * g.per, nitems = nfull + g.rem * g.P;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) sh_item = atomicAdd(&counters[xcd], 1);
      __syncthreads();
      const int it = __builtin_amdgcn_readfirstlane(sh_item);
      if (it >= nitems) break;
      if (it < nfull) {
        conv_rh_dispatch<T, BN, WM, WN, HP>(a, smem, g.base + it, 0, S, nullptr);
      } else {
        const int r = it - nfull;
        const int rt = r / g.P, part = r - rt * g.P;
        const int sa = (int)((long long)S * part / g.P), sb = (int)((long long)S * (part + 1) / g.P);
        float* dst = g.P == 1 ? nullptr : ws + ((size_t)xcd * g.per + r) * (256 * BN);
        if (sb > sa) conv_rh_dispatch<T, BN, WM, WN, HP>(a, smem, g.base + nfull + rt, sa, sb, dst);
      }
    }
  }
}

// finishes the split-K tail tiles: grid (tail tile, xcd, fragment), one 32 x 32 accumulator
// fragment per wave and block
template <typename T, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * kRhNW) void conv_rh_fixup_kernel(const RhArgs a, const float* ws, int G, int* counters) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;
  constexpr int NW = kRhNW, WTM = 256 / WM, WTN = BN / WN, FN = WTN / 32;
  const int xcd = blockIdx.y, rt = blockIdx.x, frag = blockIdx.z;
  const int i = frag / FN, j = frag - i * FN;
  const int ntn = a.Npad / BN;
  const RhSkGeom g = rh_sk_geom(((a.Q + 255) / 256) * ntn, xcd, G);
  if (rt >= g.rem || g.P == 1) return;
  const int S = a.nsteps;
  const int tile = g.base + g.rounds * g.per + rt;
  const int tile_m = tile / ntn, tile_n = tile - tile_m * ntn;
  const int q0 = tile_m * 256, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, lr = lane & 31, lh = lane >> 5;
  const float* p0 = ws + ((size_t)xcd * g.per + rt * g.P) * (256 * BN);
  const size_t o = ((size_t)frag * (64 * NW) + tid) * 16;
  f32x16_t x;
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.f;
  for (int part = 0; part < g.P; ++part) {
    if ((long long)S * (part + 1) / g.P == (long long)S * part / g.P) continue;
    x += *(const f32x16_t*)(p0 + (size_t)part * (256 * BN) + o);
  }
  const int n = n0 + wn * WTN + j * 32 + lr;
  if (n >= a.Co) return;
  const float bv = a.bias[n];
  T* out = (T*)a.out;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q >= a.Q) continue;
    const int xx = q % a.Win;
    const int zy = q / a.Win;
    const int yy = zy % a.Hin, z = zy / a.Hin;
    if (xx >= a.Wo || yy >= a.Ho) continue;
    float v = x[r] + bv;
    if (a.relu) v = v > 0.f ? v : 0.f;
    T* dst = out + act_index<T>(((size_t)(z * a.Ho + yy) * a.Wo + xx) * a.Co, n);
    Elem<T>::store(dst, v);
    if constexpr (IsSplit<T>::value) Elem<T>::store(dst + kSplitLoElems, split_lo(v));
  }
}

// ---- host side ---------------------------------------------------------------------------
int rh_halo_pieces(TileCfg cfg) {
  switch (cfg) {
    case TILE_256x32: return 6;   // 768 rows: Win <= 255 with 3x3 in-plane taps
    case TILE_256x64: return 4;   // 512 rows: Win <= 127
    case TILE_256x256:
    case TILE_256x320: return 3;  // 384 rows: Win <= 63
    default: return 0;
  }
}

bool rh_supported(TileCfg cfg, int Win, int ky, int kx) {
  const int hp = rh_halo_pieces(cfg);
  return hp > 0 && 256 + (ky - 1) * Win + (kx - 1) <= hp * kRhNW * 16;
}

template <typename T, int BN, int WM, int WN, int HP>
static int launch_rh_one(const RhArgs& a, hipStream_t stream, float* sk_ws, int sk_grid) {
  constexpr int smem = 2 * HP * kRhNW * 16 * kStepRowBytes + 4 * BN * kStepRowBytes;
  static_assert(smem <= 160 * 1024 - 64, "LDS budget");
  static DeviceOnce once;
  // the persistent form of the 320-column tile needs more registers than a wave has (it spilled, and was refused below at
  // run time): it is not compiled at all -- no kernel of the library carries a scratch segment (tests/test_kernel_resources.py)
  constexpr bool has_sk = BN >= 256 && BN < 320;
  static bool sk_ok = has_sk;
  auto kern = conv_rh_kernel<T, BN, WM, WN, HP>;
  using SkFn = decltype(&conv_rh_sk_kernel<T, BN, WM, WN, HP>);
  SkFn kern_sk = nullptr;
  if constexpr (has_sk) kern_sk = conv_rh_sk_kernel<T, BN, WM, WN, HP>;
  const int rc_once = once.run([&]() -> int {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipFuncAttributes fa;  // scratch traffic would break the counted vmcnt waits (see conv_igemm.hip)
    BSMI_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    if (fa.localSizeBytes != 0)
      BSMI_FAIL(BSMI_ERR_STATE, "raster-halo conv kernel BN=%d was compiled with %zu bytes of scratch: counted vmcnt waits are unsafe", BN,
                (size_t)fa.localSizeBytes);
    if constexpr (has_sk) {
      BSMI_HIP(hipFuncSetAttribute((const void*)kern_sk, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      BSMI_HIP(hipFuncGetAttributes(&fa, (const void*)kern_sk));
      sk_ok = fa.localSizeBytes == 0;
    }
    return BSMI_OK;
  });
  if (rc_once) return rc_once;
  const int ntiles = ceil_div(a.Q, 256) * (a.Npad / BN);
  const int rounds = ceil_div(ntiles, sk_grid > 0 ? sk_grid : 1);
  bool persistent = false;
  if constexpr (has_sk) {
    if (sk_ws && sk_ok && sk_grid >= 8 && ntiles % sk_grid != 0 && rounds <= 16 && (size_t)256 * BN <= kStreamKTileElems) {
      int* counters = (int*)(sk_ws + (size_t)sk_grid * kStreamKTileElems);
      hipLaunchKernelGGL(kern_sk, dim3(sk_grid), dim3(64 * kRhNW), smem, stream, a, sk_ws, counters);
      hipLaunchKernelGGL((conv_rh_fixup_kernel<T, BN, WM, WN>), dim3(sk_grid / 8, 8, (256 / WM / 32) * (BN / WN / 32)), dim3(64 * kRhNW),
                         0, stream, a, (const float*)sk_ws, sk_grid, counters);
      persistent = true;
    }
  }
  if (!persistent) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(64 * kRhNW), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

template <typename T>
static int launch_rh_cfg(const RhArgs& a, TileCfg cfg, hipStream_t stream, float* sk_ws, int sk_grid) {
  switch (cfg) {
    case TILE_256x32: return launch_rh_one<T, 32, 8, 1, 6>(a, stream, sk_ws, sk_grid);
    case TILE_256x64: return launch_rh_one<T, 64, 8, 1, 4>(a, stream, sk_ws, sk_grid);
    case TILE_256x256: return launch_rh_one<T, 256, 4, 2, 3>(a, stream, sk_ws, sk_grid);
    case TILE_256x320: return launch_rh_one<T, 320, 4, 2, 3>(a, stream, sk_ws, sk_grid);
    default: BSMI_FAIL(BSMI_ERR_INVALID, "raster-halo conv: no kernel for tile config %d", (int)cfg);
  }
}

int launch_conv_rh(const RhArgs& a, int precision, TileCfg cfg, hipStream_t stream, float* sk_ws, int sk_grid) {
  if (a.Q <= 0 || a.nsteps <= 0 || a.nphases <= 0 || a.Npad % tile_bn(cfg) != 0)
    BSMI_FAIL(BSMI_ERR_INVALID, "raster-halo conv launch: bad geometry Q=%d nsteps=%d Npad=%d", a.Q, a.nsteps, a.Npad);
  if (precision == BSMI_PREC_F32) return launch_rh_cfg<float>(a, cfg, stream, sk_ws, sk_grid);
  if (precision == BSMI_PREC_BF16) return launch_rh_cfg<bf16_elem>(a, cfg, stream, sk_ws, sk_grid);
  if (precision == BSMI_PREC_BF16X3) BSMI_FAIL(BSMI_ERR_INVALID, "the raster-halo kernel of the split-bf16 mode is conv_rh_x3_kernel (launch_conv_rh_x3)");
  BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
}

}  // namespace bsmi
