// Implicit-GEMM 3-D "valid" convolution for gfx950 MFMA.
//
// GEMM view of a ConvPass layer (reference models/3d_affs/unet.py:7-76):
//   out[m][n] = act( bias[n] + sum_{step} sum_{k} A_step[m][k] * W_step[n][k] )
// where m runs over output voxels (z,y,x raster order), n over output channels and
// each K-step is two 32-byte units (kernel tap, 16-channel group) of one source tensor.  The cropped
// 1x1x1 residual branch of ConvPass (unet.py:38-41,67-71) and the channel concat of
// Upsample.forward (unet.py:223) are just more K-steps reading other tensors, so a
// whole ConvPass stage is one launch with a fused bias(+ReLU) epilogue.
//
// Data layout: activations channels-last [D][H][W][Cpad]; weights pre-packed on the
// host as [step][Npad][64 B] (k contiguous), zero padded.  Tiles are staged
// HBM/L2 -> LDS by LDS-DMA as [row][64 B] images with a 16-byte-chunk XOR swizzle
// (chunk ^= (row>>2)&3) so that the ds_read_b128 fragment reads of
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 are bank-conflict free.
#include "conv_dev.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace bsmi {

#ifdef BSMI_STAMP  // dev build: time spent per tile in [loop, drain+barrier, epilogue, store drain] (100 MHz ticks)
__device__ unsigned long long g_stamp[8];
extern "C" int bsmi_debug_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// Staging is LDS-DMA (global_load_lds_dwordx4) into a ring of NSLOT = 4 K-step slots of
// [BM + BN rows][64 B].  One wave instruction moves 16 tile rows x 64 B = 1 KiB: lane l lands
// at row (l>>2), 16-byte slot (l&3) of that KiB and FETCHES the source chunk
// (l&3) ^ ((row>>2)&3): the bank swizzle is applied on the per-lane source address, the LDS
// image stays lane-linear, and the ds_read_b128 fragment reads (same XOR) are conflict free.
// The loads of K-steps h+2..h+4 are in flight while K-step h is multiplied (counted vmcnt,
// raw s_barrier); the single barrier of a K-step sits between its two MFMA groups.
//
// B_INSTR: weight pieces (16 rows x 64 B) this wave stages per K-step.  The BN / 16 pieces of a
// K-step are dealt round-robin to the waves, so B_INSTR may differ by one between the low and the
// high waves: the kernel runs one of two instantiations of this body per wave (the counted
// vmcnt immediates depend on it; the barrier count does not).
template <typename T, int BM, int BN, int WM, int WN, int B_INSTR, bool LATE>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  constexpr int NW = WM * WN;
  static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
  constexpr int ROWB = kStepRowBytes;
  constexpr int NSLOT = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int A_INSTR = BM / 16 / NW;  // LDS-DMA instructions per wave per K-step
  constexpr int G = A_INSTR + B_INSTR;
  constexpr int SLOT = (BM + BN) * ROWB;
  static_assert(BM % (16 * NW) == 0, "tile/wave mismatch");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32");
  static_assert(3 * G <= 63, "vmcnt range");

#ifdef BSMI_STAMP
  const unsigned long long st_begin = wall_clock64();
  const unsigned long long cy_begin = clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;  // 4 dwords per K-step
  const int nsteps = a.nsteps;

  // the K-steps [s0, s1) of output tile `tile`; part != nullptr: raw f32 partial sums instead of the epilogue
  const int ntn = a.Npad / BN;
  const int nloc = s1 - s0;
  const int tile_m = tile / ntn, tile_n = tile - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // rows staged by this lane: row(i) = (i*NW + wave)*16 + (lane>>2); the swizzle key
  // (row>>2)&3 = (lane>>4)&3 does not depend on i or on the wave.
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = (lane >> 4) & 3;
  const int g = lchunk ^ skey;             // source chunk that lands in LDS slot lchunk
  const bool unit1 = (g >> 1) != 0;        // which of the K-step's 2 units this lane fetches
  const uint32_t hoff = (uint32_t)((g & 1) << (IsSplit<T>::value ? 5 : 4));  // split tensors: (hi, lo) vectors interleaved
  // byte offset of row(i)'s output voxel inside each source tensor (three named arrays: a
  // runtime-indexed array would live in scratch and its reload would drain vmcnt every K-step)
  static_assert(kMaxConvTensors == 3, "three source slots");
  uint32_t ro0[A_INSTR], ro1[A_INSTR], ro2[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 16 + lrow;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    const int x = m % a.Wo;
    const int zy = m / a.Wo;
    const int y = zy % a.Ho, z = zy / a.Ho;
    ro0[i] = (uint32_t)(z * a.t[0].sz + y * a.t[0].sy + x * a.t[0].sx);
    ro1[i] = (uint32_t)(z * a.t[1].sz + y * a.t[1].sy + x * a.t[1].sx);
    ro2[i] = (uint32_t)(z * a.t[2].sz + y * a.t[2].sy + x * a.t[2].sx);
  }
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)a.Npad * ROWB;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // descriptor of K-step h (clamped), as three scalars
  struct Desc { int t, d0, d1; };
  auto fetch = [&](int h) -> Desc {  // h: K-step relative to s0
    const int ha = s0 + h;
    const cint_ptr_t d = steps + (ha < nsteps ? ha : nsteps - 1) * 4;
    return Desc{d[0], d[1], d[2]};
  };
  // Issue the LDS-DMA loads of K-step h (descriptor ds) into ring slot h & 3.  Branch free, so
  // that the whole K-loop body is one scheduling region; h past the end re-loads the last K-step
  // into a slot nobody reads any more.
  auto issue = [&](int h, const Desc& ds) {
    const bool t1 = ds.t == 1, t2 = ds.t == 2;
    const uint64_t tbase = t1 ? base1 : (t2 ? base2 : base0);
    const uint32_t lofs = (uint32_t)(unit1 ? ds.d1 : ds.d0) + hoff;
    const gptr_t abase = (gptr_t)tbase;
    const gptr_t wbase = (gptr_t)a.w + (size_t)(s0 + h < nsteps ? s0 + h : nsteps - 1) * wstep;
    const lptr_t la = (lptr_t)(smem + (h & (NSLOT - 1)) * SLOT);
    const lptr_t lb = la + BM * ROWB;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const uint32_t ro = t1 ? ro1[i] : (t2 ? ro2[i] : ro0[i]);
      __builtin_amdgcn_global_load_lds(abase + (size_t)(ro + lofs), la + (i * NW + wave) * 1024, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(wbase + (size_t)i * NW * 16 * ROWB + offb, lb + (i * NW + wave) * 1024, 16, 0, 0);
  };

  const int lr = lane & 31, lh = lane >> 5;
  // per-fragment LDS row offsets and swizzle keys are loop invariant
  uint32_t arow[FM], akey[FM], brow[FN], bkey[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm * WTM + i * 32 + lr;
    arow[i] = row * ROWB;
    akey[i] = (row >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * WTN + j * 32 + lr;
    brow[j] = BM * ROWB + row * ROWB;
    bkey[j] = (row >> 2) & 3;
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

  // prologue: K-steps 0..2 in flight, wait for K-step 0
  issue(0, fetch(0));
  issue(1, fetch(1));
  issue(2, fetch(2));
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
  __builtin_amdgcn_s_barrier();
  load_frags(smem, 0, fa[0], fb[0]);
  // A K-step is two MFMA groups with the barrier between them.  An early wave (the only kind
  // in the 4-wave kernels) stages K-step h+4 in the second group of K-step h; a LATE wave
  // (waves 4-7 of an 8-wave kernel) stages K-step h+3 in the first group instead: the two waves
  // of a SIMD then never sit in their LDS-DMA issue stalls at the same time, one of them is
  // always in a pure fragment-read + MFMA group (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  // Either way K-steps h+2 and h+3 are what may still be in flight at the barrier of K-step h.
  if constexpr (!LATE) issue(3, fetch(3));
  Desc dnext = fetch(LATE ? 3 : 4);

  for (int h = 0; h < nloc; ++h) {
    const char* st = smem + (h & (NSLOT - 1)) * SLOT;
    if constexpr (LATE) {
      issue(h + 3, dnext);
      dnext = fetch(h + 4);
    }
    load_frags(st, 1, fa[1], fb[1]);
    mma(fa[0], fb[0]);
#ifndef BSMI_NO_SCHED_HINTS
    if constexpr (LATE) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (LDS-DMA)
      }
    }
#pragma unroll
    for (int k = 0; k < FM + FN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
    }
#endif
    // K-step boundary.  fa[1]/fb[1] must be in registers before anyone may overwrite this
    // slot; K-step h+1 must have landed everywhere before it is read.  The loads of K-steps
    // h+2 and h+3 stay in flight.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * G) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (!LATE) {
      issue(h + 4, dnext);
      dnext = fetch(h + 5);
    }
    load_frags(smem + ((h + 1) & (NSLOT - 1)) * SLOT, 0, fa[0], fb[0]);
    mma(fa[1], fb[1]);
#ifndef BSMI_NO_SCHED_HINTS
    // interleave the LDS-DMA issue and the next fragment reads with this MFMA group instead
    // of letting them form a clump in front of it (one memory instruction per MFMA gap)
    if constexpr (!LATE) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);  // 1 VMEM read (LDS-DMA)
      }
    }
#pragma unroll
    for (int k = 0; k < FM + FN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);  // 1 DS read
    }
#endif
  }
#ifdef BSMI_STAMP
  const unsigned long long st0 = wall_clock64();
  const unsigned long long cy0 = clock64();
#endif
  // drain the run-ahead loads; after the barrier nobody reads or writes the LDS ring any more
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef BSMI_STAMP
  const unsigned long long st1 = wall_clock64();
#endif

  if (part) {
    // split-K: this workgroup multiplied only part of the tile's K range; leave the raw sums
    // (register order, 64 contiguous bytes per lane) for conv_fixup_kernel
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) *(f32x16_t*)(part + ((size_t)(i * FN + j) * (64 * NW) + tid) * 16) = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // Epilogue: bias (+ReLU), convert, store channels-last.  A lane of the 32x32 accumulator holds
  // 16 rows of ONE channel, so storing from registers would write 2-byte pieces (64-byte runs per
  // row: partial cache lines, measured at ~170 GB/s).  Instead every wave transposes 16 rows of
  // its tile at a time through a private LDS strip and writes 16 bytes per lane, whole rows of
  // WTN channels (256-640 contiguous bytes) per few lanes.
  constexpr int ESZ = (int)sizeof(T);
  constexpr int PITCH = WTN * ESZ + 16;       // +16: the two lane halves (rows +4) fall on different banks
  constexpr int CPR = WTN * ESZ / 16;         // 16-byte chunks per row
  constexpr int NCH = 16 * CPR;               // chunks per 16-row strip
  static_assert(NW * 16 * PITCH <= NSLOT * SLOT, "epilogue strips fit in the ring");
  constexpr bool SPLIT = IsSplit<T>::value;  // (hi, lo) output planes: the lo values go through a second set of strips
  constexpr int LO_STRIPS = NW * 16 * PITCH;
  static_assert(!SPLIT || 2 * LO_STRIPS <= NSLOT * SLOT, "both strip sets fit in the ring");
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
#pragma unroll 1  // rolled: unrolled, the f32 320-wide persistent kernel needs 13 registers too many and spills
      for (int k = 0; k < (NCH + 63) / 64; ++k) {
        const int c = lane + 64 * k;
        if (c >= NCH) break;
        const int row = c / CPR, cc = c - row * CPR;
        const u32x4_t v = *(const u32x4_t*)(strip + row * PITCH + cc * 16);
        const int m = m0 + wm * WTM + i * 32 + hf * 16 + row;
        const int n = n0 + wn * WTN + cc * (16 / ESZ);
        if (m < a.M && n < a.Co) {
          T* dst = out + act_index<T>((size_t)m * a.Co, n);
          store_stream16(dst, v);
          if constexpr (SPLIT) store_stream16(dst + kSplitLoElems, *(const u32x4_t*)(strip + LO_STRIPS + row * PITCH + cc * 16));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
#ifdef BSMI_STAMP
  const unsigned long long st2 = wall_clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a persistent workgroup stages its next tile after this
#ifdef BSMI_STAMP
  if (tid == 0) {
    const unsigned long long st3 = wall_clock64();
    atomicAdd(&g_stamp[0], st1 - st0);
    atomicAdd(&g_stamp[1], st2 - st1);
    atomicAdd(&g_stamp[2], st3 - st2);
    atomicAdd(&g_stamp[3], 1ull);
    atomicAdd(&g_stamp[4], st0 - st_begin);
    atomicAdd(&g_stamp[5], cy0 - cy_begin);  // shader cycles of the K loop: with [4] the clock it ran at
  }
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int B_INSTR, bool LATE>
__device__ __forceinline__ void conv_igemm_body16(const ConvArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  constexpr int NW = WM * WN;
  static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
  constexpr int ROWB = kStepRowBytes;
  constexpr int NSLOT = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;  // 16 x 16 accumulator fragments
  constexpr int NH0 = FN / 2, NH1 = FN - NH0;   // the two MFMA groups of a K-step split the B fragments
  constexpr int A_INSTR = BM / 16 / NW;  // LDS-DMA instructions per wave per K-step
  constexpr int G = A_INSTR + B_INSTR;
  constexpr int SLOT = (BM + BN) * ROWB;
  static_assert(BM % (16 * NW) == 0, "tile/wave mismatch");
  static_assert(WTM % 16 == 0 && WTN % 32 == 0, "wave tile shape");
  static_assert(3 * G <= 63, "vmcnt range");

#ifdef BSMI_STAMP
  const unsigned long long st_begin = wall_clock64();
  const unsigned long long cy_begin = clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const cint_ptr_t steps = (cint_ptr_t)a.steps;  // 4 dwords per K-step
  const int nsteps = a.nsteps;

  // the K-steps [s0, s1) of output tile `tile`; part != nullptr: raw f32 partial sums instead of the epilogue
  const int ntn = a.Npad / BN;
  const int nloc = s1 - s0;
  const int tile_m = tile / ntn, tile_n = tile - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // rows staged by this lane: row(i) = (i*NW + wave)*16 + (lane>>2); the swizzle key
  // (row>>2)&3 = (lane>>4)&3 does not depend on i or on the wave.
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = swz16((lane >> 4) & 3);
  const int g = lchunk ^ skey;             // source chunk that lands in LDS slot lchunk
  const bool unit1 = (g >> 1) != 0;        // which of the K-step's 2 units this lane fetches
  const uint32_t hoff = (uint32_t)((g & 1) << (IsSplit<T>::value ? 5 : 4));
  // byte offset of row(i)'s output voxel inside each source tensor (three named arrays: a
  // runtime-indexed array would live in scratch and its reload would drain vmcnt every K-step)
  static_assert(kMaxConvTensors == 3, "three source slots");
  uint32_t ro0[A_INSTR], ro1[A_INSTR], ro2[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 16 + lrow;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    const int x = m % a.Wo;
    const int zy = m / a.Wo;
    const int y = zy % a.Ho, z = zy / a.Ho;
    ro0[i] = (uint32_t)(z * a.t[0].sz + y * a.t[0].sy + x * a.t[0].sx);
    ro1[i] = (uint32_t)(z * a.t[1].sz + y * a.t[1].sy + x * a.t[1].sx);
    ro2[i] = (uint32_t)(z * a.t[2].sz + y * a.t[2].sy + x * a.t[2].sx);
  }
  const uint64_t base0 = a.t[0].base, base1 = a.t[1].base, base2 = a.t[2].base;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)a.Npad * ROWB;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // descriptor of K-step h (clamped), as three scalars
  struct Desc { int t, d0, d1; };
  auto fetch = [&](int h) -> Desc {  // h: K-step relative to s0
    const int ha = s0 + h;
    const cint_ptr_t d = steps + (ha < nsteps ? ha : nsteps - 1) * 4;
    return Desc{d[0], d[1], d[2]};
  };
  // Issue the LDS-DMA loads of K-step h (descriptor ds) into ring slot h & 3.  Branch free, so
  // that the whole K-loop body is one scheduling region; h past the end re-loads the last K-step
  // into a slot nobody reads any more.
  auto issue = [&](int h, const Desc& ds) {
    const bool t1 = ds.t == 1, t2 = ds.t == 2;
    const uint64_t tbase = t1 ? base1 : (t2 ? base2 : base0);
    const uint32_t lofs = (uint32_t)(unit1 ? ds.d1 : ds.d0) + hoff;
    const gptr_t abase = (gptr_t)tbase;
    const gptr_t wbase = (gptr_t)a.w + (size_t)(s0 + h < nsteps ? s0 + h : nsteps - 1) * wstep;
    const lptr_t la = (lptr_t)(smem + (h & (NSLOT - 1)) * SLOT);
    const lptr_t lb = la + BM * ROWB;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const uint32_t ro = t1 ? ro1[i] : (t2 ? ro2[i] : ro0[i]);
      __builtin_amdgcn_global_load_lds(abase + (size_t)(ro + lofs), la + (i * NW + wave) * 1024, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(wbase + (size_t)i * NW * 16 * ROWB + offb, lb + (i * NW + wave) * 1024, 16, 0, 0);
  };

  // v_mfma_f32_16x16x32_bf16: lane l supplies row (l & 15), K bytes 16 * (l >> 4) .. +15 of the 64-byte
  // K-step row for A and for B, and receives rows 4 * (l >> 4) + r, column (l & 15) of the 16 x 16 block
  const int lr = lane & 15, lq = lane >> 4;
  uint32_t aoff[FM], boff[FN];  // loop-invariant LDS offsets of this lane's fragment pieces
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm * WTM + i * 16 + lr;
    aoff[i] = row * ROWB + ((lq ^ swz16((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * WTN + j * 16 + lr;
    boff[j] = BM * ROWB + row * ROWB + ((lq ^ swz16((row >> 2) & 3)) << 4);
  }
  // A is used by both MFMA groups of its K-step, so the next K-step's A goes to the other buffer
  // (the K loop is unrolled by two to keep every register index a constant); the B halves alternate
  u32x4_t fa[2][FM], fb[2][NH1];
#define LOAD_A(st_, buf_) _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[buf_][i] = *(const u32x4_t*)((st_) + aoff[i])
#define LOAD_B0(st_) _Pragma("unroll") for (int j = 0; j < NH0; ++j) fb[0][j] = *(const u32x4_t*)((st_) + boff[j])
#define LOAD_B1(st_) _Pragma("unroll") for (int j = 0; j < NH1; ++j) fb[1][j] = *(const u32x4_t*)((st_) + boff[NH0 + j])
#define MMA0(buf_) _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < NH0; ++j) \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[buf_][i]), __builtin_bit_cast(bf16x8_t, fb[0][j]), acc[i][j], 0, 0, 0)
#define MMA1(buf_) _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < NH1; ++j) \
    acc[i][NH0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[buf_][i]), __builtin_bit_cast(bf16x8_t, fb[1][j]), acc[i][NH0 + j], 0, 0, 0)

  // prologue: K-steps 0..2 in flight, wait for K-step 0
  issue(0, fetch(0));
  issue(1, fetch(1));
  issue(2, fetch(2));
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
  __builtin_amdgcn_s_barrier();
  LOAD_A(smem, 0);
  LOAD_B0(smem);
  // A K-step is two MFMA groups with the barrier between them.  An early wave (the only kind
  // in the 4-wave kernels) stages K-step h+4 in the second group of K-step h; a LATE wave
  // (waves 4-7 of an 8-wave kernel) stages K-step h+3 in the first group instead: the two waves
  // of a SIMD then never sit in their LDS-DMA issue stalls at the same time, one of them is
  // always in a pure fragment-read + MFMA group (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  // Either way K-steps h+2 and h+3 are what may still be in flight at the barrier of K-step h.
  if constexpr (!LATE) issue(3, fetch(3));
  Desc dnext = fetch(LATE ? 3 : 4);

#ifndef BSMI_NO_SCHED_HINTS
#define HINTS(n_mfma_ds_, with_dma_, grp_)                                     \
  if (with_dma_) {                                                              \
    _Pragma("unroll") for (int k = 0; k < G; ++k) {                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, grp_);                     \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, grp_);                     \
    }                                                                           \
  }                                                                             \
  _Pragma("unroll") for (int k = 0; k < (n_mfma_ds_); ++k) {                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, grp_);                       \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, grp_);                       \
  }
#else
#define HINTS(n_mfma_ds_, with_dma_, grp_)
#endif
  // one K-step with A in fa[AC], the next K-step's A into fa[AN]
#define KSTEP(h_, AC, AN)                                                       \
  {                                                                             \
    const char* st = smem + ((h_) & (NSLOT - 1)) * SLOT;                        \
    if constexpr (LATE) {                                                       \
      issue((h_) + 3, dnext);                                                   \
      dnext = fetch((h_) + 4);                                                  \
    }                                                                           \
    LOAD_B1(st);                                                                \
    MMA0(AC);                                                                   \
    HINTS(NH1, LATE, 0)                                                         \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * G) : "memory");    \
    __builtin_amdgcn_s_barrier();                                               \
    if constexpr (!LATE) {                                                      \
      issue((h_) + 4, dnext);                                                   \
      dnext = fetch((h_) + 5);                                                  \
    }                                                                           \
    const char* stn = smem + (((h_) + 1) & (NSLOT - 1)) * SLOT;                 \
    LOAD_A(stn, AN);                                                            \
    LOAD_B0(stn);                                                               \
    MMA1(AC);                                                                   \
    HINTS(FM + NH0, !LATE, 1)                                                   \
  }
  for (int h = 0; h < nloc; h += 2) {  // nloc is even (host: K-step lists and split-K ranges are padded / cut to pairs)
    KSTEP(h, 0, 1)
    KSTEP(h + 1, 1, 0)
  }
#undef KSTEP
#undef HINTS
#undef LOAD_A
#undef LOAD_B0
#undef LOAD_B1
#undef MMA0
#undef MMA1
#ifdef BSMI_STAMP
  const unsigned long long st0 = wall_clock64();
  const unsigned long long cy0 = clock64();
#endif
  // drain the run-ahead loads; after the barrier nobody reads or writes the LDS ring any more
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef BSMI_STAMP
  const unsigned long long st1 = wall_clock64();
#endif

  if (part) {
    // split-K: this workgroup multiplied only part of the tile's K range; leave the raw sums
    // (register order, 64 contiguous bytes per lane) for conv_fixup_kernel
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) *(f32x4_t*)(part + ((size_t)(i * FN + j) * (64 * NW) + tid) * 4) = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // Epilogue: bias (+ReLU), convert, store channels-last.  A lane of the 32x32 accumulator holds
  // 16 rows of ONE channel, so storing from registers would write 2-byte pieces (64-byte runs per
  // row: partial cache lines, measured at ~170 GB/s).  Instead every wave transposes 16 rows of
  // its tile at a time through a private LDS strip and writes 16 bytes per lane, whole rows of
  // WTN channels (256-640 contiguous bytes) per few lanes.
  constexpr int ESZ = (int)sizeof(T);
  constexpr int PITCH = WTN * ESZ + 16;       // +16: the two lane halves (rows +4) fall on different banks
  constexpr int CPR = WTN * ESZ / 16;         // 16-byte chunks per row
  constexpr int NCH = 16 * CPR;               // chunks per 16-row strip
  static_assert(NW * 16 * PITCH <= NSLOT * SLOT, "epilogue strips fit in the ring");
  constexpr bool SPLIT = IsSplit<T>::value;  // (hi, lo) output planes: the lo values go through a second set of strips
  constexpr int LO_STRIPS = NW * 16 * PITCH;
  static_assert(!SPLIT || 2 * LO_STRIPS <= NSLOT * SLOT, "both strip sets fit in the ring");
  char* strip = smem + wave * (16 * PITCH);
  T* out = (T*)a.out;
  float bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * 16 + lr;
    bv[j] = n < a.Npad ? a.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {  // one 16-row strip per fragment row
#pragma unroll
    for (int j = 0; j < FN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[j];
        if (a.relu) v = v > 0.f ? v : 0.f;
        Elem<T>::store((T*)(strip + (4 * lq + r) * PITCH) + j * 16 + lr, v);
        if constexpr (SPLIT) Elem<T>::store((T*)(strip + LO_STRIPS + (4 * lq + r) * PITCH) + j * 16 + lr, split_lo(v));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < (NCH + 63) / 64; ++k) {
      const int c = lane + 64 * k;
      if (c >= NCH) break;
      const int row = c / CPR, cc = c - row * CPR;
      const u32x4_t v = *(const u32x4_t*)(strip + row * PITCH + cc * 16);
      const int m = m0 + wm * WTM + i * 16 + row;
      const int n = n0 + wn * WTN + cc * (16 / ESZ);
      if (m < a.M && n < a.Co) {
        T* dst = out + act_index<T>((size_t)m * a.Co, n);
        store_stream16(dst, v);
        if constexpr (SPLIT) store_stream16(dst + kSplitLoElems, *(const u32x4_t*)(strip + LO_STRIPS + row * PITCH + cc * 16));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#ifdef BSMI_STAMP
  const unsigned long long st2 = wall_clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a persistent workgroup stages its next tile after this
#ifdef BSMI_STAMP
  if (tid == 0) {
    const unsigned long long st3 = wall_clock64();
    atomicAdd(&g_stamp[0], st1 - st0);
    atomicAdd(&g_stamp[1], st2 - st1);
    atomicAdd(&g_stamp[2], st3 - st2);
    atomicAdd(&g_stamp[3], 1ull);
    atomicAdd(&g_stamp[4], st0 - st_begin);
    atomicAdd(&g_stamp[5], cy0 - cy_begin);  // shader cycles of the K loop: with [4] the clock it ran at
  }
#endif
}

// ---- fused split-bf16 body (BSMI_PREC_BF16X3) --------------------------------------------------------------------
// One LOGICAL K-step = the hi and lo planes of the activation rows (A lo, A hi) and of the weight rows (B hi, B lo),
// each staged ONCE by LDS-DMA, and six MFMA groups out of them:
//     g0  A lo x B hi[first half]      g2  A hi x B hi[first half]      g4  A hi x B lo[first half]
//     g1  A lo x B hi[second half]     g3  A hi x B hi[second half]     g5  A hi x B lo[second half]
// i.e. hi*hi + lo*hi + hi*lo with f32 accumulation, as the K-step-list form does with three K-steps, but with two
// thirds of its LDS-DMA bytes, fragment reads and barriers per MFMA (the kernels sit at ~10.5 TB/s of LDS-DMA fill on
// every tile shape: that traffic, not the MFMA pipe, is what bounds them).
// LDS: [A lo][A hi][B hi][B lo], two entries each (K-step parity).  Entries are released as early as they die:
// barrier X (after g1) frees A lo / A hi / B hi of this K-step for the loads of K-step h + 2 ("batch 1"), barrier Y
// (after g4) frees B lo ("batch 2").  One A register set: the A lo fragments are replaced by A hi ones row block by
// row block while g1 runs (and A hi by the next K-step's A lo during g5); the two B register halves alternate.
// Early waves (all waves of a 4-wave kernel, waves 0-3 of an 8-wave one) stage batch 1 in g2 / g3 and batch 2 in g5,
// LATE waves batch 1 in g5 / g0 and batch 2 in g2, so the two waves of a SIMD do not sit in LDS-DMA issue together.
// CUT: column blocks (of MS) at the end of this wave's tile that are padding and not multiplied -- N = 300 in a 320-column
// tile: the waves of the second column half (the late waves: wn = wave / WM in the 8-wave kernels) own columns 160 .. 319 of
// which 304 .. 319 meet nothing but zero weights and masked stores, so they run 9 column blocks instead of 10 and every SIMD,
// which hosts one wave of each half, 19 MFMA blocks per row block instead of 20.
template <int BM, int BN, int WM, int WN, int MS, int B_INSTR, bool LATE, int CUT = 0>
__device__ __forceinline__ void conv_x3_body(const ConvArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  using T = bf16f_elem;
  constexpr int NW = WM * WN;
  static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
  static_assert(!LATE || NW == 8, "late waves exist in 8-wave kernels only");
  static_assert(CUT == 0 || (LATE && MS == 16), "the cut belongs to the late waves (second column half)");
  constexpr int ROWB = kStepRowBytes;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / MS, FN = WTN / MS - CUT;
  constexpr int NH0 = FN / 2, NH1 = FN - NH0;
  static_assert(MS == 16 || MS == 32, "MFMA shape");
  static_assert(WTM % MS == 0 && WTN % MS == 0 && NH0 >= 1, "wave tile: whole fragments, two B halves");
  constexpr int SUB = MS == 16 ? 1 : 2;  // 16-byte reads per fragment: 32x32x16 walks the 64-byte K-step in two halves
  constexpr int A_INSTR = BM / 16 / NW;
  constexpr int EA = BM * ROWB, EB = BN * ROWB;
  constexpr int OFF_AL = 0, OFF_AH = 2 * EA, OFF_BH = 4 * EA, OFF_BL = 4 * EA + 2 * EB;
  constexpr int G1 = 2 * A_INSTR + B_INSTR, G2 = B_INSTR;  // LDS-DMA instructions of batch 1 / batch 2 per wave
  constexpr int P1 = (G1 + 1) / 2;                         // batch 1 goes out in two parts
  constexpr bool PAIRED = NW == 4;
  // outstanding loads allowed at the barriers (see the schedule above): what was issued after the batch that must have landed
  constexpr int NX = LATE ? G1 : G1 + G2;
  constexpr int NY = LATE ? G2 : G1 + G2;
  static_assert(G1 + G2 <= 63, "vmcnt range");
  static_assert(BM % (16 * NW) == 0, "tile/wave mismatch");
  typedef typename std::conditional<MS == 16, f32x4_t, f32x16_t>::type acc_t;

  // every field of the launch arguments this body needs, as locals (the lambdas below capture these, not `a`)
  // The launch arguments are read from the kernarg segment through a pointer the optimiser cannot see through, where
  // they are needed (tile geometry here, output fields in the epilogue): loaded once at kernel entry they would sit in
  // scalar registers across the whole persistent loop, and the spilled scalars cost the vector registers the 320-wide
  // tile does not have.  (`a` is the first kernel argument of both kernels.)
  typedef const __attribute__((address_space(4))) ConvArgs* kargs_t;
  kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  (void)a;
  const cint_ptr_t steps = (cint_ptr_t)ka->steps;
  const int nsteps = ka->nsteps;
  gptr_t w_hi = (gptr_t)ka->w, w_lo = (gptr_t)ka->w_lo;
  const int aM = ka->M, aNpad = ka->Npad, aWo = ka->Wo, aHo = ka->Ho;
  // opaque per call: the lane geometry below is recomputed per tile rather than hoisted out of a persistent
  // kernel's tile loop, where it would stay live across the whole body (the 256x320 form has no registers for that)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 8-wave kernels: waves 0-3 (early) own the first column half, waves 4-7 (late) the second: a SIMD hosts one of each
  const int wm = NW == 8 ? wave % WM : wave / WN, wn = NW == 8 ? wave / WM : wave % WN;
  const int ntn = aNpad / BN;
  const int nloc = s1 - s0;
  // batched launch (ConvArgs::nbatch): the tile rows of batch b follow those of batch b - 1
  const int ntm = (aM + BM - 1) / BM;
  const int tile_mb = tile / ntn, tile_n = tile - tile_mb * ntn;
  const int bat = tile_mb / ntm, tile_m = tile_mb - bat * ntm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
#ifdef BSMI_STAMP  // dev build: where a tile of a BATCHED launch (the Winograd GEMMs) spends its time, 100 MHz ticks summed over tiles:
  // [0] K loop, [1] drain + barrier, [2] epilogue (strips + store issue), [3] tiles, [4] prologue (tile start -> loop), [5] store drain, [6] K-steps
  const bool stamp_on = ka->nbatch > 1;
  const unsigned long long st_begin = wall_clock64();
#endif

  // staging geometry: as conv_igemm_body (lane l lands at row l >> 2, 16-byte slot l & 3 of its KiB and fetches the
  // source chunk (l & 3) ^ key(row)); the key function belongs to the MFMA shape's fragment reads
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int skey = MS == 16 ? swz16((lane >> 4) & 3) : ((lane >> 4) & 3);
  const int g = lchunk ^ skey;
  const bool unit1 = (g >> 1) != 0;
  const uint32_t hoff = (uint32_t)((g & 1) << 5);  // 8-channel vectors are 32 bytes apart: (hi, lo) interleaved
  static_assert(kMaxConvTensors == 3, "three source slots");
  uint32_t ro0[A_INSTR], ro1[A_INSTR], ro2[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 16 + lrow;
    int m = m0 + row;
    m = m < aM ? m : aM - 1;
    const int x = m % aWo;
    const int zy = m / aWo;
    const int y = zy % aHo, z = zy / aHo;
    ro0[i] = (uint32_t)(z * ka->t[0].sz + y * ka->t[0].sy + x * ka->t[0].sx);
    ro1[i] = (uint32_t)(z * ka->t[1].sz + y * ka->t[1].sy + x * ka->t[1].sx);
    ro2[i] = (uint32_t)(z * ka->t[2].sz + y * ka->t[2].sy + x * ka->t[2].sx);
  }
  const uint64_t abat = (uint64_t)bat * (uint64_t)ka->a_batch;
  const uint64_t base0 = ka->t[0].base + abat, base1 = ka->t[1].base + abat, base2 = ka->t[2].base + abat;
  const uint32_t offb = (uint32_t)((n0 + wave * 16 + lrow) * ROWB + ((lchunk ^ skey) << 4));
  const size_t wstep = (size_t)aNpad * ROWB;
  {
    const size_t wbat = (size_t)bat * (size_t)ka->w_batch;
    w_hi += wbat;
    w_lo += wbat;
  }

  acc_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < (MS == 16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

#define X3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  // (without them: the wide tiles spill, the others run 2-5 % slower)
  struct Desc { int t, d0, d1; };
  auto fetch = [&](int h) __attribute__((always_inline)) -> Desc {
    const int ha = s0 + h;
    const cint_ptr_t d = steps + (ha < nsteps ? ha : nsteps - 1) * 4;
    return Desc{d[0], d[1], d[2]};
  };
#ifdef BSMI_DBG_SKIP_A  // dev build, WRONG RESULTS: the activation rows of 8 of 9 K-steps are not staged -- what the loop would cost if a
  // halo form cut the LDS-DMA fill of the A operand ninefold (timing experiment for the narrow layers)
#define X3_DBG_SKIP_A(h_) (BN <= 64 && (s0 + (h_)) % 9 != 0)
#else
#define X3_DBG_SKIP_A(h_) false
#endif
  // Batch 1 of K-step h_, instructions [K0_, K1_): k < 2 A_INSTR: the A rows, else B hi.  The lo and the hi vectors of a
  // row share their 128-byte lines: the 4-wave kernels issue them back to back (PAIRED; in the other order the second
  // access finds the line evicted again: the 360 -> 60 channel layer 3.04 -> 2.48 ms), the 8-wave kernels all lo row blocks
  // first (measured 3-5 % faster there, two row blocks per wave only).  Macros, not lambdas:
  // a closure that selects among the three row-offset arrays keeps them (and everything else it captures) in scratch.
#define ISSUE1(h_, ds_, K0_, K1_)                                                                                         \
  do {                                                                                                                     \
    const bool t1_ = (ds_).t == 1, t2_ = (ds_).t == 2;                                                                     \
    const gptr_t abase_ = (gptr_t)(t1_ ? base1 : (t2_ ? base2 : base0));                                                   \
    const uint32_t lofs_ = (uint32_t)(unit1 ? (ds_).d1 : (ds_).d0) + hoff;                                                 \
    const int p_ = (h_) & 1;                                                                                               \
    const size_t wi_ = (size_t)(s0 + (h_) < nsteps ? s0 + (h_) : nsteps - 1) * wstep;                                      \
    _Pragma("unroll") for (int k_ = (K0_); k_ < (K1_); ++k_) {                                                             \
      if (k_ < 2 * A_INSTR) {                                                                                              \
        if (X3_DBG_SKIP_A(h_)) continue;                                                                                   \
        const int i_ = PAIRED ? k_ >> 1 : (k_ < A_INSTR ? k_ : k_ - A_INSTR);                                              \
        const bool lo_ = PAIRED ? (k_ & 1) == 0 : k_ < A_INSTR;                                                            \
        const uint32_t r0_ = ro0[i_], r1_ = ro1[i_], r2_ = ro2[i_];                                                        \
        const uint32_t src_ = (t1_ ? r1_ : (t2_ ? r2_ : r0_)) + lofs_ + (lo_ ? 16u : 0u);                                  \
        const lptr_t dst_ = (lptr_t)(smem + (lo_ ? OFF_AL : OFF_AH) + p_ * EA) + (i_ * NW + wave) * 1024;                  \
        __builtin_amdgcn_global_load_lds(abase_ + (size_t)src_, dst_, 16, 0, 0);                                           \
      } else {                                                                                                             \
        const int i_ = k_ - 2 * A_INSTR;                                                                                   \
        __builtin_amdgcn_global_load_lds(w_hi + wi_ + (size_t)i_ * NW * 16 * ROWB + offb,                                  \
                                         (lptr_t)(smem + OFF_BH + p_ * EB) + (i_ * NW + wave) * 1024, 16, 0, 0);           \
      }                                                                                                                    \
    }                                                                                                                      \
  } while (0)
  // batch 2: B lo of K-step h_
#define ISSUE2(h_)                                                                                                         \
  do {                                                                                                                     \
    const int p_ = (h_) & 1;                                                                                               \
    const size_t wi_ = (size_t)(s0 + (h_) < nsteps ? s0 + (h_) : nsteps - 1) * wstep;                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < B_INSTR; ++i_)                                                                 \
      __builtin_amdgcn_global_load_lds(w_lo + wi_ + (size_t)i_ * NW * 16 * ROWB + offb,                                    \
                                       (lptr_t)(smem + OFF_BL + p_ * EB) + (i_ * NW + wave) * 1024, 16, 0, 0);             \
  } while (0)

  // fragment addresses of this lane (loop invariant).  The swizzle key of a row is (row >> 2) & 3 and every fragment row
  // block starts at a multiple of 16 rows, so the key -- and with it the chunk -- is the same for all blocks: one base
  // offset per operand, the block's i * MS rows go into the instruction's immediate offset.
  const int lr = MS == 16 ? (lane & 15) : (lane & 31);
  const int lq = MS == 16 ? (lane >> 4) : (lane >> 5);
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && MS % 16 == 0, "row blocks keep the swizzle key");
  uint32_t aoff0[SUB], boff0[SUB];
#pragma unroll
  for (int sb = 0; sb < SUB; ++sb) {
    const int key = MS == 16 ? swz16((lr >> 2) & 3) : ((lr >> 2) & 3);
    const int chunk = MS == 16 ? (lq ^ key) : ((2 * sb + lq) ^ key);
    aoff0[sb] = (uint32_t)((wm * WTM + lr) * ROWB + (chunk << 4));
    boff0[sb] = (uint32_t)((wn * WTN + lr) * ROWB + (chunk << 4));
  }
  u32x4_t RA[FM][SUB], RB0[NH0][SUB], RB1[NH1][SUB];  // indexed by compile-time constants only
  // tiles with registers to spare: the A hi fragments get their own set and are read a group early (the 1500-channel layers
  // 12.7 -> 12.3 ms); the 320-wide tile replaces its A lo fragments row block by row block while group 1 runs
  constexpr bool TWO_A = BN <= 256;
  u32x4_t RAH[TWO_A ? FM : 1][SUB];
#define AH(i_) (TWO_A ? RAH[(i_)] : RA[(i_)])
  auto rdA = [&](u32x4_t* dst, const char* entry, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) dst[sb] = *(const u32x4_t*)(entry + aoff0[sb] + i * (MS * ROWB));
  };
  auto rdB = [&](u32x4_t* dst, const char* entry, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) dst[sb] = *(const u32x4_t*)(entry + boff0[sb] + j * (MS * ROWB));
  };
  auto fma = [&](acc_t& c, const u32x4_t* x, const u32x4_t* y) __attribute__((always_inline)) {
    if constexpr (MS == 16) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, x[0]), __builtin_bit_cast(bf16x8_t, y[0]), c, 0, 0, 0);
    } else {
#pragma unroll
      for (int sb = 0; sb < SUB; ++sb)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, x[sb]), __builtin_bit_cast(bf16x8_t, y[sb]), c, 0, 0, 0);
    }
  };

  // prologue: K-steps 0 and 1 (late waves: K-step 1 up to the first part of batch 1, the rest goes out in the loop)
  {
    const Desc d0 = fetch(0), d1 = fetch(1);
    ISSUE1(0, d0, 0, G1);
    ISSUE2(0);
    ISSUE1(1, d1, 0, P1);
    if constexpr (!LATE) {
      ISSUE1(1, d1, P1, G1);
      ISSUE2(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G1 + G2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P1) : "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < FM; ++i) rdA(RA[i], smem + OFF_AL, i);
#pragma unroll
  for (int j = 0; j < NH0; ++j) rdB(RB0[j], smem + OFF_BH, j);
  Desc dl = fetch(1);  // late waves: descriptor of the batch 1 whose second part is still to go out
#ifdef BSMI_STAMP
  const unsigned long long st_loop = wall_clock64();
#endif

  for (int h = 0; h < nloc; ++h) {
    const int p = h & 1, q = p ^ 1;
    const char* eAH = smem + OFF_AH + p * EA;
    const char* eBH = smem + OFF_BH + p * EB;
    const char* eBL = smem + OFF_BL + p * EB;
    // g0: A lo x B hi, first half; the second half of B hi arrives in RB1
    if constexpr (LATE) { ISSUE1(h + 1, dl, P1, G1); }
#pragma unroll
    for (int j = 0; j < NH1; ++j) rdB(RB1[j], eBH, NH0 + j);
    if constexpr (TWO_A) {
#pragma unroll
      for (int i = 0; i < FM; ++i) rdA(RAH[i], eAH, i);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < NH0; ++j) fma(acc[i][j], RA[i], RB0[j]);
    X3_SCHED_FENCE();  // keep the groups apart: reads hoisted across them cost registers the 8-wave kernels do not have
    // g1: A lo x B hi, second half, row block by row block; each A lo block gives way to its A hi block
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < NH1; ++j) fma(acc[i][NH0 + j], RA[i], RB1[j]);
      if constexpr (!TWO_A) rdA(RA[i], eAH, i);
    }
    X3_SCHED_FENCE();
    // X: A lo / A hi / B hi of this K-step are read; batch 2 of this K-step (B lo) has landed everywhere
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NX) : "memory");
    __builtin_amdgcn_s_barrier();
    const Desc dn = fetch(h + 2);
    X3_SCHED_FENCE();
    // g2: A hi x B hi, first half
    if constexpr (!LATE) { ISSUE1(h + 2, dn, 0, P1); }
    else ISSUE2(h + 1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < NH0; ++j) fma(acc[i][j], AH(i), RB0[j]);
    X3_SCHED_FENCE();
    // g3: A hi x B hi, second half; the first half of B lo arrives in RB0
    if constexpr (!LATE) { ISSUE1(h + 2, dn, P1, G1); }
#pragma unroll
    for (int j = 0; j < NH0; ++j) rdB(RB0[j], eBL, j);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < NH1; ++j) fma(acc[i][NH0 + j], AH(i), RB1[j]);
    X3_SCHED_FENCE();
    // g4: A hi x B lo, first half; the second half of B lo arrives in RB1
#pragma unroll
    for (int j = 0; j < NH1; ++j) rdB(RB1[j], eBL, NH0 + j);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < NH0; ++j) fma(acc[i][j], AH(i), RB0[j]);
    X3_SCHED_FENCE();
    // Y: B lo of this K-step is read; batch 1 of K-step h + 1 has landed everywhere
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NY) : "memory");
    __builtin_amdgcn_s_barrier();
    X3_SCHED_FENCE();
    // g5: A hi x B lo, second half; A lo and the first half of B hi of the next K-step arrive
    if constexpr (!LATE) ISSUE2(h + 2);
    else { dl = dn; ISSUE1(h + 2, dl, 0, P1); }
#pragma unroll
    for (int j = 0; j < NH0; ++j) rdB(RB0[j], smem + OFF_BH + q * EB, j);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < NH1; ++j) fma(acc[i][NH0 + j], AH(i), RB1[j]);
      rdA(RA[i], smem + OFF_AL + q * EA, i);
    }
  }
#ifdef BSMI_STAMP
  const unsigned long long st0 = wall_clock64();
#endif
  // drain the run-ahead loads; after the barrier nobody reads or writes the staging area any more
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef BSMI_STAMP
  const unsigned long long st1 = wall_clock64();
#endif

  if (part) {
    constexpr int NR = MS == 16 ? 4 : 16;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) *(acc_t*)(part + ((size_t)(i * (WTN / MS) + j) * (64 * NW) + tid) * NR) = acc[i][j];  // conv_fixup_kernel's fragment order (a cut wave leaves its last fragments unwritten: padding columns, never stored)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // epilogue: bias (+ReLU), (hi, lo) planes through per-wave LDS strips, 16-byte streaming stores (as the other bodies)
  constexpr int ESZ = 2;
  constexpr int PITCH = WTN * ESZ + 16;
  constexpr int CPR = WTN * ESZ / 16;
  constexpr int NCH = 16 * CPR;
  constexpr int LO_STRIPS = NW * 16 * PITCH;
  static_assert(2 * LO_STRIPS <= 4 * (EA + EB), "both strip sets fit in the staging area");
  char* strip = smem + wave * (16 * PITCH);
  kargs_t kb = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kb));
  const float* const bias = kb->bias;
  const int aCo = kb->Co, relu = kb->relu;
  T* out = (T*)kb->out;
  constexpr int NSTRIP = MS / 16;  // 16-row strips per fragment row block
  const size_t orow0 = (size_t)bat * (size_t)aM;  // first output row of this batch
  if (kb->raw) {
    // transform-domain sums of a Winograd layer (wino.hip): f32 as they are, one strip of 16 rows x WTN floats per wave
    constexpr int PITCHF = WTN * 4 + 16;
    constexpr int CPRF = WTN * 4 / 16;
    constexpr int NCHF = 16 * CPRF;
    static_assert(NW * 16 * PITCHF <= 2 * LO_STRIPS, "f32 strips fit where the (hi, lo) strips do");
    char* stripf = smem + wave * (16 * PITCHF);
    float* outf = (float*)kb->out;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int hf = 0; hf < NSTRIP; ++hf) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
          for (int rr = 0; rr < (MS == 16 ? 4 : 8); ++rr) {
            const int row = MS == 16 ? 4 * lq + rr : (rr & 3) + 8 * (rr >> 2) + 4 * lq;
            *((float*)(stripf + row * PITCHF) + j * MS + lr) = acc[i][j][MS == 16 ? rr : hf * 8 + rr];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < (NCHF + 63) / 64; ++k) {
          const int c = lane + 64 * k;
          if (c >= NCHF) break;
          const int row = c / CPRF, cc = c - row * CPRF;
          const u32x4_t v = *(const u32x4_t*)(stripf + row * PITCHF + cc * 16);
          const int m = m0 + wm * WTM + i * MS + hf * 16 + row;
          const int n = n0 + wn * WTN + cc * 4;
          if (m < aM && n < aCo) store_stream16(outf + (orow0 + (size_t)m) * aCo + n, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
#ifdef BSMI_STAMP
    const unsigned long long st2 = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef BSMI_STAMP
    if (tid == 0 && stamp_on) {
      const unsigned long long st3 = wall_clock64();
      atomicAdd(&g_stamp[0], st0 - st_loop);
      atomicAdd(&g_stamp[1], st1 - st0);
      atomicAdd(&g_stamp[2], st2 - st1);
      atomicAdd(&g_stamp[3], 1ull);
      atomicAdd(&g_stamp[4], st_loop - st_begin);
      atomicAdd(&g_stamp[5], st3 - st2);
      atomicAdd(&g_stamp[6], (unsigned long long)nloc);
    }
#endif
    return;
  }
  float bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * MS + lr;
    bv[j] = n < aNpad ? bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int hf = 0; hf < NSTRIP; ++hf) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int rr = 0; rr < (MS == 16 ? 4 : 8); ++rr) {
          // 16x16: register r holds row 4 (lane >> 4) + r; 32x32: register hf * 8 + rr holds row (rr & 3) + 8 (rr >> 2) + 4 (lane >> 5) of half hf
          const int row = MS == 16 ? 4 * lq + rr : (rr & 3) + 8 * (rr >> 2) + 4 * lq;
          float v = acc[i][j][MS == 16 ? rr : hf * 8 + rr] + bv[j];
          if (relu) v = v > 0.f ? v : 0.f;
          Elem<T>::store((T*)(strip + row * PITCH) + j * MS + lr, v);
          Elem<T>::store((T*)(strip + LO_STRIPS + row * PITCH) + j * MS + lr, split_lo(v));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < (NCH + 63) / 64; ++k) {
        const int c = lane + 64 * k;
        if (c >= NCH) break;
        const int row = c / CPR, cc = c - row * CPR;
        const u32x4_t v = *(const u32x4_t*)(strip + row * PITCH + cc * 16);
        const int m = m0 + wm * WTM + i * MS + hf * 16 + row;
        const int n = n0 + wn * WTN + cc * (16 / ESZ);
        if (m < aM && n < aCo) {
          T* dst = out + act_index<T>((orow0 + (size_t)m) * aCo, n);
          store_stream16(dst, v);
          store_stream16(dst + kSplitLoElems, *(const u32x4_t*)(strip + LO_STRIPS + row * PITCH + cc * 16));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a persistent workgroup stages its next tile after this
}
#undef ISSUE1
#undef ISSUE2
#undef X3_SCHED_FENCE
#undef AH

// wave -> (weight-piece count, early/late) instantiation of the body; MS = MFMA shape (32: 32x32x16 /
// 32x32x2, 16: v_mfma_f32_16x16x32_bf16, which holds a ~12 % higher clock on real data: MI355X_MICROARCH.md,
// DVFS give-back item 7)
template <typename T, int BM, int BN, int WM, int WN, int MS, int BI, bool LATE, int CUT = 0>
__device__ __forceinline__ void conv_igemm_body_ms(const ConvArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  if constexpr (IsFused<T>::value) conv_x3_body<BM, BN, WM, WN, MS, BI, LATE, CUT>(a, smem, tile, s0, s1, part);
  else if constexpr (MS == 16) conv_igemm_body16<T, BM, BN, WM, WN, BI, LATE>(a, smem, tile, s0, s1, part);
  else conv_igemm_body<T, BM, BN, WM, WN, BI, LATE>(a, smem, tile, s0, s1, part);
}

template <typename T, int BM, int BN, int WM, int WN, int MS, int CUT = 0>
__device__ __forceinline__ void conv_igemm_dispatch(const ConvArgs& a, char* smem, int tile, int s0, int s1, float* part) {
  constexpr int NW = WM * WN, NBP = BN / 16;
  constexpr int HI = (NBP + NW - 1) / NW, LO = NBP / NW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (NW == 8) {
    static_assert(HI == LO || NBP % NW == 4, "the uneven split must coincide with the early/late split");
    if (wave < 4) conv_igemm_body_ms<T, BM, BN, WM, WN, MS, HI, false>(a, smem, tile, s0, s1, part);
    else conv_igemm_body_ms<T, BM, BN, WM, WN, MS, LO, true, CUT>(a, smem, tile, s0, s1, part);
  } else if constexpr (HI == LO) {
    conv_igemm_body_ms<T, BM, BN, WM, WN, MS, HI, false>(a, smem, tile, s0, s1, part);
  } else {
    if (wave < NBP % NW) conv_igemm_body_ms<T, BM, BN, WM, WN, MS, HI, false>(a, smem, tile, s0, s1, part);
    else conv_igemm_body_ms<T, BM, BN, WM, WN, MS, LO, false>(a, smem, tile, s0, s1, part);
  }
}

// T: element type; BM x BN block tile; WM x WN waves: 4 (one per SIMD, up to 512 registers
// each: 128x128 register tiles) or 8 (two per SIMD, 256 registers each: while one wave of a
// SIMD sits in the 60-185 cycles an LDS-DMA instruction costs its issuer, the other multiplies).
template <typename T, int BM, int BN, int WM, int WN, int MS, int CUT = 0>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_igemm_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NSLOT][A: BM rows | B: BN rows][64 B]
  // XCD-aware tile map: consecutive block ids are dealt round-robin to the 8 XCDs, so give
  // each XCD a contiguous run of tiles; n fastest, so the blocks resident on one XCD cover few
  // row panels x all weight panels and the tap re-reads of an activation line hit its L2.
  const int ntiles = ((a.M + BM - 1) / BM) * (a.Npad / BN) * (a.nbatch > 1 ? a.nbatch : 1);
  const int q = ntiles >> 3, r = ntiles & 7;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  conv_igemm_dispatch<T, BM, BN, WM, WN, MS, CUT>(a, smem, tile, 0, a.nsteps, nullptr);
}

// Persistent form with a split-K tail: one workgroup per CU.  A plain launch of 779 tiles on 256
// CUs runs 4 rounds for 3.04 rounds of work.  Here the tiles keep the XCD-contiguous order of
// conv_igemm_kernel (the 32 workgroups of an XCD multiply 32 neighbouring tiles at any time, so
// an activation panel and the weight panels are shared through that XCD's L2); every workgroup
// multiplies its whole rounds of full tiles, and the `rem` (< 32) tiles an XCD has left over are
// each cut along K into P = min(32 / rem, 16) parts.  The raw partial sums go
// to `ws` ([xcd][32 slots][BM x BN] f32) and conv_fixup_kernel finishes those tiles.
struct SkGeom {
  int base, count, per, rounds, rem, P;  // this XCD's first tile, tile count, workgroups, full rounds, tail tiles, parts per tail tile
};
__device__ __forceinline__ SkGeom sk_geom(int ntiles, int xcd, int G) {
  SkGeom g;
  const int q = ntiles >> 3, r = ntiles & 7;
  g.base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  g.count = q + (xcd < r ? 1 : 0);
  g.per = G >> 3;
  g.rounds = g.count / g.per;
  g.rem = g.count - g.rounds * g.per;
  g.P = g.rem ? (g.per / g.rem < 16 ? g.per / g.rem : 16) : 1;
  return g;
}

// Work is handed out dynamically: per XCD a queue (an atomic counter in `counters`) of that XCD's
// full tiles followed by its tail parts.  A workgroup drains the queue of the XCD it runs on, then
// helps the others, so a CU that other streams keep busy (the segmentation lanes of the block
// pipeline) only delays its own share by one tile; a workgroup that becomes resident late finds
// the queues empty and leaves.  conv_fixup_kernel re-zeroes the counters for the next launch.
template <typename T, int BM, int BN, int WM, int WN, int MS, int CUT = 0>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_igemm_sk_kernel(const ConvArgs a, float* ws, int* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the queue item travels through the first word of the staging area (free between two tiles): a static variable would push
  // the 64-column tiles (80 KB of staging, two workgroups per CU) over half a CU's LDS
  volatile int* sh_item = (volatile int*)smem;
  const int ntiles = ((a.M + BM - 1) / BM) * (a.Npad / BN) * (a.nbatch > 1 ? a.nbatch : 1);
  const int S = a.nsteps;
  // the XCD this workgroup really runs on (blockIdx % 8 is only the usual placement: a CU-masked queue deals differently)
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  for (int k = 0; k < 8; ++k) {
    const int xcd = ((int)(xcc & 7) + k) & 7;
    const SkGeom g = sk_geom(ntiles, xcd, gridDim.x);
    const int nfull = g.rounds * g.per, nitems = nfull + g.rem * g.P;
    for (;;) {
      __syncthreads();  // the previous tile's epilogue strips are read
      if (threadIdx.x == 0) *sh_item = atomicAdd(&counters[xcd], 1);
      __syncthreads();
      const int it = __builtin_amdgcn_readfirstlane(*sh_item);
      __syncthreads();  // everybody has the item before the next prologue stages over it
      if (it >= nitems) break;
      // one call site for full tiles and tail parts (the body is large: two inlined copies cost the fused kernels their registers)
      int tile_i = g.base + it, sa = 0, sb = S;
      float* dst = nullptr;
      if (it >= nfull) {
        const int r = it - nfull;
        const int rt = r / g.P, part = r - rt * g.P;
        sa = (int)((long long)(S / 2) * part / g.P) * 2;  // K ranges in whole pairs of K-steps (16x16x32 body)
        sb = part + 1 == g.P ? S : (int)((long long)(S / 2) * (part + 1) / g.P) * 2;
        dst = g.P == 1 ? nullptr : ws + ((size_t)xcd * g.per + r) * (BM * BN);
        tile_i = g.base + nfull + rt;
      }
      if (sb > sa) conv_igemm_dispatch<T, BM, BN, WM, WN, MS, CUT>(a, smem, tile_i, sa, sb, dst);
    }
  }
}

// Finishes the tail tiles of conv_igemm_sk_kernel: sum of the P partial tiles in part order
// (deterministic), bias, ReLU, store.  Grid (32, 8): (tail tile, xcd); same thread -> element map
// as the epilogue.
template <typename T, int BM, int BN, int WM, int WN, int MS>
__global__ __launch_bounds__(64 * WM * WN) void conv_fixup_kernel(const ConvArgs a, const float* ws, int G, int* counters) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;
  constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN, FN = WTN / MS;
  constexpr int NR = MS == 16 ? 4 : 16;  // accumulator registers of one fragment
  const int xcd = blockIdx.y, rt = blockIdx.x, frag = blockIdx.z;  // grid: (tail tile, xcd, fragment)
  const int i = frag / FN, j = frag - i * FN;
  const int ntn = a.Npad / BN, ntm = (a.M + BM - 1) / BM;
  const SkGeom g = sk_geom(ntm * ntn * (a.nbatch > 1 ? a.nbatch : 1), xcd, G);
  if (rt >= g.rem || g.P == 1) return;
  const int S = a.nsteps;
  const int tile = g.base + g.rounds * g.per + rt;
  const int tile_mb = tile / ntn, tile_n = tile - tile_mb * ntn;
  const int bat = tile_mb / ntm, tile_m = tile_mb - bat * ntm;  // batched launches: as conv_x3_body
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const size_t orow0 = (size_t)bat * (size_t)a.M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr bool HALVES = IsFused<T>::value && WM * WN == 8;  // conv_x3_body: waves 0-3 own the first column half
  const int wm = HALVES ? wave % WM : wave / WN, wn = HALVES ? wave / WM : wave % WN;
  const float* p0 = ws + ((size_t)xcd * g.per + rt * g.P) * (BM * BN);
  const size_t o = ((size_t)frag * (64 * NW) + tid) * NR;
  float x[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) x[r] = 0.f;
  for (int part = 0; part < g.P; ++part) {
    const int sa = (int)((long long)(S / 2) * part / g.P) * 2;
    const int sb = part + 1 == g.P ? S : (int)((long long)(S / 2) * (part + 1) / g.P) * 2;
    if (sb <= sa) continue;  // empty K range: nothing was written
#pragma unroll
    for (int r = 0; r < NR; ++r) x[r] += p0[(size_t)part * (BM * BN) + o + r];
  }
  const int n = n0 + wn * WTN + j * MS + (MS == 16 ? (lane & 15) : (lane & 31));
  if (n >= a.Co) return;
  if (a.raw) {  // transform-domain sums of a Winograd layer: f32 as they are
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = MS == 16 ? 4 * (lane >> 4) + r : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + wm * WTM + i * MS + row;
      if (m < a.M) ((float*)a.out)[(orow0 + (size_t)m) * a.Co + n] = x[r];
    }
    return;
  }
  const float bv = a.bias[n];
  T* out = (T*)a.out;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = MS == 16 ? 4 * (lane >> 4) + r : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int m = m0 + wm * WTM + i * MS + row;
    if (m < a.M) {
      float v = x[r] + bv;
      if (a.relu) v = v > 0.f ? v : 0.f;
      T* dst = out + act_index<T>((orow0 + (size_t)m) * a.Co, n);
      Elem<T>::store(dst, v);
      if constexpr (IsSplit<T>::value) Elem<T>::store(dst + kSplitLoElems, split_lo(v));
    }
  }
}

bool two_waves_per_simd();
bool mfma_16x16();
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
  // TILE_256x320 exists only as the 8-wave kernel (4 x 2 waves of 64 x 160 register tiles)
  if (!two_waves_per_simd()) eff[4] = 0.01;
  if (const char* e = getenv("BSMI_TILE_EFF")) sscanf(e, "%lf,%lf,%lf,%lf,%lf", &eff[0], &eff[1], &eff[2], &eff[3], &eff[4]);  // experiments
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

template <typename T, int BM, int BN, int WM, int WN, int MS = 32, int CUT = 0>
static int launch_one(const ConvArgs& a, hipStream_t stream, float* sk_ws, int sk_grid) {
  constexpr int smem_need = 4 * (BM + BN) * kStepRowBytes;
  // BSMI_LDS_PAD_KB (dev): every launch asks for at least this much LDS, e.g. 84: no two of these workgroups share a CU
  static const int smem = [] { const char* e = getenv("BSMI_LDS_PAD_KB"); const int pad = e ? atoi(e) * 1024 : 0; return pad > smem_need ? std::min(pad, 160 * 1024) : smem_need; }();
  static DeviceOnce once;
  // the persistent form of the four-wave 256 x 256 tile (BSMI_WAVES8=0, tests) spilled and was refused at run time: it is not
  // compiled at all -- no kernel of the library carries a scratch segment (tests/test_kernel_resources.py)
  constexpr bool has_sk = !(BM == 256 && BN == 256 && WM * WN == 4);
  static bool sk_ok = has_sk;
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, MS, CUT>;
  using SkFn = decltype(&conv_igemm_sk_kernel<T, BM, BN, WM, WN, MS, CUT>);
  SkFn kern_sk = nullptr;
  if constexpr (has_sk) kern_sk = conv_igemm_sk_kernel<T, BM, BN, WM, WN, MS, CUT>;
  const int rc_once = once.run([&]() -> int {
    BSMI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if constexpr (has_sk) BSMI_HIP(hipFuncSetAttribute((const void*)kern_sk, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    // The K loops count their outstanding LDS-DMA groups with s_waitcnt vmcnt(N).  Scratch (spill)
    // traffic is counted by the same counter, so a kernel body that spills would wait for the wrong
    // loads: refuse it (the plain kernel) or do not use it (the persistent form).
    hipFuncAttributes fa;
    BSMI_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    if (fa.localSizeBytes != 0)
      BSMI_FAIL(BSMI_ERR_STATE, "conv kernel %dx%d (%d waves) was compiled with %zu bytes of scratch: counted vmcnt waits are unsafe",
                BM, BN, WM * WN, (size_t)fa.localSizeBytes);
    if constexpr (has_sk) {
      BSMI_HIP(hipFuncGetAttributes(&fa, (const void*)kern_sk));
      sk_ok = fa.localSizeBytes == 0;
    }
    return BSMI_OK;
  });
  if (rc_once) return rc_once;
  const int nbatch = a.nbatch > 1 ? a.nbatch : 1;
  if ((nbatch > 1 || a.raw) && !IsFused<T>::value) BSMI_FAIL(BSMI_ERR_INVALID, "batched / raw-sum conv launches exist in the fused split-bf16 kernels only");
  const int ntiles = ceil_div(a.M, BM) * (a.Npad / BN) * nbatch;
  // persistent + split-K tail when whole rounds would leave a large share of the last one idle (a batched launch is long: any round count)
  const bool big_tile = BN >= 256;
  // the fused 64-column tile: two workgroups per CU (80 KB of staging each).  The 2 565 tiles of the 360 -> 60 channel layer are
  // 5.01 rounds of 512, and the five tiles of the sixth had a tenth of the launch to themselves: 2.62 -> 2.51 ms in the
  // persistent form; with a last round that is more than half full the plain launch is as fast or faster (60 -> 60: 0.54 / 0.57)
  static const bool sk64 = [] { const char* e = getenv("BSMI_X3_SK64"); return !(e && e[0] == '0'); }();
  const bool two_per_cu = IsFused<T>::value && BN == 64 && 2 * smem <= 160 * 1024 && sk64;  // (smem: with a dev pad, one per CU)
  const int grid_sk = two_per_cu ? 2 * sk_grid : sk_grid;
  const int rounds = ceil_div(ntiles, grid_sk > 0 ? grid_sk : 1);
  const bool thin_tail = grid_sk > 0 && 2 * (ntiles % grid_sk) < grid_sk;
  bool persistent = false;
  if constexpr (has_sk)
  if (sk_ws && sk_ok && sk_grid >= 8 && (big_tile || (two_per_cu && rounds >= 2 && rounds <= 8 && thin_tail)) && ntiles % grid_sk != 0 && (rounds <= 16 || nbatch > 1) &&
      (size_t)BM * BN * (two_per_cu ? 2 : 1) <= kStreamKTileElems) {
    int* counters = (int*)(sk_ws + (size_t)sk_grid * kStreamKTileElems);
    const int sk_grid_arg = grid_sk;
    hipLaunchKernelGGL(kern_sk, dim3(sk_grid_arg), dim3(64 * WM * WN), smem, stream, a, sk_ws, counters);
    // fix-up grid: only as many tail tiles per XCD as the fullest XCD has (sk_geom's arithmetic); a (32, 8, fragments) grid
    // dispatched thousands of workgroups that left at once, 75-99 us per launch
    int max_rem = 0;
    {
      const int q = ntiles >> 3, r = ntiles & 7, per = grid_sk >> 3;
      for (int x = 0; x < 8; ++x) {
        const int count = q + (x < r ? 1 : 0);
        max_rem = std::max(max_rem, count - count / per * per);
      }
    }
    hipLaunchKernelGGL((conv_fixup_kernel<T, BM, BN, WM, WN, MS>), dim3(std::max(max_rem, 1), 8, (BM / WM / MS) * (BN / WN / MS)),
                       dim3(64 * WM * WN), 0, stream, a, (const float*)sk_ws, grid_sk, counters);
    persistent = true;
  }
  if (!persistent) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(64 * WM * WN), smem, stream, a);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

bool mfma_16x16() {
  static const bool on = [] { const char* e = getenv("BSMI_MFMA16"); return !e || e[0] != '0'; }();
  return on;
}

bool two_waves_per_simd() {
  static const bool on = [] { const char* e = getenv("BSMI_WAVES8"); return !e || e[0] != '0'; }();
  return on;
}

#ifndef BSMI_X3_64_MS
#define BSMI_X3_64_MS 16  // MFMA shape of the fused 256 x 64 / 256 x 160 tiles (measured 1-4 % faster than 32)
#endif
#ifndef BSMI_X3_64_BM
#define BSMI_X3_64_BM 256  // rows of the 64-column fused tile (dev builds: 512)
#endif
#ifndef BSMI_X3_320_MS
#define BSMI_X3_320_MS 16  // MFMA shape of the fused 256 x 320 tile (dev builds: 32)
#endif
template <typename T>
static int launch_cfg(const ConvArgs& a, TileCfg cfg, hipStream_t stream, float* sk_ws, int sk_grid) {
  if constexpr (IsFused<T>::value) {
    // conv_x3_body: 16x16x32 where the wave tile allows (it needs two halves of B fragments), 32x32x16 for the wide wave tiles
    switch (cfg) {
      case TILE_256x32: return launch_one<T, 256, 32, 4, 1, 16>(a, stream, sk_ws, sk_grid);
      case TILE_256x64: return launch_one<T, BSMI_X3_64_BM, 64, 4, 1, BSMI_X3_64_MS>(a, stream, sk_ws, sk_grid);  // 8 x 1 waves: measured no faster
      case TILE_256x160: return launch_one<T, 256, 160, 4, 1, BSMI_X3_64_MS>(a, stream, sk_ws, sk_grid);
      case TILE_256x320: {
        // one tile column whose last 16 columns are padding (N = 300): the second column half runs 9 blocks of 16 instead of 10
        static const bool cut_on = [] { const char* e = getenv("BSMI_X3_CUT"); return !(e && e[0] == '0'); }();
        if (BSMI_X3_320_MS == 16 && cut_on && a.Npad == 320 && a.Co <= 304) return launch_one<T, 256, 320, 4, 2, 16, 1>(a, stream, sk_ws, sk_grid);
        return launch_one<T, 256, 320, 4, 2, BSMI_X3_320_MS>(a, stream, sk_ws, sk_grid);
      }
      case TILE_256x256: return launch_one<T, 256, 256, 4, 2, 16>(a, stream, sk_ws, sk_grid);
      default: BSMI_FAIL(BSMI_ERR_INVALID, "unknown tile config %d", (int)cfg);
    }
  } else {
  switch (cfg) {
    case TILE_256x32: return launch_one<T, 256, 32, 4, 1>(a, stream, sk_ws, sk_grid);
    case TILE_256x64: return launch_one<T, 256, 64, 4, 1>(a, stream, sk_ws, sk_grid);
    case TILE_256x160: return launch_one<T, 256, 160, 4, 1>(a, stream, sk_ws, sk_grid);
    case TILE_256x320: return launch_one<T, 256, 320, 4, 2>(a, stream, sk_ws, sk_grid);
    case TILE_256x256:
      if (two_waves_per_simd()) {
        if constexpr (!std::is_same<T, float>::value)
          if (mfma_16x16()) return launch_one<T, 256, 256, 4, 2, 16>(a, stream, sk_ws, sk_grid);
        return launch_one<T, 256, 256, 4, 2>(a, stream, sk_ws, sk_grid);
      }
      return launch_one<T, 256, 256, 2, 2>(a, stream, sk_ws, sk_grid);
    default: BSMI_FAIL(BSMI_ERR_INVALID, "unknown tile config %d", (int)cfg);
  }
  }
}

int launch_conv_igemm(const ConvArgs& a, int precision, TileCfg cfg, hipStream_t stream, float* sk_ws, int sk_grid) {
  if (a.M <= 0 || a.nsteps <= 0 || a.Npad % tile_bn(cfg) != 0)
    BSMI_FAIL(BSMI_ERR_INVALID, "conv launch: bad geometry M=%d nsteps=%d Npad=%d", a.M, a.nsteps, a.Npad);
  if (precision == BSMI_PREC_F32) return launch_cfg<float>(a, cfg, stream, sk_ws, sk_grid);
  if (precision == BSMI_PREC_BF16) return launch_cfg<bf16_elem>(a, cfg, stream, sk_ws, sk_grid);
  if (precision == BSMI_PREC_BF16X3) {
    if (a.w_lo) {  // logical K-steps, hi and lo weight images: the fused form
      if (!two_waves_per_simd()) BSMI_FAIL(BSMI_ERR_STATE, "the fused split-bf16 kernels are 8-wave kernels (BSMI_WAVES8=0 is set)");
      return launch_cfg<bf16f_elem>(a, cfg, stream, sk_ws, sk_grid);
    }
    return launch_cfg<bf16s_elem>(a, cfg, stream, sk_ws, sk_grid);  // every K-step listed three times
  }
  BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
}

}  // namespace bsmi
