// Segmentation half of the hot path on gfx950: seeded-watershed fragments
// (reference post/ws.py:8-112) and mean-affinity hierarchical agglomeration (reference call
// site post/watershed.py:333-338; algorithm specified in oracle/seg_ref.c).
//
// Everything is integer work held bit-exact to the oracle:
//   fragments (fragments_in_xy): one workgroup per z-slice computes the foreground mask
//     (a_y + a_x >= 256), the exact squared EDT, the separable reflect-border max filter,
//     the maxima, their 4-connected components numbered in raster order; a tiny scan gives
//     the running id offset of ws.py:74-92; then one wave per slice replays skimage's
//     priority flood exactly (binary heap keyed (value, age), label-at-push) with the heap in
//     LDS -- the flood order is inherently sequential per slice, the parallelism is across
//     the slices (128 per block, 65,536 per 1024^3 volume).
//   agglomeration: region-adjacency graph by parallel scan + device hash table (sum, count
//     per edge), then one wave per volume replays the specified sequential merge loop
//     (min-queue over the total order (score, initial edge key)), then a parallel relabel.
#include <atomic>
#include <vector>
#include <new>

#include <hipcub/hipcub.hpp>
#include <thread>
#include <chrono>

#include "common.h"

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

namespace bsmi {
// agglo_host.cpp: the merge loop of the histogram-quantile scorers
void host_agglomerate_hist(uint32_t nn, uint32_t ne, const uint32_t* eu, const uint32_t* ev, uint32_t* hist, int quantile,
                           int init_with_max, const float* thresholds, int nthr, uint32_t* roots_out);
}

namespace bsmi {

// ------------------------------------------------------------------------------------------
// watershed fragments
// ------------------------------------------------------------------------------------------
constexpr int WS_T = 1024;           // threads per slice workgroup (seeds kernel): 16 waves, so that its LDS loops hide their latency
#ifndef BSMI_FLOOD_WAVES
#define BSMI_FLOOD_WAVES 8
#endif
// slices per flood workgroup (one wave each).  8, i.e. 20 workgroups of 64 KB of LDS per 160-slice block, two to a CU: with the
// stages run one after the other (the default) the chip is the lanes' alone, and spread over twice the CUs the floods of 20
// blocks side by side finish in 32 ms where 16 waves per workgroup took 41 (4 waves: as 8).  16 suited the overlapped mode, where
// a flood workgroup keeps a convolution workgroup off its CU.
constexpr int FLOOD_WAVES = BSMI_FLOOD_WAVES;
constexpr int FLOOD_LDS_HEAP = 1024; // heap entries per slice kept in LDS (8 B each); the rest spills to HBM

__device__ __forceinline__ int reflect_dup(int i, int n) {
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i < n ? i : p - 1 - i;
}
// the same for -n <= i < 2 n (a filter window no wider than the axis), without the division
__device__ __forceinline__ int reflect_near(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

// scratch per slice (global memory, L2 resident): mask u8, g/d2/mf int32, parent int32, lab int32
struct WsScratch {
  uint8_t* mask;
  int32_t* g;
  int32_t* d2;
  int32_t* mf;
  int32_t* par;
  int32_t* lab;     // local seed labels (1..n), later flood labels
  int32_t* nseeds;  // [D]
  uint64_t* offs;   // [D] exclusive scan of nseeds
  int32_t* seedlab; // optional (return_seeds): the seed labels before masking (ws.py:20-23), slice-local numbering
};

// LDS = true: the row distances / filter intermediates (uint16) and the squared distances
// (int32) of the slice live in LDS ([H][W+2] uint16 + [H*W] int32, <= 160 KiB for slices up to
// 160 x 160); LDS = false: same algorithm on the global scratch arrays (any slice size).
template <bool LDS>
__global__ __launch_bounds__(WS_T) void ws_seeds_kernel(const uint8_t* __restrict__ affs, int D, int H, int W,
                                                        int msd, WsScratch s, int compact) {
  extern __shared__ __attribute__((aligned(16))) char ws_smem[];
  const int z = blockIdx.x;
  const int n = H * W;
  const size_t vol = (size_t)D * n;
  const uint8_t* ay = affs + vol + (size_t)z * n;
  const uint8_t* ax = affs + 2 * vol + (size_t)z * n;
  uint8_t* mask = s.mask + (size_t)z * n;
  int32_t* g = s.g + (size_t)z * n;
  int32_t* d2 = s.d2 + (size_t)z * n;
  int32_t* mf = s.mf + (size_t)z * n;
  int32_t* par = s.par + (size_t)z * n;
  int32_t* lab = s.lab + (size_t)z * n;
  const int tid = threadIdx.x;
  __shared__ int sh_any_bg;
  __shared__ int sh_wave[WS_T / 64];
  const bool near = msd <= H && msd <= W;  // the maximum filter's window reflects at most once
  if (tid == 0) sh_any_bg = 0;
  __syncthreads();
  // The sequential per-row loops below (row distances, run labelling, numbering) must not walk global memory: a load
  // per step and row, 160 steps, at the L2 latency of a busy chip made a slice's workgroup -- which holds a whole CU's LDS --
  // live 0.6 ms, 6.4 ms per block under 16 lanes.  So the LDS path keeps what those loops read in LDS: the mask, and a
  // flag byte per voxel (is a maximum / is a root).  LDS: sg u16 [H][W+2] | sd2 u16 [H*W] | smask u8 [H*W] | sflag u8 [H*W].
  uint8_t* smask = nullptr;
  uint8_t* sflag = nullptr;
  const uint16_t* sd2_lds = nullptr;
  // compact (LDS path only): the flood's state of a voxel is ONE 32-bit record in `lab` -- marker label | squared distance << 16,
  // inside the mask <=> distance > 0 -- instead of three arrays (see ws_flood_kernel); mask and d2 then stay out of HBM
  if constexpr (!LDS) compact = 0;
  if constexpr (LDS) {
    const size_t o_sd2 = ((size_t)H * (W + 2) * 2 + 15) & ~(size_t)15;
    sd2_lds = (const uint16_t*)(ws_smem + o_sd2);
    const size_t o_mask = (o_sd2 + (size_t)n * 2 + 15) & ~(size_t)15;
    smask = (uint8_t*)(ws_smem + o_mask);
    sflag = smask + (((size_t)n + 15) & ~(size_t)15);
  }
  // a. mask  (0.5*(a_y+a_x) > 0.5*255  <=>  a_y + a_x >= 256)
  int bg = 0;
  for (int i = tid; i < n; i += WS_T) {
    const int m = (int)ay[i] + (int)ax[i] >= 256;
    if (!compact) mask[i] = (uint8_t)m;
    if constexpr (LDS) smask[i] = (uint8_t)m;
    bg |= !m;
  }
  if (bg) sh_any_bg = 1;
  __syncthreads();
  const int any_bg = sh_any_bg;
  constexpr int INF = 1 << 28;
  if constexpr (LDS) {
    // ---- LDS path: sg = row distances, later the x-filtered d2 (all values < 65535) ----------
    const int Wp = W + 2;  // row stride in uint16: consecutive rows fall into different banks
    uint16_t* sg = (uint16_t*)ws_smem;
    uint16_t* sd2 = (uint16_t*)(ws_smem + (((size_t)H * Wp * 2 + 15) & ~(size_t)15));  // squared distances < 65535 (launcher)
    constexpr int GINF = 0xffff;
    if (any_bg) {
      for (int y = tid; y < H; y += WS_T) {
        int last = -INF;
        for (int x = 0; x < W; ++x) {
          if (!smask[y * W + x]) last = x;
          sg[y * Wp + x] = (uint16_t)(last <= -INF ? GINF : x - last);
        }
        last = INF;
        for (int x = W - 1; x >= 0; --x) {
          if (!smask[y * W + x]) last = x;
          const int d = last >= INF ? GINF : last - x;
          if (d < (int)sg[y * Wp + x]) sg[y * Wp + x] = (uint16_t)d;
        }
      }
      __syncthreads();
      for (int i = tid; i < n; i += WS_T) {
        const int y = i / W, x = i - y * W;
        // min over rows of g(row, x)^2 + (y - row)^2, outwards from the own row: a row k away cannot improve on a best <= k^2
        const int g0 = sg[y * Wp + x];
        int best = g0 == GINF ? INF : g0 * g0;
        for (int k = 1; k < H && k * k < best; ++k) {
          if (y - k >= 0) {
            const int gg = sg[(y - k) * Wp + x];
            const int v = gg == GINF ? INF : gg * gg + k * k;
            best = v < best ? v : best;
          }
          if (y + k < H) {
            const int gg = sg[(y + k) * Wp + x];
            const int v = gg == GINF ? INF : gg * gg + k * k;
            best = v < best ? v : best;
          }
        }
        sd2[i] = (uint16_t)best;
        if (!compact) d2[i] = best;
      }
    } else {
      for (int i = tid; i < n; i += WS_T) {
        const int y = i / W, x = i - y * W;
        const int v = (y + 1) * (y + 1) + x * x;
        sd2[i] = (uint16_t)v;
        if (!compact) d2[i] = v;
      }
    }
    __syncthreads();
    const int left = msd / 2, right = msd - 1 - msd / 2;
    for (int i = tid; i < n; i += WS_T) {
      const int y = i / W, x = i - y * W;
      int m = INT32_MIN;
      for (int k = x - left; k <= x + right; ++k) {
        const int v = sd2[y * W + (near ? reflect_near(k, W) : reflect_dup(k, W))];
        m = v > m ? v : m;
      }
      sg[y * Wp + x] = (uint16_t)m;
    }
    __syncthreads();
    for (int i = tid; i < n; i += WS_T) {
      const int y = i / W, x = i - y * W;
      int m = INT32_MIN;
      for (int k = y - left; k <= y + right; ++k) {
        const int v = sg[(near ? reflect_near(k, H) : reflect_dup(k, H)) * Wp + x];
        m = v > m ? v : m;
      }
      sflag[i] = (uint8_t)(m == (int)sd2[i]);   // d. is this voxel a maximum of the filtered distance?
    }
    __syncthreads();
  } else {
  if (any_bg) {
    // b1. per row: distance along x to the nearest background voxel
    for (int y = tid; y < H; y += WS_T) {
      int last = -INF;
      for (int x = 0; x < W; ++x) {
        if (!mask[y * W + x]) last = x;
        g[y * W + x] = last <= -INF ? INF : x - last;
      }
      last = INF;
      for (int x = W - 1; x >= 0; --x) {
        if (!mask[y * W + x]) last = x;
        const int d = last >= INF ? INF : last - x;
        if (d < g[y * W + x]) g[y * W + x] = d;
      }
    }
    __syncthreads();
    // b2. per voxel: min over y' of g(y',x)^2 + (y-y')^2   (exact, integers)
    for (int i = tid; i < n; i += WS_T) {
      const int y = i / W, x = i - y * W;
      int best = INF;
      for (int yy = 0; yy < H; ++yy) {
        const int gg = g[yy * W + x];
        if (gg < INF) {
          const int dy = y - yy;
          const int v = gg * gg + dy * dy;
          best = v < best ? v : best;
        }
      }
      d2[i] = best;
    }
  } else {
    // scipy's behaviour without any background voxel: as if the only one sat at (-1, 0)
    for (int i = tid; i < n; i += WS_T) {
      const int y = i / W, x = i - y * W;
      d2[i] = (y + 1) * (y + 1) + x * x;
    }
  }
  __syncthreads();
  // c. maximum_filter(size=msd), window [i - msd/2, i + msd - 1 - msd/2], reflect border
  const int left = msd / 2, right = msd - 1 - msd / 2;
  for (int i = tid; i < n; i += WS_T) {
    const int y = i / W, x = i - y * W;
    int m = INT32_MIN;
    for (int k = x - left; k <= x + right; ++k) {
      const int v = d2[y * W + (near ? reflect_near(k, W) : reflect_dup(k, W))];
      m = v > m ? v : m;
    }
    g[i] = m;
  }
  __syncthreads();
  for (int i = tid; i < n; i += WS_T) {
    const int y = i / W, x = i - y * W;
    int m = INT32_MIN;
    for (int k = y - left; k <= y + right; ++k) {
      const int v = g[(near ? reflect_near(k, H) : reflect_dup(k, H)) * W + x];
      m = v > m ? v : m;
    }
    mf[i] = m;
  }
  __syncthreads();
  }
  // d/e. maxima and their 4-connected components.  Rows are labelled as runs first (each
  // maximum points at the first voxel of its run), then vertically adjacent runs are united
  // once, at the first column where they overlap (union-find, smaller index wins).
  for (int y = tid; y < H; y += WS_T) {
    int start = -1;
    for (int x = 0; x < W; ++x) {
      const int i = y * W + x;
      bool is_max;
      if constexpr (LDS) is_max = sflag[i] != 0;
      else is_max = mf[i] == d2[i];
      if (is_max) {
        if (start < 0) start = i;
        par[i] = start;
      } else {
        par[i] = -1;
        start = -1;
      }
    }
  }
  __syncthreads();
  auto find = [&](int a) {
    int p = par[a];
    while (p != a) {
      a = p;
      p = __hip_atomic_load(&par[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return a;
  };
  auto unite = [&](int a, int b) {
    for (;;) {
      a = find(a);
      b = find(b);
      if (a == b) return;
      if (a < b) { const int t = a; a = b; b = t; }
      const int old = atomicMin(&par[a], b);
      if (old == a) return;
      a = old;
    }
  };
  for (int i = tid; i < n; i += WS_T) {
    if (i < W || par[i] < 0 || par[i - W] < 0) continue;
    const int x = i % W;
    if (x > 0 && par[i - 1] >= 0 && par[i - W - 1] >= 0) continue;  // this run pair was united further left
    unite(i, i - W);
  }
  __syncthreads();
  // raster-order numbering of the roots (scipy.ndimage.label): chunked scan (LDS path: over root flags gathered by a
  // coalesced pass, not over the global parent array element by element)
  if constexpr (LDS) {
    for (int i = tid; i < n; i += WS_T) sflag[i] = (uint8_t)(par[i] == i);
    __syncthreads();
  }
  const int chunk = (n + WS_T - 1) / WS_T;
  const int c0 = tid * chunk, c1 = min(n, c0 + chunk);
  int cnt = 0;
  for (int i = c0; i < c1; ++i) {
    if constexpr (LDS) cnt += sflag[i];
    else cnt += (par[i] == i);
  }
  // exclusive scan of the chunk counts over the workgroup's threads: in a wave by shuffles, the 16 wave totals by wave 0
  const int lane = tid & 63, wave = tid >> 6;
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh_wave[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int w = lane < WS_T / 64 ? sh_wave[lane] : 0;
    for (int o = 1; o < WS_T / 64; o <<= 1) {
      const int t = __shfl_up(w, o);
      if (lane >= o) w += t;
    }
    if (lane < WS_T / 64) sh_wave[lane] = w;
  }
  __syncthreads();
  int id = (wave ? sh_wave[wave - 1] : 0) + incl - cnt;
  if (tid == WS_T - 1) s.nseeds[z] = id + cnt;
  for (int i = c0; i < c1; ++i) {
    bool root;
    if constexpr (LDS) root = sflag[i] != 0;
    else root = par[i] == i;
    if (root) g[i] = ++id;  // g reused: root index -> label
  }
  __syncthreads();
  // markers = label * mask (seeds outside the mask vanish inside skimage)
  for (int i = tid; i < n; i += WS_T) {
    int l = 0;
    if (par[i] >= 0) {
      const int sl = g[find(i)];
      if (s.seedlab) s.seedlab[(size_t)z * n + i] = sl;
      bool inside;
      if constexpr (LDS) inside = smask[i] != 0;
      else inside = mask[i] != 0;
      if (inside) l = sl;
    } else if (s.seedlab) {
      s.seedlab[(size_t)z * n + i] = 0;
    }
    if (compact) ((uint32_t*)lab)[i] = (uint32_t)l | ((uint32_t)sd2_lds[i] << 16);
    else lab[i] = l;
  }
}

__global__ void ws_offsets_kernel(int D, WsScratch s, uint64_t* max_id) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint64_t acc = 0;
    for (int z = 0; z < D; ++z) { s.offs[z] = acc; acc += (uint64_t)s.nseeds[z]; }
    *max_id = acc;
  }
}

// lane 0's 64-bit value to the whole wave
__device__ __forceinline__ uint64_t bcast0(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// heap entry: [63:40] = MAXD2 - d2 (24 bit) | [39:20] = age (20 bit) | [19:0] = voxel index.
// Ordering ignores the index bits (skimage compares (value, age) only).
// (The scalar unit, where the wave-uniform flood loop runs, has no 64-bit ordered compare; a vector compare and the trip of its
// result back to a scalar register are ten instructions, and the floods of a stage's blocks side by side are bound by
// instruction issue.  The sign of the difference of the two 44-bit keys is scalar work: two shifts, a subtract with borrow.)
__device__ __forceinline__ bool flood_smaller(uint64_t a, uint64_t b) { return (int64_t)((a >> 20) - (b >> 20)) < 0; }

// COMPACT: a voxel's state is the 32-bit record ws_seeds_kernel leaves in `lab` (label | squared distance << 16; in the mask <=>
// distance > 0): a pop touches three cache lines (the rows above, of and below the voxel) instead of nine, and a slice is 100 KB
// instead of 230 -- side by side, the floods of a stage's blocks are bound by the lines they pull through L2, not by one wave's
// latency chain.
template <bool COMPACT>
__global__ __launch_bounds__(64 * FLOOD_WAVES) void ws_flood_kernel(int D, int H, int W, WsScratch s, uint64_t* heap_spill,
                                                     size_t spill_stride, uint64_t* __restrict__ frags, int* status) {
  __shared__ uint64_t hl_all[FLOOD_WAVES][FLOOD_LDS_HEAP];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.x * FLOOD_WAVES + wave;
  if (z >= D) return;  // whole wave exits; no workgroup barrier is used below
  uint64_t* hl = hl_all[wave];
  const int n = H * W;
  const uint8_t* mask = s.mask + (size_t)z * n;
  const int32_t* d2 = s.d2 + (size_t)z * n;
  int32_t* lab = s.lab + (size_t)z * n;
  uint64_t* hg = heap_spill + (size_t)z * spill_stride;  // entries >= FLOOD_LDS_HEAP live here
  // The queue order is inherently sequential, so the whole wave walks the same loop in lockstep (every lane holds the
  // same `items`, `age` and heap values; lane 0 alone owns the heap).  What the other lanes buy: the four neighbours
  // of a popped voxel are fetched side by side -- lane k reads mask, label and distance of neighbour k in one round of
  // loads -- where a single lane would chain up to twelve dependent global loads per voxel.  Label stores are issued by
  // all four fetching lanes (same address, same value), so that each lane's later loads follow its own stores in program
  // order.
  // The compiler must KNOW that the loop is uniform: every value that comes out of memory goes through v_readfirstlane /
  // v_readlane (bcast0, uni), so that counters, heap indices and comparison results live in scalar registers and the loops
  // branch on the scalar unit.  Left to its divergence analysis it guarded each `if` of the sift loops with exec-mask
  // save / restore sequences -- about 100 instructions per heap level, 1000 per pop -- and the floods of a stage's blocks,
  // three waves to a SIMD, paid for it in instruction issue (20 blocks side by side: the last flood ends after 25.6 ms instead of 29.3).
  {
    constexpr uint64_t MAXD2 = (1u << 24) - 1;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // queue entry: [63:40] MAXD2 - d2 | [39:20] age | [19:0] voxel.  COMPACT (d2 < 2^16, fewer than 2^15 voxels, so fewer pushes):
    // [63:48] 65535 - d2 | [47:32] age | [31:16] label | [15:0] voxel -- the same order, the whole key is the upper word (one
    // scalar compare), and the label the voxel was given travels with it: a pop does not read the voxel's own record
    auto entry = [](uint64_t dd, uint32_t age_, uint64_t q_, uint32_t lbl_) -> uint64_t {
      return COMPACT ? ((65535ull - dd) << 48) | ((uint64_t)age_ << 32) | ((uint64_t)lbl_ << 16) | q_
                     : ((MAXD2 - dd) << 40) | ((uint64_t)age_ << 20) | q_;
    };
    auto smaller = [](uint64_t a_, uint64_t b_) -> bool {
      if constexpr (COMPACT) return (uint32_t)(a_ >> 32) < (uint32_t)(b_ >> 32);
      else return flood_smaller(a_, b_);
    };
    int items = 0;
    // the heap is written by lane 0 alone; its HBM spill is also read by lane 0 alone (and broadcast), so that those loads
    // follow that lane's stores in its own program order; LDS operations of a wave execute in order anyway
    // (Tried: the first 64 entries -- the six top levels every sift-down walks -- in registers, entry i in lane i, read with
    // v_readlane: fragments of a 128^3 block 12.7 -> 12.6 ms.  Also tried: label, distance and mask bit of a voxel packed into
    // one 8-byte record, five loads per pop instead of thirteen: 12.6 -> 11.8 ms alone, nothing under the pipeline's 16 lanes.)
    auto hget = [&](int i) -> uint64_t {
      if (i < FLOOD_LDS_HEAP) return hl[i];
      uint64_t v = 0;
      if (lane == 0) v = hg[i - FLOOD_LDS_HEAP];
      return bcast0(v);
    };
    auto hset = [&](int i, uint64_t v) {
      if (lane == 0) {
        if (i < FLOOD_LDS_HEAP) hl[i] = v; else hg[i - FLOOD_LDS_HEAP] = v;
      }
    };
    auto push = [&](uint64_t it) {
      int c = uni(items++);
      if constexpr (COMPACT) {
        if (c < FLOOD_LDS_HEAP) {  // (as below, with the key and the voxel of an entry as two 32-bit values; loads from a uniform LDS address are uniform to the compiler: no v_readfirstlane needed)
          const uint32_t ik = (uint32_t)(it >> 32), ix = (uint32_t)it;
          while (c > 0) {
            const int p = (c - 1) >> 1;
            const uint64_t pr = hl[p];
            const uint32_t pk = (uint32_t)(pr >> 32), px = (uint32_t)pr;
            if (!(ik < pk)) break;
            hl[c] = ((uint64_t)pk << 32) | px;
            c = p;
          }
          hl[c] = ((uint64_t)ik << 32) | ix;
          return;
        }
      }
      if (c < FLOOD_LDS_HEAP) {  // the whole path to the root is in LDS: no range checks per level
        while (c > 0) {
          const int p = (c - 1) >> 1;
          const uint64_t pv = hl[p];
          if (!smaller(it, pv)) break;
          hl[c] = pv;
          c = p;
        }
        hl[c] = it;
        return;
      }
      while (c > 0) {
        const int p = (c + 1) / 2 - 1;
        const uint64_t pv = hget(p);
        if (smaller(it, pv)) { hset(c, pv); c = uni(p); } else break;
      }
      hset(c, it);
    };
    // seeds in raster order, age 0
    uint32_t* rec = (uint32_t*)lab;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int li = i < n ? (COMPACT ? (int)(rec[i] & 0xffffu) : lab[i]) : 0;
      unsigned long long seeds = __ballot(li != 0);
      while (seeds) {
        const int k = __ffsll(seeds) - 1;
        seeds &= seeds - 1;
        const int j = i0 + k;
        const uint32_t rj = COMPACT ? (uint32_t)uni((int)rec[j]) : 0u;
        const uint64_t dj = COMPACT ? (uint64_t)(rj >> 16) : (uint64_t)(uint32_t)uni(d2[j]);
        push(entry(dj, 0u, (uint64_t)j, rj & 0xffffu));
      }
    }
    uint32_t age = 0;
    const int k4 = lane & 3;
    const int dq = k4 == 0 ? -W : (k4 == 1 ? -1 : (k4 == 2 ? 1 : W));
    // y = idx / W without the division (idx < 2^20): exact for W < 4096 with the rounded-up reciprocal
    const bool rcp_ok = W > 1 && W < 4096;  // (W = 1: the reciprocal is 2^32)
    const uint32_t rcpW = (uint32_t)((((uint64_t)1 << 32) + (uint32_t)W - 1) / (uint32_t)W);
    // The pops run with lanes 0-3 alone (the four neighbour fetchers): inside, nothing is guarded by a lane test any more --
    // the heap writes of the LDS-only loops are issued by all active lanes (same address, same value) -- and the exec-mask
    // save / restore around every such write is gone.
    if (lane < 4)
    while (items > 0) {
      const uint64_t e = hget(0);
      --items;
      // the popped voxel's label and its neighbours' state are requested first: their latency hides behind the sift
      const int idx = COMPACT ? (int)((uint32_t)e & 0xffffu) : (int)(e & 0xfffffu);
      const int y = rcp_ok ? (int)__umulhi((uint32_t)idx, rcpW) : idx / W, x = idx - y * W;
      // neighbour order [-W, -1, +1, +W]: lane k < 4 looks at neighbour k; the other lanes stay out of global memory
      const bool okk = k4 == 0 ? y > 0 : (k4 == 1 ? x > 0 : (k4 == 2 ? x < W - 1 : y < H - 1));
      const int qk = okk ? idx + dq : idx;
      int lme = 0, mk = 0, lk = 0, dk = 0;
      {
        if constexpr (COMPACT) {
          const uint32_t rk = rec[qk];  // (a neighbour beyond the slice: the voxel's own record -- labelled, so no candidate)
          lme = (int)(((uint32_t)e >> 16) & 0xffffu);
          lk = (int)(rk & 0xffffu);
          dk = (int)(rk >> 16);
          mk = dk != 0;
        } else {
          lme = lab[idx];
          mk = mask[qk];
          lk = lab[qk];
          dk = d2[qk];
        }
      }
      if (items > 0) {
        // sift the last element down from the root (skimage heappop order)
        const uint64_t last = hget(items);
        int i = 0;
        // the levels whose two children both exist and live in LDS (all but the last one of a heap that fits): the smaller
        // child (the left one on a tie), then that one against `last` -- the same choice as the general form below makes
        const int lim = items < FLOOD_LDS_HEAP ? items : FLOOD_LDS_HEAP;
        bool placed = false;
        if constexpr (COMPACT) {  // (the loop below with the key and the voxel of an entry as two 32-bit values)
          const uint32_t lk = (uint32_t)(last >> 32);
          while (2 * i + 2 < lim) {
            const int c1 = 2 * i + 1;
            const uint64_t r1 = hl[c1], r2 = hl[c1 + 1];
            const uint32_t k1 = (uint32_t)(r1 >> 32), k2 = (uint32_t)(r2 >> 32);
            const uint32_t x1 = (uint32_t)r1, x2 = (uint32_t)r2;
            const bool right = k2 < k1;
            const uint32_t ck = right ? k2 : k1, cx = right ? x2 : x1;
            if (!(ck < lk)) { placed = true; break; }
            hl[i] = ((uint64_t)ck << 32) | cx;
            i = c1 + (right ? 1 : 0);
          }
        }
        while (!COMPACT && 2 * i + 2 < lim) {
          const int c1 = 2 * i + 1;
          const uint64_t r1 = hl[c1], r2 = hl[c1 + 1];
          const uint64_t v1 = r1, v2 = r2;
          const bool right = smaller(v2, v1);
          const uint64_t cv = right ? v2 : v1;
          if (!smaller(cv, last)) { placed = true; break; }
          hl[i] = cv;
          i = c1 + (right ? 1 : 0);
        }
        for (; !placed;) {
          const int c1 = 2 * i + 1, c2 = c1 + 1;
          if (c1 >= items) break;
          uint64_t v1, v2;
          if (c2 < FLOOD_LDS_HEAP) {  // both children with one LDS round trip (entry c2 = items is read and not looked at)
            const uint64_t r1 = hl[c1], r2 = hl[c2];
            v1 = r1;
            v2 = r2;
          } else {
            v1 = hget(c1);
            v2 = c2 < items ? hget(c2) : 0;
          }
          int sm = i;
          uint64_t smv = last;
          if (smaller(v1, smv)) { sm = c1; smv = v1; }
          if (c2 < items && smaller(v2, smv)) { sm = c2; smv = v2; }
          if (sm == i) break;
          hset(i, smv);
          i = uni(sm);
        }
        hset(i, last);
      }
      const int l = uni(lme);
      const bool cand = okk && mk && lk == 0;
      // the neighbours to take, in the order [-W, -1, +1, +W] (a voxel is taken once, so one per pop on average)
      for (uint32_t m = (uint32_t)__ballot(cand) & 0xfu; m; m &= m - 1) {
        const int k = __ffs((int)m) - 1;
        const int q = __builtin_amdgcn_readlane(qk, k);
        const uint64_t dd = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(dk, k);
        ++age;
        if constexpr (COMPACT) rec[q] = (uint32_t)l | ((uint32_t)dd << 16);
        else lab[q] = l;
        push(entry(dd, age, (uint64_t)q, (uint32_t)l));
      }
    }
  }
  // the label writes become visible to the whole wave before the copy-out
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint64_t off = s.offs[z];
  uint64_t* out = frags + (size_t)z * n;
  for (int i = lane; i < n; i += 64) {
    const int l = COMPACT ? (int)(((const uint32_t*)lab)[i] & 0xffffu) : lab[i];
    out[i] = l ? (uint64_t)l + off : 0ull;
  }
  (void)status;
}

// ------------------------------------------------------------------------------------------
// agglomeration
// ------------------------------------------------------------------------------------------
constexpr uint64_t HEMPTY = 0xffffffffffffffffull;
constexpr uint64_t HTOMB = 0xfffffffffffffffeull;
constexpr uint32_t NOEDGE = 0xffffffffu;
constexpr int AGG_LDS_HEAP = 12288;  // entries (8 B) of the merge queue kept in LDS

__device__ __forceinline__ uint64_t hmix(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

struct AggWs {
  // node table
  uint32_t* rank_of_id;  // [id_cap]: direct-address table id -> rank (after scan), 0xffffffff = absent
  uint32_t id_cap;
  uint64_t* ids;         // [node_cap] rank -> id
  uint32_t node_cap;
  uint32_t* counters;    // [0]=nn, [1]=ne, [2]=heap_n, [3]=overflow flags of the call in flight (zeroed by every call)
  uint32_t* sticky;      // [1] overflow flags of every call since the last bsmi_seg_status (the last kernel of a call ORs [3] in)
  // hash table over edges
  uint64_t* hkeys;       // [hcap]
  uint32_t* hvals;       // [hcap] -> edge index
  unsigned long long* hsum;  // [hcap] (during build)
  uint32_t* hcnt;        // [hcap]
  uint32_t hcap;         // power of two
  // edge arrays [edge_cap]
  uint32_t* eu; uint32_t* ev; uint64_t* ekey0; unsigned long long* esum; uint32_t* ecnt;
  uint32_t* enextu; uint32_t* enextv; uint8_t* eflags;  // bit0 deleted, bit1 stale
  uint32_t edge_cap;
  uint32_t* head;        // [node_cap]
  uint32_t* parent;      // [node_cap]
  uint32_t* roots;       // [nthr_cap][node_cap]
  uint64_t* heap_spill;  // [edge_cap]
  uint64_t* maxid;       // [1]
  // RAG scoring path (arbitrary 64-bit ids): id hash, sorted edge numbering, bin queue, merge tree
  uint64_t* idkeys;      // [icap] open-addressing set of fragment ids
  uint32_t* idvals;      // [icap] -> rank
  uint32_t icap;         // power of two
  uint64_t* idu;         // [node_cap] distinct ids before sorting
  uint64_t* skeys;       // [hcap] edge keys sorted
  uint32_t* sslot;       // [hcap] hash slot of the sorted key
  uint32_t* iota;        // [hcap]
  uint32_t* qnext;       // [edge_cap] FIFO links of the bin queue
  uint32_t* tnext;       // [2 * node_cap] merge tree parent
  float* tscore;         // [2 * node_cap]
  uint32_t* cur;         // [node_cap] tree node of a cluster root
  uint32_t* ha; uint32_t* hb;  // [node_cap] merge history (ranks)
  float* escore;         // [edge_cap] stored score: the score at the edge's last (re)scoring = its place in the queue
  uint32_t* etime;       // [edge_cap] RAG path: merge clock at the edge's last scoring
  uint32_t* ntime;       // [node_cap] RAG path: merge clock when the node last survived a merge (edges scored before are stale)
  int xcd_hint;          // XCD the sequential merge loop of this workspace should run on (see xcd_claim)
  // label table (bs refine): per id-hash slot
  unsigned long long* tcount;  // [icap]
  int* tzmin; int* tzmax;      // [icap]
};

__global__ void agg_maxid_kernel(const uint64_t* __restrict__ frags, size_t n, AggWs w) {
  unsigned long long m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = frags[i] > m ? frags[i] : m;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_down(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax((unsigned long long*)w.maxid, m);
}

__global__ void agg_mark_kernel(const uint64_t* __restrict__ frags, size_t n, AggWs w) {
  const uint64_t maxid = *w.maxid;
  if (maxid >= w.id_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&w.counters[3], 1u);
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[i];
    if (f) w.rank_of_id[f] = 1u;  // benign race: every writer stores 1
  }
}

// single workgroup: exclusive scan of the presence flags -> ranks (ascending id = sorted order)
__global__ __launch_bounds__(1024) void agg_rank_kernel(AggWs w) {
  __shared__ uint32_t sh[1024];
  if (w.counters[3]) return;
  const uint64_t maxid = *w.maxid;
  const uint32_t n = (uint32_t)maxid + 1;
  const uint32_t chunk = (n + 1023) / 1024;
  const uint32_t c0 = threadIdx.x * chunk, c1 = min(n, c0 + chunk);
  uint32_t cnt = 0;
  for (uint32_t i = c0; i < c1 && i < n; ++i) cnt += w.rank_of_id[i] == 1u;
  sh[threadIdx.x] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int t = 0; t < 1024; ++t) { const uint32_t c = sh[t]; sh[t] = acc; acc += c; }
    w.counters[0] = acc;
    if (acc > w.node_cap) atomicOr(&w.counters[3], 2u);
  }
  __syncthreads();
  if (w.counters[3]) return;
  uint32_t r = sh[threadIdx.x];
  for (uint32_t i = c0; i < c1 && i < n; ++i) {
    if (w.rank_of_id[i] == 1u) {
      w.rank_of_id[i] = r;
      w.ids[r] = i;
      w.head[r] = NOEDGE;
      w.parent[r] = r;
      ++r;
    } else {
      w.rank_of_id[i] = 0xffffffffu;
    }
  }
}

template <bool HASH>
__device__ __forceinline__ uint32_t agg_rank(const AggWs& w, uint64_t f) {
  if constexpr (!HASH) return w.rank_of_id[f];
  uint32_t s = (uint32_t)hmix(f) & (w.icap - 1);
  for (uint32_t probe = 0; probe < w.icap; ++probe) {
    const uint64_t k = w.idkeys[s];
    if (k == f) return w.idvals[s];
    if (k == HEMPTY) break;
    s = (s + 1) & (w.icap - 1);
  }
  return 0;  // unreachable: every voxel id was inserted by rag_ids_kernel
}

template <bool HASH>
__global__ void agg_edges_kernel(const uint8_t* __restrict__ affs, const uint64_t* __restrict__ frags, int D, int H,
                                 int W, AggWs w) {
  if (w.counters[3]) return;
  const size_t n = (size_t)D * H * W;
  const size_t hw = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f1 = frags[p];
    if (!f1) continue;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int z = (int)(p / hw);
    const uint32_t r1 = agg_rank<HASH>(w, f1);
    const bool ok[3] = {z > 0, y > 0, x > 0};
    const size_t st[3] = {hw, (size_t)W, 1};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!ok[d]) continue;
      const uint64_t f2 = frags[p - st[d]];
      if (!f2 || f2 == f1) continue;
      const uint32_t r2 = agg_rank<HASH>(w, f2);
      const uint32_t u = r1 < r2 ? r1 : r2, v = r1 < r2 ? r2 : r1;
      const uint64_t key = ((uint64_t)u << 32) | v;
      uint32_t slot = (uint32_t)hmix(key) & (w.hcap - 1);
      bool placed = false;
      for (uint32_t probe = 0; probe < w.hcap; ++probe) {
        const unsigned long long old = atomicCAS((unsigned long long*)&w.hkeys[slot], HEMPTY, key);
        if (old == HEMPTY || old == key) { placed = true; break; }
        slot = (slot + 1) & (w.hcap - 1);
      }
      if (!placed) { atomicOr(&w.counters[3], 4u); return; }
      atomicAdd(&w.hsum[slot], (unsigned long long)affs[(size_t)d * n + p]);
      atomicAdd(&w.hcnt[slot], 1u);
    }
  }
}

__global__ void agg_compact_kernel(AggWs w) {
  if (w.counters[3]) return;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < w.hcap; s += gridDim.x * blockDim.x) {
    const uint64_t key = w.hkeys[s];
    if (key == HEMPTY) continue;
    const uint32_t e = atomicAdd(&w.counters[1], 1u);
    if (e >= w.edge_cap) { atomicOr(&w.counters[3], 8u); continue; }
    const uint32_t u = (uint32_t)(key >> 32), v = (uint32_t)key;
    w.eu[e] = u; w.ev[e] = v; w.ekey0[e] = key;
    w.esum[e] = w.hsum[s]; w.ecnt[e] = w.hcnt[s];
    w.eflags[e] = 0;
    w.hvals[s] = e;
    w.enextu[e] = atomicExch(&w.head[u], e);
    w.enextv[e] = atomicExch(&w.head[v], e);
  }
}

// Histogram-quantile scorers (reference post/watershed.py:230-243): the 256-bin histogram of every edge's affinities, in a
// second scan once the edges are numbered (hist [ne][256]; the merge loop of these scorers runs on the host, agglo_host.cpp).
__global__ void agg_hist_kernel(const uint8_t* __restrict__ affs, const uint64_t* __restrict__ frags, int D, int H, int W, AggWs w,
                                uint32_t* __restrict__ hist) {
  if (w.counters[3]) return;
  const size_t n = (size_t)D * H * W;
  const size_t hw = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f1 = frags[p];
    if (!f1) continue;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int z = (int)(p / hw);
    const uint32_t r1 = w.rank_of_id[f1];
    const bool ok[3] = {z > 0, y > 0, x > 0};
    const size_t st[3] = {hw, (size_t)W, 1};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!ok[d]) continue;
      const uint64_t f2 = frags[p - st[d]];
      if (!f2 || f2 == f1) continue;
      const uint32_t r2 = w.rank_of_id[f2];
      const uint32_t u = r1 < r2 ? r1 : r2, v = r1 < r2 ? r2 : r1;
      const uint64_t key = ((uint64_t)u << 32) | v;
      uint32_t slot = (uint32_t)hmix(key) & (w.hcap - 1);
      while (w.hkeys[slot] != key) slot = (slot + 1) & (w.hcap - 1);  // present: agg_edges_kernel inserted every pair
      atomicAdd(&hist[(size_t)w.hvals[slot] * 256 + affs[(size_t)d * n + p]], 1u);
    }
  }
}

// Ties of the merge queue are broken by the edge's initial key (oracle/seg_ref.c).  The mean affinities of uint8 sums tie
// often, and looking the two keys up costs the single-lane loop two trips to L2 per comparison: 3.3 us per pop.  So the
// edges are ranked by their key once, here (one workgroup, keys in LDS, rank = number of smaller keys), and the queue
// entries carry the rank: a comparison is then one 64-bit compare.  Graphs of more than kRankMax edges keep the look-up.
// erank = w.qnext (an array the mean-agglomeration path does not use otherwise); counters[7] = ranks valid.
constexpr uint32_t kRankMax = AGG_LDS_HEAP;  // ranked <=> the merge loop's FAST form (queue, flags and ranks fit the LDS)
__global__ __launch_bounds__(1024) void agg_edge_rank_kernel(AggWs w) {
  extern __shared__ uint64_t rank_keys[];
  if (w.counters[3]) return;
  const uint32_t ne = w.counters[1];
  if (ne > kRankMax || ne > w.edge_cap) return;
  for (uint32_t e = threadIdx.x; e < ne; e += blockDim.x) rank_keys[e] = w.ekey0[e];
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < ne; e += blockDim.x) {
    const uint64_t k = rank_keys[e];
    uint32_t r = 0;
    for (uint32_t j = 0; j < ne; ++j) r += rank_keys[j] < k ? 1u : 0u;
    w.qnext[e] = r;
  }
  if (threadIdx.x == 0) w.counters[7] = 1;
}

__device__ __forceinline__ float agg_score(unsigned long long sum, uint32_t cnt) {
  return 1.0f - (float)((double)sum / (255.0 * (double)cnt));
}

// The sequential merge loops hold one CU (its LDS) for tens of milliseconds.  A one-workgroup launch always lands on
// the same XCD, so the eight lanes of the block pipeline would take eight CUs of ONE XCD away from the U-Net's
// persistent conv workgroups (measured with dummy kernels: 8 x 98 KB of LDS held that way cost the predict stream
// 13 %, one CU in each XCD 3 %).  So the loops are launched as 8 workgroups, which the dispatcher deals round-robin
// to the 8 XCDs, and exactly one of them -- the one on the workspace's XCD if there is one, else the last to
// arrive -- does the work; the others leave at once.  claim[0] = taken, claim[1] = arrivals (zero before the launch).
__device__ __forceinline__ bool xcd_claim(uint32_t* claim, int target) {
  __shared__ int sh_run;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    bool run = false;
    if ((int)(xcc & 7) == target) run = atomicCAS(&claim[0], 0u, 1u) == 0u;
    const unsigned arrived = atomicAdd(&claim[1], 1u);
    if (!run && arrived == gridDim.x - 1) run = atomicCAS(&claim[0], 0u, 1u) == 0u;
    sh_run = run ? 1 : 0;
  }
  __syncthreads();
  return sh_run != 0;
}

// One wave per volume; lane 0 replays the sequential merge loop of oracle/seg_ref.c (waterz mergeUntil / mergeRegions:
// on a shared neighbour the dearer of the two parallel edges, by STORED score, is merged into the cheaper one, which
// keeps its place in the queue).  mergeRegions also marks every edge incident to the survivor stale; with this queue --
// a total order on (stored score, initial key) -- rescoring an edge whose sums did not change puts it back exactly
// where it was, so that marking cannot be observed and is not replayed here (the bin queue of rag_merge_kernel, where
// a re-insertion moves the edge to the back of its bin, does replay it).
struct AggThresholds {  // by value: no host-to-device copy per call
  float v[16];
};

// FAST: the graph fits -- ne <= AGG_LDS_HEAP edges, ranked by agg_edge_rank_kernel.  The queue never holds more entries than
// there are edges, so every queue access is a plain LDS access (with the HBM overflow in the same expression the compiler
// selects between the two addresses and emits FLAT loads: 0.25 us per sift level, 3 us per pop); an entry carries
// rank << 16 | edge below the score, so a pop needs no look-up; and the edge flags live in LDS.
template <bool FAST>
__device__ __forceinline__ void agg_merge_body(const AggWs& w, const AggThresholds& thr_arg, int nthr, uint64_t* hl, uint8_t* fl_lds,
                                               int* sh_dummy_p, uint32_t nn, uint32_t ne) {
  const float* thresholds = thr_arg.v;
  int& sh_dummy = *sh_dummy_p;
  if constexpr (FAST) {
    for (uint32_t i = threadIdx.x; i < ne; i += 64) fl_lds[i] = 0;  // agg_compact_kernel left every flag at 0
    __syncthreads();
  }
  auto fget = [&](uint32_t e) -> uint8_t {
    if constexpr (FAST) return fl_lds[e];
    else return w.eflags[e];
  };
  auto fset = [&](uint32_t e, uint8_t v) {
    if constexpr (FAST) fl_lds[e] = v;
    else w.eflags[e] = v;
  };
  const int lane = threadIdx.x;
  int items = 0;  // meaningful on lane 0 only
  const float tmax = thresholds[nthr - 1];
  // entry: [63:32] float bits of the score (scores are >= 0: bit pattern order == value order),
  //        [31:0] edge index; ties on the score are broken by the edge's initial key.
  auto hget = [&](int i) -> uint64_t {
    if constexpr (FAST) return hl[i];
    else return i < AGG_LDS_HEAP ? hl[i] : w.heap_spill[i - AGG_LDS_HEAP];
  };
  auto hset = [&](int i, uint64_t v) {
    if constexpr (FAST) hl[i] = v;
    else { if (i < AGG_LDS_HEAP) hl[i] = v; else w.heap_spill[i - AGG_LDS_HEAP] = v; }
  };
  // low word of an entry: FAST: rank by initial key << 16 | edge; else the edge, and ties look the keys up
  auto less = [&](uint64_t a, uint64_t b) -> bool {
    if constexpr (FAST) return a < b;
    const uint32_t sa = (uint32_t)(a >> 32), sb = (uint32_t)(b >> 32);
    if (sa != sb) return sa < sb;
    return w.ekey0[(uint32_t)a] < w.ekey0[(uint32_t)b];
  };
  auto entry = [&](float sc, uint32_t e) -> uint64_t {
    if constexpr (FAST) return ((uint64_t)__float_as_uint(sc) << 32) | ((uint64_t)w.qnext[e] << 16) | e;
    else return ((uint64_t)__float_as_uint(sc) << 32) | e;
  };
  auto edge_of = [&](uint64_t top) -> uint32_t {
    if constexpr (FAST) return (uint32_t)top & 0xffffu;
    else return (uint32_t)top;
  };
  auto sift_down = [&](int i, uint64_t val) {
    for (;;) {
      const int c1 = 2 * i + 1, c2 = c1 + 1;
      if (c1 >= items) break;
      int sm = c1;
      uint64_t smv = hget(c1);
      if (c2 < items) {
        const uint64_t v2 = hget(c2);
        if (less(v2, smv)) { sm = c2; smv = v2; }
      }
      if (!less(smv, val)) break;
      hset(i, smv);
      i = sm;
    }
    hset(i, val);
  };
  auto push = [&](uint64_t val) {
    int c = items++;
    while (c > 0) {
      const int p = (c - 1) / 2;
      const uint64_t pv = hget(p);
      if (less(val, pv)) { hset(c, pv); c = p; } else break;
    }
    hset(c, val);
  };
  bool fail = false;
  auto hfind = [&](uint64_t key) -> int64_t {
    uint32_t s = (uint32_t)hmix(key) & (w.hcap - 1);
    for (uint32_t probe = 0; probe < w.hcap; ++probe) {
      const uint64_t k = w.hkeys[s];
      if (k == key) return (int64_t)s;
      if (k == HEMPTY) return -1;
      s = (s + 1) & (w.hcap - 1);
    }
    return -1;
  };
  auto hput = [&](uint64_t key, uint32_t val) {
    uint32_t s = (uint32_t)hmix(key) & (w.hcap - 1);
    uint32_t probe = 0;
    for (; probe < w.hcap; ++probe) {
      const uint64_t k = w.hkeys[s];
      if (k == HEMPTY || k == HTOMB || k == key) break;
      s = (s + 1) & (w.hcap - 1);
    }
    if (probe == w.hcap) { fail = true; return; }
    w.hkeys[s] = key;
    w.hvals[s] = val;
  };
  auto norm_key = [](uint32_t x, uint32_t y) -> uint64_t {
    return x < y ? (((uint64_t)x << 32) | y) : (((uint64_t)y << 32) | x);
  };

  if (lane == 0) {
    // initial queue: only edges below the largest threshold can ever be popped
    for (uint32_t e = 0; e < ne; ++e) {
      const float sc = agg_score(w.esum[e], w.ecnt[e]);
      w.escore[e] = sc;
      if (sc < tmax) hset(items++, entry(sc, e));
    }
    for (int i = items / 2 - 1; i >= 0; --i) sift_down(i, hget(i));  // Floyd heapify
  }
  for (int t = 0; t < nthr; ++t) {
    if (lane == 0) {
      const float thr = thresholds[t];
      while (items > 0) {
        const uint64_t top = hget(0);
        if (!(__uint_as_float((uint32_t)(top >> 32)) < thr)) break;
        --items;
        if (items > 0) sift_down(0, hget(items));
        const uint32_t e = edge_of(top);
        const uint8_t fl = fget(e);
        if (fl & 1) continue;
        if (fl & 2) {
          fset(e, fl & ~2);
          const float sc = agg_score(w.esum[e], w.ecnt[e]);
          w.escore[e] = sc;
          if (sc < tmax) push(entry(sc, e));
          continue;
        }
        const uint32_t eu = w.eu[e], evv = w.ev[e];
        const uint32_t a = eu < evv ? eu : evv, b = eu < evv ? evv : eu;
        // Every access below is a dependent trip to L2 (~0.3 us), so the loads that do not depend on each other are
        // issued together: all fields of f at once, the first probes of both hash lookups at once, the sums at once.
        uint32_t f = w.head[b];
        while (f != NOEDGE && !fail) {
          const uint32_t fu = w.eu[f], fv = w.ev[f], fnu = w.enextu[f], fnv = w.enextv[f];
          const uint8_t ffl = fget(f);
          const bool b_in_u = fu == b;
          const uint32_t nxt = b_in_u ? fnu : fnv;
          if (f != e && !(ffl & 1)) {
            const uint32_t nb = b_in_u ? fv : fu;
            const uint64_t fkey = norm_key(fu, fv), gkey = norm_key(a, nb);
            uint32_t sf = (uint32_t)hmix(fkey) & (w.hcap - 1), sg = (uint32_t)hmix(gkey) & (w.hcap - 1);
            uint64_t kf = w.hkeys[sf], kg = w.hkeys[sg];  // both first probes in flight together
            int64_t fs = -1, gs = -1;
            for (uint32_t probe = 0; probe < w.hcap; ++probe) {
              if (kf == fkey) { fs = (int64_t)sf; break; }
              if (kf == HEMPTY) break;
              sf = (sf + 1) & (w.hcap - 1);
              kf = w.hkeys[sf];
            }
            if (fs >= 0) w.hkeys[fs] = HTOMB;
            // (gkey != fkey: a tombstone at fs and the key it replaces both mean "keep probing" to the lookup of gkey)
            for (uint32_t probe = 0; probe < w.hcap; ++probe) {
              if (kg == gkey) { gs = (int64_t)sg; break; }
              if (kg == HEMPTY) break;
              sg = (sg + 1) & (w.hcap - 1);
              kg = sg == (uint32_t)fs ? HTOMB : w.hkeys[sg];
            }
            bool move_f = true;  // f becomes {a, nb}
            if (gs >= 0) {
              const uint32_t g = w.hvals[gs];
              const unsigned long long sum_f = w.esum[f], sum_g = w.esum[g];
              const uint32_t cnt_f = w.ecnt[f], cnt_g = w.ecnt[g];
              const float st_f = w.escore[f], st_g = w.escore[g];
              const uint8_t gfl = fget(g);
              if (st_f > st_g) {  // the a-side edge is the cheaper one: it takes f's sums and keeps its place
                w.esum[g] = sum_g + sum_f;
                w.ecnt[g] = cnt_g + cnt_f;
                fset(g, gfl | 2);
                fset(f, ffl | 1);
                move_f = false;
              } else {            // f is the cheaper one (or ties): it takes g's sums and g's slot in the edge table
                w.esum[f] = sum_f + sum_g;
                w.ecnt[f] = cnt_f + cnt_g;
                fset(g, gfl | 1);
                w.hvals[gs] = f;
              }
            }
            if (move_f) {
              // b is replaced IN ITS SLOT so that nb's list keeps following the link that belongs to nb's slot;
              // f joins a's list through b's old slot
              const uint32_t ha = w.head[a];
              if (b_in_u) { w.eu[f] = a; w.enextu[f] = ha; } else { w.ev[f] = a; w.enextv[f] = ha; }
              w.head[a] = f;
              fset(f, ffl | 2);
              if (gs < 0) hput(gkey, f);
            }
          }
          f = nxt;
        }
        if (fail) break;
        {
          const int64_t es = hfind(norm_key(eu, evv));
          if (es >= 0) w.hkeys[es] = HTOMB;
        }
        fset(e, fget(e) | 1);
        w.parent[b] = a;
      }
      sh_dummy = items;
      if (fail) atomicOr(&w.counters[3], 16u);
    }
    __syncthreads();
    // snapshot of the roots at this threshold (parents always point to smaller ranks)
    for (uint32_t i = lane; i < nn; i += 64) {
      uint32_t r = i;
      for (;;) {
        const uint32_t p = __hip_atomic_load(&w.parent[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == r) break;
        r = p;
      }
      w.roots[(size_t)t * w.node_cap + i] = r;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void agg_merge_kernel(AggWs w, const AggThresholds thr_arg, int nthr) {
  __shared__ uint64_t hl[AGG_LDS_HEAP];
  __shared__ uint8_t fl_lds[AGG_LDS_HEAP];
  __shared__ int sh_dummy;
  static_assert(AGG_LDS_HEAP <= 65536, "FAST entries hold 16-bit ranks and edges");
  if (!xcd_claim(&w.counters[5], w.xcd_hint)) return;
  if (w.counters[3]) return;
  const uint32_t nn = w.counters[0];
  const uint32_t ne = min(w.counters[1], w.edge_cap);
  if (ne <= (uint32_t)AGG_LDS_HEAP && w.counters[7] != 0) agg_merge_body<true>(w, thr_arg, nthr, hl, fl_lds, &sh_dummy, nn, ne);
  else agg_merge_body<false>(w, thr_arg, nthr, hl, fl_lds, &sh_dummy, nn, ne);
}

// Last kernel of a call: an overflow of this call outlives the next call's reset of counters[3] (a pipeline that checks
// bsmi_seg_status once per lane after many blocks must still see it).
__device__ __forceinline__ void keep_overflow(const AggWs& w) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t f = w.counters[3];
    if (f) atomicOr(w.sticky, f);
  }
}

__global__ void agg_relabel_kernel(const uint64_t* __restrict__ frags, size_t n, int nthr, AggWs w,
                                   uint64_t* __restrict__ segs) {
  keep_overflow(w);
  if (w.counters[3]) return;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[p];
    if (!f) {
      for (int t = 0; t < nthr; ++t) segs[(size_t)t * n + p] = 0;
      continue;
    }
    const uint32_t r = w.rank_of_id[f];
    for (int t = 0; t < nthr; ++t) segs[(size_t)t * n + p] = w.ids[w.roots[(size_t)t * w.node_cap + r]];
  }
}

// ------------------------------------------------------------------------------------------
// per-block RAG edge scoring (reference post/blockwise/waterz_agglom.py:106-170); restated in
// oracle/seg_ref.c seg_rag_merge_scores_u8
// ------------------------------------------------------------------------------------------
__global__ void rag_ids_kernel(const uint64_t* __restrict__ frags, size_t n, int W, AggWs w) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[p];
    if (!f) continue;
    if (p % W != 0 && frags[p - 1] == f) continue;  // the run's first voxel inserts the id
    if (f >= HTOMB) { atomicOr(&w.counters[3], 1u); continue; }
    uint32_t s = (uint32_t)hmix(f) & (w.icap - 1);
    bool done = false;
    for (uint32_t probe = 0; probe < w.icap; ++probe) {
      const unsigned long long old = atomicCAS((unsigned long long*)&w.idkeys[s], HEMPTY, f);
      if (old == HEMPTY) {
        const uint32_t i = atomicAdd(&w.counters[0], 1u);
        if (i < w.node_cap) {
          w.idu[i] = f;
          w.idvals[s] = i;  // number in order of insertion (bsmi_rag_graph_u8 works with these; rag_rank_kernel replaces them by ranks)
        } else {
          atomicOr(&w.counters[3], 2u);
        }
        done = true;
        break;
      }
      if (old == f) { done = true; break; }
      s = (s + 1) & (w.icap - 1);
    }
    if (!done) atomicOr(&w.counters[3], 2u);
  }
}

__global__ void rag_pad_kernel(AggWs w) {
  const uint32_t nn = min(w.counters[0], w.node_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w.node_cap; i += gridDim.x * blockDim.x)
    if (i >= nn) w.idu[i] = HEMPTY;
}

__global__ void rag_rank_kernel(AggWs w) {
  if (w.counters[3]) return;
  const uint32_t nn = w.counters[0];
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nn; r += gridDim.x * blockDim.x) {
    const uint64_t f = w.ids[r];
    uint32_t s = (uint32_t)hmix(f) & (w.icap - 1);
    while (w.idkeys[s] != f) s = (s + 1) & (w.icap - 1);
    w.idvals[s] = r;
    w.head[r] = NOEDGE;
    w.parent[r] = r;
    w.cur[r] = r;
    w.tnext[r] = NOEDGE;
    w.ntime[r] = 0;
  }
}

__global__ void rag_iota_kernel(AggWs w) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w.hcap; i += gridDim.x * blockDim.x) w.iota[i] = i;
}

// edges numbered in ascending (u, v) order: edge e = position of its key in the sorted table
__global__ void rag_compact_kernel(AggWs w) {
  if (w.counters[3]) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w.hcap; i += gridDim.x * blockDim.x) {
    const uint64_t key = w.skeys[i];
    if (key == HEMPTY) continue;
    if (i >= w.edge_cap) { atomicOr(&w.counters[3], 8u); continue; }
    const uint32_t e = i, slot = w.sslot[i];
    const uint32_t u = (uint32_t)(key >> 32), v = (uint32_t)key;
    w.eu[e] = u; w.ev[e] = v; w.ekey0[e] = key;
    w.esum[e] = w.hsum[slot]; w.ecnt[e] = w.hcnt[slot];
    w.eflags[e] = 0;
    w.hvals[slot] = e;
    w.enextu[e] = atomicExch(&w.head[u], e);
    w.enextv[e] = atomicExch(&w.head[v], e);
    atomicMax(&w.counters[1], i + 1);
  }
}

__device__ __forceinline__ int64_t agg_hfind(const AggWs& w, uint64_t key) {
  uint32_t s = (uint32_t)hmix(key) & (w.hcap - 1);
  for (uint32_t probe = 0; probe < w.hcap; ++probe) {
    const uint64_t k = w.hkeys[s];
    if (k == key) return (int64_t)s;
    if (k == HEMPTY) return -1;
    s = (s + 1) & (w.hcap - 1);
  }
  return -1;
}

__device__ __forceinline__ bool agg_hput(const AggWs& w, uint64_t key, uint32_t val) {
  uint32_t s = (uint32_t)hmix(key) & (w.hcap - 1);
  uint32_t probe = 0;
  for (; probe < w.hcap; ++probe) {
    const uint64_t k = w.hkeys[s];
    if (k == HEMPTY || k == HTOMB || k == key) break;
    s = (s + 1) & (w.hcap - 1);
  }
  if (probe == w.hcap) return false;
  w.hkeys[s] = key;
  w.hvals[s] = val;
  return true;
}

__device__ __forceinline__ uint64_t agg_norm_key(uint32_t x, uint32_t y) {
  return x < y ? (((uint64_t)x << 32) | y) : (((uint64_t)y << 32) | x);
}

// ---- RAG path: bin-queue merge loop ---------------------------------------------------------------------------------
// Where the loop's per-edge state lives.  FAST (ne <= kRagFastEdges, nn <= kRagFastNodes: every 160^3 read box so far):
// endpoints, queue links, merge clocks, flags (16-bit / 8-bit), affinity sums, counts and stored scores (32-bit) as LDS
// arrays, 138 KB, so that a pop -- and most pops only find a stale edge to re-score: every merge makes all edges at the
// survivor stale, ten or more re-scorings per merge -- touches no HBM at all.  Otherwise the arrays of the workspace.
constexpr int kRagFastEdges = 6144, kRagFastNodes = 6144;
template <bool FAST>
struct RagMem {
  const AggWs& w;
  uint16_t *eu16, *ev16, *qn16, *et16, *nt16;
  uint8_t* fl8;
  uint32_t *sum32, *cnt32;  // a sum of affinity bytes over at most 3 * 160^3 voxel faces fits 32 bits
  float* sc32;
  __device__ __forceinline__ unsigned long long sum(uint32_t e) const { if constexpr (FAST) return sum32[e]; else return w.esum[e]; }
  __device__ __forceinline__ uint32_t cnt(uint32_t e) const { if constexpr (FAST) return cnt32[e]; else return w.ecnt[e]; }
  __device__ __forceinline__ void fold(uint32_t into, uint32_t from) const {   // the sums of `from` join those of `into`
    if constexpr (FAST) { sum32[into] += sum32[from]; cnt32[into] += cnt32[from]; }
    else { w.esum[into] += w.esum[from]; w.ecnt[into] += w.ecnt[from]; }
  }
  __device__ __forceinline__ float score(uint32_t e) const { if constexpr (FAST) return sc32[e]; else return w.escore[e]; }
  __device__ __forceinline__ void set_score(uint32_t e, float v) const { if constexpr (FAST) sc32[e] = v; else w.escore[e] = v; }
  __device__ __forceinline__ uint32_t eu(uint32_t e) const { if constexpr (FAST) return eu16[e]; else return w.eu[e]; }
  __device__ __forceinline__ uint32_t ev(uint32_t e) const { if constexpr (FAST) return ev16[e]; else return w.ev[e]; }
  __device__ __forceinline__ void set_eu(uint32_t e, uint32_t v) const { if constexpr (FAST) eu16[e] = (uint16_t)v; else w.eu[e] = v; }
  __device__ __forceinline__ void set_ev(uint32_t e, uint32_t v) const { if constexpr (FAST) ev16[e] = (uint16_t)v; else w.ev[e] = v; }
  __device__ __forceinline__ bool dead(uint32_t e) const { if constexpr (FAST) return fl8[e] & 1; else return w.eflags[e] & 1; }
  __device__ __forceinline__ void kill(uint32_t e) const { if constexpr (FAST) fl8[e] |= 1; else w.eflags[e] |= 1; }
  __device__ __forceinline__ uint32_t qnext(uint32_t e) const {
    if constexpr (FAST) { const uint16_t v = qn16[e]; return v == 0xffffu ? NOEDGE : (uint32_t)v; }
    else return w.qnext[e];
  }
  __device__ __forceinline__ void set_qnext(uint32_t e, uint32_t v) const { if constexpr (FAST) qn16[e] = (uint16_t)v; else w.qnext[e] = v; }
  __device__ __forceinline__ uint32_t etime(uint32_t e) const { if constexpr (FAST) return et16[e]; else return w.etime[e]; }
  __device__ __forceinline__ void set_etime(uint32_t e, uint32_t v) const { if constexpr (FAST) et16[e] = (uint16_t)v; else w.etime[e] = v; }
  __device__ __forceinline__ uint32_t ntime(uint32_t n) const { if constexpr (FAST) return nt16[n]; else return w.ntime[n]; }
  __device__ __forceinline__ void set_ntime(uint32_t n, uint32_t v) const { if constexpr (FAST) nt16[n] = (uint16_t)v; else w.ntime[n] = v; }
};

// contract live edge e: b = larger endpoint is absorbed by a (same rewiring as agg_merge_kernel).  Staleness in this path
// is a clock comparison (rag_merge_body), so nothing is flagged here: every edge that ends up incident to a is stale
// because a's clock moves.  Sequential form (one lane).
template <bool FAST>
__device__ __forceinline__ bool agg_contract(const RagMem<FAST>& M, uint32_t e) {
  const AggWs& w = M.w;
  const uint32_t eu = M.eu(e), evv = M.ev(e);
  const uint32_t a = eu < evv ? eu : evv, b = eu < evv ? evv : eu;
  uint32_t f = w.head[b];
  while (f != NOEDGE) {
    const uint32_t fu = M.eu(f), fv = M.ev(f);
    const bool b_in_u = fu == b;
    const uint32_t nxt = b_in_u ? w.enextu[f] : w.enextv[f];
    if (f != e && !M.dead(f)) {
      const uint32_t nb = b_in_u ? fv : fu;
      const uint64_t gkey = agg_norm_key(a, nb);
      const int64_t fs = agg_hfind(w, agg_norm_key(fu, fv));
      if (fs >= 0) w.hkeys[fs] = HTOMB;
      const int64_t gs = agg_hfind(w, gkey);
      bool move_f = true;
      if (gs >= 0) {
        const uint32_t g = w.hvals[gs];
        if (M.score(f) > M.score(g)) {
          M.fold(g, f);
          M.kill(f);
          move_f = false;
        } else {
          M.fold(f, g);
          M.kill(g);
          w.hvals[gs] = f;
        }
      }
      if (move_f) {
        if (b_in_u) { M.set_eu(f, a); w.enextu[f] = w.head[a]; } else { M.set_ev(f, a); w.enextv[f] = w.head[a]; }
        w.head[a] = f;
        if (gs < 0 && !agg_hput(w, gkey, f)) return false;
      }
    }
    f = nxt;
  }
  const int64_t es = agg_hfind(w, agg_norm_key(eu, evv));
  if (es >= 0) w.hkeys[es] = HTOMB;
  M.kill(e);
  w.parent[b] = a;
  return true;
}

constexpr int kMaxQueueBins = 1024;
constexpr int kSweepCap = 2048;  // incident edges the cooperative sweep takes per contraction (more: lane 0 alone, agg_contract)

// agg_contract by the whole wave (all 64 lanes call it with the same e, after a __syncthreads()).  The edges at b are
// independent of one another -- each meets a different neighbour -- so lane 0 only walks b's list into LDS (one memory
// round trip per edge) and the lanes then take one incident edge each: the two hash look-ups, the comparison of the stored
// scores, the fold or the move.  What must keep the order of the sequential loop does: the moved edges are linked into a's
// list in traversal order.  New hash entries go in by compare-and-swap (the slot an entry lands in may differ from the
// sequential run's; look-ups do not care).  -> false: hash table full.
template <bool FAST>
__device__ __forceinline__ bool agg_contract_wave(const RagMem<FAST>& M, uint32_t e, uint32_t* lst, int* sh, uint32_t& a_out, uint32_t& b_out) {
  const AggWs& w = M.w;
  const int lane = threadIdx.x;
  const uint32_t eu = M.eu(e), evv = M.ev(e);
  const uint32_t a = eu < evv ? eu : evv, b = eu < evv ? evv : eu;
  a_out = a;
  b_out = b;
  if (lane == 0) {
    int n = 0;
    uint32_t f = w.head[b];
    while (f != NOEDGE && n < kSweepCap) {
      const uint32_t fu = M.eu(f), nu = w.enextu[f], nv = w.enextv[f];  // independent loads: one round trip per edge
      lst[n++] = f;
      f = fu == b ? nu : nv;
    }
    sh[0] = f == NOEDGE ? n : -1;
  }
  __syncthreads();
  const int n = sh[0];
  if (n < 0) {  // a hub with more edges than the list holds: the sequential form
    if (lane == 0) sh[1] = agg_contract<FAST>(M, e) ? 1 : 0;
    __syncthreads();
    return sh[1] != 0;
  }
  uint32_t head_a = w.head[a];
  bool fail = false;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    uint32_t f = NOEDGE;
    bool live = false, b_in_u = false, move_f = false, put_f = false;
    uint64_t gkey = 0;
    if (i < n) {
      f = lst[i];
      live = f != e && !M.dead(f);
    }
    if (live) {
      const uint32_t fu = M.eu(f), fv = M.ev(f);
      b_in_u = fu == b;
      const uint32_t nb = b_in_u ? fv : fu;
      gkey = agg_norm_key(a, nb);
      const int64_t fs = agg_hfind(w, agg_norm_key(fu, fv));
      const int64_t gs = agg_hfind(w, gkey);
      if (fs >= 0) w.hkeys[fs] = HTOMB;
      move_f = true;
      if (gs >= 0) {
        const uint32_t g = w.hvals[gs];
        if (M.score(f) > M.score(g)) {
          M.fold(g, f);
          M.kill(f);
          move_f = false;
        } else {
          M.fold(f, g);
          M.kill(g);
          w.hvals[gs] = f;
        }
      } else {
        put_f = true;
      }
    }
    // link the moved edges into a's list in traversal order: each one in front of the previous mover
    const unsigned long long movers = __ballot(move_f);
    const unsigned long long lower = movers & ((1ull << lane) - 1ull);
    const int prev_lane = lower ? 63 - __builtin_clzll(lower) : lane;
    const uint32_t prev_f = __shfl(f, prev_lane);  // every lane takes part
    if (move_f) {
      const uint32_t link = lower ? prev_f : head_a;
      if (b_in_u) { M.set_eu(f, a); w.enextu[f] = link; } else { M.set_ev(f, a); w.enextv[f] = link; }
    }
    if (movers) head_a = __shfl(f, 63 - __builtin_clzll(movers));
    // new entries (a, nb) -> f
    if (put_f) {
      uint32_t s = (uint32_t)hmix(gkey) & (w.hcap - 1);
      bool placed = false;
      for (uint32_t probe = 0; probe < 2 * w.hcap && !placed; ++probe) {
        const unsigned long long k = ((volatile unsigned long long*)w.hkeys)[s];
        if (k == HEMPTY || k == HTOMB) {
          if (atomicCAS((unsigned long long*)&w.hkeys[s], k, (unsigned long long)gkey) == k) {
            w.hvals[s] = f;
            placed = true;
          }
          continue;  // lost the slot to another lane this instant: look at it again (it holds that lane's key now)
        }
        s = (s + 1) & (w.hcap - 1);
      }
      if (!placed) fail = true;
    }
    __syncthreads();  // the next chunk's look-ups see this chunk's table
  }
  if (lane == 0) {
    w.head[a] = head_a;
    const int64_t es = agg_hfind(w, agg_norm_key(eu, evv));
    if (es >= 0) w.hkeys[es] = HTOMB;
    M.kill(e);
    w.parent[b] = a;
  }
  const bool any_fail = __any(fail);
  __syncthreads();  // the sweep's stores are in place before lane 0 goes on
  return !any_fail;
}

// One wave replays the sequential bin-queue merge loop and grows the merge tree: lane 0 owns the queue and the
// decisions, the contraction of a popped edge is shared by the lanes (agg_contract_wave).
template <bool FAST>
__device__ __forceinline__ void rag_merge_body(const RagMem<FAST>& M, float threshold, int nbins, uint32_t nn, uint32_t ne, uint32_t* bhead,
                                               uint32_t* btail, uint32_t* lst, int* sh) {
  const AggWs& w = M.w;
  const int lane = threadIdx.x;
  int minbin = nbins;            // lane 0's
  const float scale = (float)(nbins - 1);
  auto push = [&](uint32_t e, float sc) {
    int b = (int)(sc * scale);
    b = b < 0 ? 0 : (b > nbins - 1 ? nbins - 1 : b);
    M.set_qnext(e, NOEDGE);
    if (bhead[b] == NOEDGE) bhead[b] = e; else M.set_qnext(btail[b], e);
    btail[b] = e;
    if (b < minbin) minbin = b;
  };
  // initial state and scores by all lanes, the pushes (edge order) by lane 0
  for (uint32_t e = lane; e < ne; e += 64) {
    if constexpr (FAST) {
      M.set_eu(e, w.eu[e]);
      M.set_ev(e, w.ev[e]);
      M.fl8[e] = 0;  // rag_compact_kernel left every flag at 0
      M.sum32[e] = (uint32_t)w.esum[e];
      M.cnt32[e] = w.ecnt[e];
    }
    M.set_score(e, agg_score(M.sum(e), M.cnt(e)));
    M.set_etime(e, 0);
  }
  if constexpr (FAST)
    for (uint32_t n = lane; n < nn; n += 64) M.set_ntime(n, 0);  // as rag_rank_kernel left the workspace's
  __syncthreads();
  if (lane == 0)
    for (uint32_t e = 0; e < ne; ++e) {
      const float sc = M.score(e);
      if (sc < threshold) push(e, sc);
    }
  // mergeRegions marks every edge incident to the survivor stale -- its own, the moved and the merged ones alike.
  // That is a clock: a merge stamps its survivor (ntime), scoring stamps the edge (etime), and an edge is stale when
  // one of its endpoints was stamped after it.
  uint32_t nm = 0, clock = 0;    // lane 0's
  for (;;) {
    // lane 0: pop until an edge is due for a merge (sh[2] = the edge, NOEDGE: queue empty)
    if (lane == 0) {
      uint32_t pick = NOEDGE;
      for (;;) {
        while (minbin < nbins && bhead[minbin] == NOEDGE) ++minbin;
        if (minbin >= nbins) break;
        const uint32_t e = bhead[minbin];
        bhead[minbin] = M.qnext(e);
        if (M.dead(e)) continue;
        const uint32_t tu = M.ntime(M.eu(e)), tv = M.ntime(M.ev(e));
        if (M.etime(e) < (tu > tv ? tu : tv)) {
          const float sc = agg_score(M.sum(e), M.cnt(e));
          M.set_score(e, sc);
          M.set_etime(e, clock);
          if (sc < threshold) push(e, sc);
          continue;
        }
        pick = e;
        break;
      }
      sh[2] = (int)pick;
    }
    __syncthreads();
    const uint32_t e = (uint32_t)sh[2];
    if (e == NOEDGE) break;
    uint32_t a, b;
    if (!agg_contract_wave<FAST>(M, e, lst, sh, a, b)) {
      if (lane == 0) atomicOr(&w.counters[3], 16u);
      break;
    }
    if (lane == 0) {
      const float sc = M.score(e);
      M.set_ntime(a, ++clock);
      const uint32_t t = nn + nm;
      w.tnext[w.cur[a]] = t;
      w.tnext[w.cur[b]] = t;
      w.cur[a] = t;
      w.tnext[t] = NOEDGE;
      w.tscore[t] = sc;
      w.ha[nm] = a;
      w.hb[nm] = b;
      ++nm;
    }
  }
  if (lane == 0) w.counters[4] = nm;
}

__global__ __launch_bounds__(64) void rag_merge_kernel(AggWs w, float threshold, int nbins, int allow_fast) {
  __shared__ uint32_t bhead[kMaxQueueBins], btail[kMaxQueueBins];
  __shared__ uint32_t lst[kSweepCap];
  __shared__ int sh[4];
  __shared__ uint16_t l_eu[kRagFastEdges], l_ev[kRagFastEdges], l_qn[kRagFastEdges], l_et[kRagFastEdges], l_nt[kRagFastNodes];
  __shared__ uint8_t l_fl[kRagFastEdges];
  __shared__ uint32_t l_sum[kRagFastEdges], l_cnt[kRagFastEdges];
  __shared__ float l_sc[kRagFastEdges];
  if (!xcd_claim(&w.counters[5], w.xcd_hint)) return;
  if (w.counters[3]) return;
  for (int b = threadIdx.x; b < nbins; b += 64) bhead[b] = btail[b] = NOEDGE;
  __syncthreads();
  const uint32_t nn = w.counters[0];
  const uint32_t ne = min(w.counters[1], w.edge_cap);
  if (allow_fast && ne <= (uint32_t)kRagFastEdges && nn <= (uint32_t)kRagFastNodes) {
    const RagMem<true> M{w, l_eu, l_ev, l_qn, l_et, l_nt, l_fl, l_sum, l_cnt, l_sc};
    rag_merge_body<true>(M, threshold, nbins, nn, ne, bhead, btail, lst, sh);
  } else {
    const RagMem<false> M{w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    rag_merge_body<false>(M, threshold, nbins, nn, ne, bhead, btail, lst, sh);
  }
}

// score of RAG edge {u, v} = score of the lowest common ancestor in the merge tree (tree nodes
// are numbered in creation order, so the smaller index is always the one to climb)
__global__ void rag_scores_kernel(AggWs w, uint64_t* __restrict__ edges, float* __restrict__ scores, uint64_t cap,
                                  uint64_t* __restrict__ merges, float* __restrict__ mscores, uint64_t* __restrict__ counts) {
  keep_overflow(w);
  if (w.counters[3]) return;
  const uint32_t nn = w.counters[0];
  const uint32_t ne = min(w.counters[1], w.edge_cap);
  const uint32_t nm = w.counters[4];
  if (ne > cap) {  // the caller's edge buffer is too small: say so, and how many entries the block needs (counts[0] > capacity)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      atomicOr(&w.counters[3], 32u);
      atomicOr(w.sticky, 32u);
      counts[0] = ne; counts[1] = nm; counts[2] = nn;
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { counts[0] = ne; counts[1] = nm; counts[2] = nn; }
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
    const uint64_t key = w.ekey0[e];
    uint32_t x = (uint32_t)(key >> 32), y = (uint32_t)key;
    edges[2 * (size_t)e] = w.ids[x];
    edges[2 * (size_t)e + 1] = w.ids[y];
    float sc = __uint_as_float(0x7fc00000u);
    for (;;) {
      if (x == y) { sc = w.tscore[x]; break; }
      if (x < y) { const uint32_t nx = w.tnext[x]; if (nx == NOEDGE) break; x = nx; }
      else { const uint32_t ny = w.tnext[y]; if (ny == NOEDGE) break; y = ny; }
    }
    scores[e] = sc;
  }
  if (merges)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
      merges[2 * (size_t)i] = w.ids[w.ha[i]];
      merges[2 * (size_t)i + 1] = w.ids[w.hb[i]];
      if (mscores) mscores[i] = w.tscore[nn + i];
    }
}

// The region graph straight out of the edge hash table, for bsmi_rag_graph_u8: nodes carry the numbers rag_ids_kernel gave them
// (insertion order), edges leave in table order; bsmi_rag_merge_scores_host brings them into ascending (id, id) order.  No sort
// on the device: the two radix sorts of the ordered form are 46 of its 60 launches, and a block's task is launch-bound.
__global__ void rag_graph_hash_out_kernel(AggWs w, uint64_t* __restrict__ edges, uint64_t* __restrict__ sums, uint32_t* __restrict__ cnts,
                                          uint64_t cap, uint64_t* __restrict__ counts) {
  keep_overflow(w);
  if (w.counters[3]) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[2] = w.counters[0];
  for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < w.hcap; sl += gridDim.x * blockDim.x) {
    const uint64_t key = w.hkeys[sl];
    if (key == HEMPTY) continue;
    const unsigned long long e = atomicAdd((unsigned long long*)&counts[0], 1ull);  // (zeroed with the call's tables)
    if (e >= cap) {  // the caller's buffers are too small: say so; counts[0] ends at the number of entries the block needs
      atomicOr(w.sticky, 32u);
      continue;
    }
    const uint64_t a = w.idu[(uint32_t)(key >> 32)], b = w.idu[(uint32_t)key];
    edges[2 * (size_t)e] = a < b ? a : b;
    edges[2 * (size_t)e + 1] = a < b ? b : a;
    sums[e] = w.hsum[sl];
    cnts[e] = w.hcnt[sl];
  }
}

// fragments -> ids of their merged clusters after rag_merge_kernel, in place (a cluster is named by its smallest id: the
// survivor of every merge is the smaller rank, and ranks ascend with the ids)
__global__ void rag_relabel_kernel(uint64_t* __restrict__ frags, size_t n, AggWs w) {
  keep_overflow(w);
  if (w.counters[3]) return;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[p];
    if (!f) continue;
    uint32_t r = agg_rank<true>(w, f);
    for (;;) {
      const uint32_t q = w.parent[r];
      if (q == r) break;
      r = q;
    }
    frags[p] = w.ids[r];
  }
}

// affinity sum and voxel-pair count of every INITIAL edge of the last RAG call, in edge order (the merge loop changes
// esum / ecnt; the hash table rows the edges were compacted from still hold the initial values)
__global__ void rag_edge_stats_kernel(AggWs w, uint64_t* __restrict__ sums, uint64_t* __restrict__ counts, uint64_t cap) {
  if (w.counters[3]) return;
  const uint32_t ne = min(w.counters[1], w.edge_cap);
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne && e < cap; e += gridDim.x * blockDim.x) {
    const uint32_t slot = w.sslot[e];
    sums[e] = w.hsum[slot];
    counts[e] = w.hcnt[slot];
  }
}

// segmentation lookup: out[p] = vals[k] where keys[k] == in[p] (keys ascending), 0 stays 0, an id
// that is not a key maps to itself.  volara Relabel + LUT (post/watershed.py:187-202).
__global__ void lut_relabel_kernel(const uint64_t* __restrict__ in, size_t n, const uint64_t* __restrict__ keys,
                                   const uint64_t* __restrict__ vals, uint64_t m, uint64_t* __restrict__ out) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = in[p];
    uint64_t r = f;
    if (f && m) {
      uint64_t lo = 0, hi = m;
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < f) lo = mid + 1; else hi = mid;
      }
      if (lo < m && keys[lo] == f) r = vals[lo];
    }
    out[p] = r;
  }
}

// The same for T value columns at once (one segmentation per threshold out of one fragment volume): a thread takes RUN
// consecutive voxels and searches only when the id changes -- fragments are compact, the next voxel along x mostly carries
// the same id --, and the one look-up serves all T outputs (three single passes over the slab: 17 dependent L2 reads per
// voxel and pass, 0.75 ms per pass and 20 blocks; this: one pass).
constexpr int kLutMaxColumns = 8;
struct LutColumns {
  const uint64_t* vals[kLutMaxColumns];
  uint64_t* out[kLutMaxColumns];
};
__global__ void lut_relabel_multi_kernel(const uint64_t* __restrict__ in, size_t n, const uint64_t* __restrict__ keys, uint64_t m, int T,
                                         LutColumns c) {
  constexpr int RUN = 8;
  const size_t nruns = (n + RUN - 1) / RUN;
  for (size_t r0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r0 < nruns; r0 += (size_t)gridDim.x * blockDim.x) {
    uint64_t last = 0, idx = m;  // idx == m: no key
    const size_t p0 = r0 * RUN, p1 = p0 + RUN < n ? p0 + RUN : n;
    for (size_t p = p0; p < p1; ++p) {
      const uint64_t f = in[p];
      if (f != last) {
        last = f;
        idx = m;
        if (f && m) {
          uint64_t lo = 0, hi = m;
          while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (keys[mid] < f) lo = mid + 1; else hi = mid;
          }
          if (lo < m && keys[lo] == f) idx = lo;
        }
      }
      for (int t = 0; t < T; ++t) c.out[t][p] = idx < m ? c.vals[t][idx] : f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// blockwise fragment post-processing (reference post/blockwise/watershed_frags.py:148-156,181-224)
// ------------------------------------------------------------------------------------------
struct FragWs {
  unsigned long long* lsum;  // [id_cap] sum of a_z + a_y + a_x over the voxels of a fragment
  uint32_t* lcnt;            // [id_cap] voxel count
  uint32_t id_cap;
  int32_t* par;              // [max_vox] union-find parent (crop volume)
  int32_t* rank;             // [max_vox] raster-order rank of a root
  uint32_t* blk;             // [max_vox / 1024 + 2] per-block root counts / offsets
  uint32_t* flags;           // [0] overflow (id >= id_cap)
};

__global__ void frag_stats_kernel(const uint8_t* __restrict__ affs, const uint64_t* __restrict__ frags, size_t n, FragWs w) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[p];
    if (!f) continue;
    if (f >= w.id_cap) { atomicOr(w.flags, 1u); continue; }
    atomicAdd(&w.lsum[f], (unsigned long long)((uint32_t)affs[p] + affs[n + p] + affs[2 * n + p]));
    atomicAdd(&w.lcnt[f], 1u);
  }
}

// filter_avg_fragments: mean of the 3-channel average affinity (u8 / 255) below filter_value;
// remove_small_objects: fewer than min_size voxels.  Both decided per fragment on the read ROI
// and recorded in bit 31 of the count.  The reference accumulates float64 values
// ((a0/255 + a1/255) + a2/255) / 3 in raster order (numpy mean over axis 0, scipy.ndimage.mean =
// bincount); the exact rational S / (765 n) decides unless it lies within 1e-9 of the filter,
// where the float64 accumulation is replayed sequentially so that the outcome is the reference's.
__global__ void frag_decide_kernel(const uint8_t* __restrict__ affs, const uint64_t* __restrict__ frags, size_t n, FragWs w,
                                   double filter_value, long long min_size) {
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < w.id_cap; f += gridDim.x * blockDim.x) {
    const uint32_t c = w.lcnt[f];
    if (!c || !f) continue;
    bool drop = false;
    if (filter_value > 0.0) {
      double mean = (double)w.lsum[f] / (765.0 * (double)c);
      if (fabs(mean - filter_value) <= 1e-9) {
        double sum = 0.0;
        for (size_t p = 0; p < n; ++p)
          if (frags[p] == f)
            sum += (((double)affs[p] / 255.0 + (double)affs[n + p] / 255.0) + (double)affs[2 * n + p] / 255.0) / 3.0;
        mean = sum / (double)c;
      }
      drop = mean < filter_value;
    }
    if (min_size > 0) drop = drop || (long long)c < min_size;
    if (drop) w.lcnt[f] = c | 0x80000000u;
  }
}

__global__ void frag_filter_kernel(uint64_t* __restrict__ frags, size_t n, FragWs w) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t f = frags[p];
    if (!f || f >= w.id_cap) continue;
    if (w.lcnt[f] & 0x80000000u) frags[p] = 0;
  }
}

__global__ void crop_u64_kernel(const uint64_t* __restrict__ in, int H, int W, int oz, int oy, int ox, int cd, int ch,
                                int cw, uint64_t* __restrict__ out) {
  const size_t n = (size_t)cd * ch * cw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw);
    const int y = (int)((i / cw) % ch);
    const int z = (int)(i / ((size_t)cw * ch));
    out[i] = in[((size_t)(z + oz) * H + (y + oy)) * W + (x + ox)];
  }
}

__device__ __forceinline__ int cc_find(int32_t* par, int a) {
  int p = __hip_atomic_load(&par[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != a) {
    a = p;
    p = __hip_atomic_load(&par[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return a;
}

__global__ void cc26_init_kernel(const uint64_t* __restrict__ x, size_t n, FragWs w) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x)
    w.par[p] = x[p] ? (int32_t)p : -1;
}

// unite every voxel with its 13 raster-preceding neighbours of equal value (26-connectivity)
__global__ void cc26_union_kernel(const uint64_t* __restrict__ x, int D, int H, int W, FragWs w) {
  const size_t n = (size_t)D * H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t v = x[p];
    if (!v) continue;
    const int xx = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int z = (int)(p / ((size_t)W * H));
    for (int dz = -1; dz <= 0; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
          const int zz = z + dz, yy = y + dy, x2 = xx + dx;
          if (zz < 0 || yy < 0 || yy >= H || x2 < 0 || x2 >= W) continue;
          const size_t q = ((size_t)zz * H + yy) * W + x2;
          if (x[q] != v) continue;
          int a = (int)p, b = (int)q;
          for (;;) {
            a = cc_find(w.par, a);
            b = cc_find(w.par, b);
            if (a == b) break;
            if (a < b) { const int t = a; a = b; b = t; }
            const int old = atomicMin(&w.par[a], b);
            if (old == a) break;
            a = old;
          }
        }
  }
}

// roots per 1024-voxel block (raster order), then one workgroup scans the block counts
__global__ __launch_bounds__(1024) void cc26_count_kernel(size_t n, FragWs w) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const size_t p = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (p < n && w.par[p] == (int32_t)p) atomicAdd(&cnt, 1u);
  __syncthreads();
  if (threadIdx.x == 0) w.blk[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(1024) void cc26_scan_kernel(uint32_t nblk, FragWs w, uint64_t* num_out) {
  __shared__ uint32_t sh[1024];
  const uint32_t chunk = (nblk + 1023) / 1024;
  const uint32_t c0 = threadIdx.x * chunk, c1 = min(nblk, c0 + chunk);
  uint32_t s = 0;
  for (uint32_t i = c0; i < c1; ++i) s += w.blk[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int t = 0; t < 1024; ++t) { const uint32_t c = sh[t]; sh[t] = acc; acc += c; }
    *num_out = acc;
  }
  __syncthreads();
  uint32_t acc = sh[threadIdx.x];
  for (uint32_t i = c0; i < c1; ++i) { const uint32_t c = w.blk[i]; w.blk[i] = acc; acc += c; }
}

// rank of every root = number of roots before it in raster order (+1)
__global__ __launch_bounds__(1024) void cc26_rank_kernel(size_t n, FragWs w) {
  __shared__ uint32_t sh[1024];
  const size_t p = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t flag = (p < n && w.par[p] == (int32_t)p) ? 1u : 0u;
  sh[threadIdx.x] = flag;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
    const uint32_t v = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  if (flag) w.rank[p] = (int32_t)(w.blk[blockIdx.x] + sh[threadIdx.x]);
}

__global__ void cc26_write_kernel(const uint64_t* __restrict__ x, size_t n, FragWs w, uint64_t id_offset, uint64_t* __restrict__ out) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    if (!x[p]) { out[p] = 0; continue; }
    out[p] = id_offset + (uint64_t)w.rank[cc_find(w.par, (int)p)];
  }
}

// thresholded-affinity connected components (reference post/cc.py:7-74): voxel p is linked with p + e_d when
// affs[d][p] > cut; a voxel is labelled if it has a link of its own (even one that leaves the volume) or is the far end
// of a neighbour's link.  Roots are the raster-first voxels of their components, so the cc26 ranking kernels give the
// reference's numbering (depth-first fills started in raster order).
__global__ void ccaff_init_kernel(size_t n, FragWs w, uint64_t* __restrict__ touched) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    w.par[p] = (int32_t)p;
    touched[p] = 0;
  }
}

__global__ void ccaff_union_kernel(const uint8_t* __restrict__ affs, int D, int H, int W, int cut, FragWs w, uint64_t* __restrict__ touched) {
  const size_t n = (size_t)D * H * W, hw = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / hw);
    const bool ok[3] = {z + 1 < D, y + 1 < H, x + 1 < W};
    const size_t st[3] = {hw, (size_t)W, 1};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if ((int)affs[(size_t)d * n + p] <= cut) continue;
      touched[p] = 1;
      if (!ok[d]) continue;
      const size_t q = p + st[d];
      touched[q] = 1;
      int a = (int)p, b = (int)q;
      for (;;) {
        a = cc_find(w.par, a);
        b = cc_find(w.par, b);
        if (a == b) break;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&w.par[a], b);
        if (old == a) break;
        a = old;
      }
    }
  }
}

__global__ void ccaff_finalize_kernel(size_t n, FragWs w, const uint64_t* __restrict__ touched) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x)
    if (!touched[p]) w.par[p] = -1;
}

// label table of a block (reference refine.py:98-109 `_global_sizes`, :228-250 z extents): every distinct non-zero id
// with its voxel count and the first / last z slice it occurs in.  Runs of equal ids along x are counted once.
__global__ void ltab_clear_kernel(AggWs w) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < w.icap; i += (size_t)gridDim.x * blockDim.x) {
    w.idkeys[i] = HEMPTY;
    w.tcount[i] = 0;
    w.tzmin[i] = 0x7fffffff;
    w.tzmax[i] = -0x7fffffff;
  }
}

__global__ void ltab_scan_kernel(const uint64_t* __restrict__ lab, int D, int H, int W, int z0, AggWs w) {
  const size_t nrows = (size_t)D * H;
  for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += (size_t)gridDim.x * blockDim.x) {
    const int z = (int)(row / H);
    const uint64_t* p = lab + row * W;
    int x = 0;
    while (x < W) {
      const uint64_t f = p[x];
      int len = 1;
      while (x + len < W && p[x + len] == f) ++len;
      x += len;
      if (!f) continue;
      if (f >= HTOMB) { atomicOr(&w.counters[3], 1u); continue; }
      uint32_t s = (uint32_t)hmix(f) & (w.icap - 1);
      bool ok = false;
      for (uint32_t probe = 0; probe < w.icap; ++probe) {
        const unsigned long long old = atomicCAS((unsigned long long*)&w.idkeys[s], HEMPTY, f);
        if (old == HEMPTY) {
          if (atomicAdd(&w.counters[0], 1u) >= w.node_cap) atomicOr(&w.counters[3], 2u);
          ok = true;
          break;
        }
        if (old == f) { ok = true; break; }
        s = (s + 1) & (w.icap - 1);
      }
      if (!ok) { atomicOr(&w.counters[3], 2u); continue; }
      atomicAdd(&w.tcount[s], (unsigned long long)len);
      atomicMin(&w.tzmin[s], z0 + z);
      atomicMax(&w.tzmax[s], z0 + z);
    }
  }
}

__global__ void ltab_compact_kernel(AggWs w) {
  if (w.counters[3]) return;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < w.icap; s += gridDim.x * blockDim.x) {
    if (w.idkeys[s] == HEMPTY) continue;
    const uint32_t i = atomicAdd(&w.counters[1], 1u);
    if (i < w.node_cap) {
      w.idu[i] = w.idkeys[s];
      w.ha[i] = s;
    }
  }
}

__global__ void ltab_pad_kernel(AggWs w) {
  const uint32_t nn = min(w.counters[0], w.node_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w.node_cap; i += gridDim.x * blockDim.x)
    if (i >= nn) { w.idu[i] = HEMPTY; w.ha[i] = 0; }
}

__global__ void ltab_gather_kernel(AggWs w, uint64_t* __restrict__ ids, uint64_t* __restrict__ counts, int32_t* __restrict__ zmin,
                                   int32_t* __restrict__ zmax, uint64_t cap, uint64_t* __restrict__ n_out) {
  keep_overflow(w);
  if (w.counters[3]) return;
  const uint32_t nn = w.counters[0];
  if (nn > cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicOr(&w.counters[3], 32u); atomicOr(w.sticky, 32u); }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = nn;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
    const uint32_t s = w.hb[i];
    ids[i] = w.ids[i];
    counts[i] = w.tcount[s];
    zmin[i] = w.tzmin[s];
    zmax[i] = w.tzmax[s];
  }
}

// ------------------------------------------------------------------------------------------
// fragments_in_xy = False (reference post/ws.py:97-110): one 3-D domain.  Same steps as the per-slice path --
// mask (a_z + a_y + a_x >= 383), exact squared EDT, separable reflect max filter, 6-connected maxima components in
// raster order, literal replay of skimage's heap -- but the flood is ONE sequential queue over the whole volume (the
// reference's own algorithm), so this mode is latency-bound on a single wave; the parallel steps are plain kernels.
// ------------------------------------------------------------------------------------------
constexpr int INF3 = 1 << 28;

__global__ void ws3_mask_kernel(const uint8_t* __restrict__ affs, size_t n, uint8_t* __restrict__ mask, int* __restrict__ any_bg) {
  int bg = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)affs[i] + (int)affs[n + i] + (int)affs[2 * n + i] >= 383;
    mask[i] = (uint8_t)m;
    bg |= !m;
  }
  if (bg) atomicOr(any_bg, 1);
}

// x pass: squared distance to the nearest background voxel of the same row (INF3 if none); one thread per row
__global__ void ws3_edt_x_kernel(const uint8_t* __restrict__ mask, int rows, int W, int32_t* __restrict__ g) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const uint8_t* m = mask + (size_t)r * W;
    int32_t* o = g + (size_t)r * W;
    int last = -INF3;
    for (int x = 0; x < W; ++x) {
      if (!m[x]) last = x;
      o[x] = last <= -INF3 ? INF3 : (x - last) * (x - last);
    }
    last = INF3;
    for (int x = W - 1; x >= 0; --x) {
      if (!m[x]) last = x;
      if (last < INF3) {
        const int d = (last - x) * (last - x);
        if (d < o[x]) o[x] = d;
      }
    }
  }
}

// out[p] = min over k along `axis` of in[p with coordinate k] + (coord - k)^2 (exact; brute force over the axis)
__global__ void ws3_edt_axis_kernel(const int32_t* __restrict__ in, int D, int H, int W, int axis, int32_t* __restrict__ out) {
  const size_t n = (size_t)D * H * W;
  const size_t stride = axis == 0 ? (size_t)H * W : (size_t)W;
  const int len = axis == 0 ? D : H;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int c = axis == 0 ? (int)(p / ((size_t)H * W)) : (int)((p / W) % H);
    const size_t base = p - (size_t)c * stride;
    int best = INF3;
    for (int k = 0; k < len; ++k) {
      const int v = in[base + (size_t)k * stride];
      if (v >= INF3) continue;
      const int d = v + (c - k) * (c - k);
      best = d < best ? d : best;
    }
    out[p] = best;
  }
}

// scipy's result when the volume has no background voxel at all: as if the only one sat at index (-1, 0, 0)
__global__ void ws3_edt_nobg_kernel(int D, int H, int W, const int* __restrict__ any_bg, int32_t* __restrict__ d2) {
  if (*any_bg) return;
  const size_t n = (size_t)D * H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / ((size_t)H * W));
    d2[p] = (z + 1) * (z + 1) + y * y + x * x;
  }
}

// maximum over the window [c - size/2, c + size - 1 - size/2] along `axis`, border mode reflect (edge duplicated)
__global__ void ws3_maxfilter_kernel(const int32_t* __restrict__ in, int D, int H, int W, int axis, int size, int32_t* __restrict__ out) {
  const size_t n = (size_t)D * H * W;
  const size_t stride = axis == 0 ? (size_t)H * W : (axis == 1 ? (size_t)W : 1);
  const int len = axis == 0 ? D : (axis == 1 ? H : W);
  const int left = size / 2, right = size - 1 - size / 2;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int c = axis == 0 ? (int)(p / ((size_t)H * W)) : (axis == 1 ? (int)((p / W) % H) : (int)(p % W));
    const size_t base = p - (size_t)c * stride;
    int m = INT32_MIN;
    for (int k = c - left; k <= c + right; ++k) {
      const int v = in[base + (size_t)reflect_dup(k, len) * stride];
      m = v > m ? v : m;
    }
    out[p] = m;
  }
}

// maxima flag (as the uint64 "value" array of the cc kernels) and union-find initialisation
__global__ void ws3_maxima_kernel(const int32_t* __restrict__ d2, const int32_t* __restrict__ mf, size_t n, uint64_t* __restrict__ flag, FragWs w) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const bool mx = d2[p] == mf[p];
    flag[p] = mx ? 1ull : 0ull;
    w.par[p] = mx ? (int32_t)p : -1;
  }
}

// 6-connected union of equal non-zero values with the three raster-preceding neighbours
__global__ void cc6_union_kernel(const uint64_t* __restrict__ x, int D, int H, int W, FragWs w) {
  const size_t n = (size_t)D * H * W, hw = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t v = x[p];
    if (!v) continue;
    const int xx = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / hw);
    const bool ok[3] = {z > 0, y > 0, xx > 0};
    const size_t st[3] = {hw, (size_t)W, 1};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!ok[d] || x[p - st[d]] != v) continue;
      int a = (int)p, b = (int)(p - st[d]);
      for (;;) {
        a = cc_find(w.par, a);
        b = cc_find(w.par, b);
        if (a == b) break;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&w.par[a], b);
        if (old == a) break;
        a = old;
      }
    }
  }
}

// markers: component rank of the maxima inside the mask, 0 elsewhere
__global__ void ws3_markers_kernel(const uint64_t* __restrict__ flag, const uint8_t* __restrict__ mask, size_t n, FragWs w, int32_t* __restrict__ lab,
                                   uint64_t* __restrict__ seeds) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int l = flag[p] ? (int)w.rank[cc_find(w.par, (int)p)] : 0;
    lab[p] = mask[p] ? l : 0;
    if (seeds) seeds[p] = (uint64_t)l;
  }
}

// return_seeds of the xy mode: slice-local seed labels + the slice's id offset (ws.py:24, 82-90)
__global__ void ws_seeds_out_kernel(int D, size_t n, WsScratch s, uint64_t* __restrict__ seeds) {
  const size_t total = (size_t)D * n;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const int l = s.seedlab[p];
    seeds[p] = l ? (uint64_t)l + s.offs[p / n] : 0ull;
  }
}

// heap entry of the 3-D flood: [63:46] = MAXD2 - d2 (18 bit) | [45:23] = age (23 bit) | [22:0] = voxel index
__device__ __forceinline__ bool flood3_smaller(uint64_t a, uint64_t b) { return (a >> 23) < (b >> 23); }

__global__ __launch_bounds__(64) void ws3_flood_kernel(int D, int H, int W, WsScratch s, uint64_t* __restrict__ hg, uint64_t* __restrict__ frags) {
  __shared__ uint64_t hl[8192];
  constexpr int LH = 8192;
  const size_t n = (size_t)D * H * W, hw = (size_t)H * W;
  const int lane = threadIdx.x;
  const uint8_t* mask = s.mask;
  const int32_t* d2 = s.d2;
  int32_t* lab = s.lab;
  // As in ws_flood_kernel the wave walks the one sequential loop in lockstep: lane 0 writes the LDS part of the heap, every
  // lane its HBM part (same address, same value: each lane's later loads then follow its own stores), lanes k < 6 fetch
  // neighbour k of the popped voxel in one round of loads issued before the sift-down, and write the labels.
  {
    constexpr uint64_t MAXD2 = (1u << 18) - 1;
    size_t items = 0;
    auto hget = [&](size_t i) -> uint64_t { return i < LH ? hl[i] : hg[i - LH]; };
    auto hset = [&](size_t i, uint64_t v) {
      if (i < LH) {
        if (lane == 0) hl[i] = v;
      } else {
        hg[i - LH] = v;
      }
    };
    auto push = [&](uint64_t it) {
      size_t c = items++;
      while (c > 0) {
        const size_t p = (c + 1) / 2 - 1;
        const uint64_t pv = hget(p);
        if (flood3_smaller(it, pv)) { hset(c, pv); c = p; } else break;
      }
      hset(c, it);
    };
    for (size_t i0 = 0; i0 < n; i0 += 64) {
      const size_t i = i0 + lane;
      const int li = i < n ? lab[i] : 0;
      unsigned long long seeds = __ballot(li != 0);
      while (seeds) {
        const int k = __ffsll(seeds) - 1;
        seeds &= seeds - 1;
        const size_t j = i0 + k;
        push(((MAXD2 - (uint64_t)d2[j]) << 46) | (uint64_t)j);
      }
    }
    uint64_t age = 0;
    const int k8 = lane & 7;
    const long long dq = k8 == 0 ? -(long long)hw : (k8 == 1 ? -(long long)W : (k8 == 2 ? -1 : (k8 == 3 ? 1 : (k8 == 4 ? (long long)W : (long long)hw))));
    while (items > 0) {
      const uint64_t e = hget(0);
      --items;
      const size_t idx = (size_t)(e & 0x7fffffu);
      const int x = (int)(idx % W), y = (int)((idx / W) % H), z = (int)(idx / hw);
      // neighbour order [-HW, -W, -1, +1, +W, +HW]
      const bool okk = k8 == 0 ? z > 0 : (k8 == 1 ? y > 0 : (k8 == 2 ? x > 0 : (k8 == 3 ? x < W - 1 : (k8 == 4 ? y < H - 1 : z < D - 1))));
      const size_t qk = okk ? (size_t)((long long)idx + dq) : idx;
      int lme = 0, mk = 0, lk = 0, dk = 0;
      if (lane < 6) {
        lme = lab[idx];
        mk = mask[qk];
        lk = lab[qk];
        dk = d2[qk];
      }
      const int l = __builtin_amdgcn_readfirstlane(lme);
      if (items > 0) {
        const uint64_t last = hget(items);
        size_t i = 0;
        for (;;) {
          const size_t c1 = 2 * i + 1, c2 = c1 + 1;
          if (c1 >= items) break;
          const uint64_t v1 = hget(c1);
          size_t sm = i;
          uint64_t smv = last;
          if (flood3_smaller(v1, smv)) { sm = c1; smv = v1; }
          if (c2 < items) {
            const uint64_t v2 = hget(c2);
            if (flood3_smaller(v2, smv)) { sm = c2; smv = v2; }
          }
          if (sm == i) break;
          hset(i, smv);
          i = sm;
        }
        hset(i, last);
      }
      const int cand = (lane < 6 && okk && mk && lk == 0) ? 1 : 0;
      const int qlo = (int)qk;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (!__shfl(cand, k)) continue;  // wave uniform
        const size_t q = (size_t)(uint32_t)__shfl(qlo, k);
        const uint64_t dd = (uint64_t)(uint32_t)__shfl(dk, k);
        ++age;
        if (lane < 6) lab[q] = l;
        push(((MAXD2 - dd) << 46) | (age << 23) | (uint64_t)q);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (size_t i = lane; i < n; i += 64) {
    const int l = lab[i];
    frags[i] = l ? (uint64_t)l : 0ull;
  }
}

// fragments of the 3-D mode = the flooded labels (bsmi_seg_set_host_flood: the flood ran on the host)
__global__ void ws3_labels_out_kernel(const int32_t* __restrict__ lab, size_t n, uint64_t* __restrict__ frags) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) frags[i] = (uint64_t)(uint32_t)lab[i];
}

// per-label voxel count and coordinate sums (RAG node attributes, watershed_frags.py:230-246)
__global__ void label_stats_kernel(const uint64_t* __restrict__ lab, int D, int H, int W, uint64_t id_offset, uint64_t num,
                                   unsigned long long* __restrict__ size, unsigned long long* __restrict__ sums) {
  const size_t n = (size_t)D * H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t l = lab[p];
    if (l <= id_offset || l - id_offset > num) continue;
    const uint64_t k = l - id_offset - 1;
    atomicAdd(&size[k], 1ull);
    atomicAdd(&sums[3 * k + 0], (unsigned long long)(p / ((size_t)W * H)));
    atomicAdd(&sums[3 * k + 1], (unsigned long long)((p / W) % H));
    atomicAdd(&sums[3 * k + 2], (unsigned long long)(p % W));
  }
}

// clears the per-call tables of the agglomeration (instead of four runtime fill kernels)
__global__ void seg_clear_kernel(AggWs w) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t0; i < w.id_cap; i += stride) w.rank_of_id[i] = 0;
  for (size_t i = t0; i < w.hcap; i += stride) {
    w.hkeys[i] = HEMPTY;
    w.hsum[i] = 0;
    w.hcnt[i] = 0;
  }
}

// Several buffers set to a 32-bit pattern by ONE launch.  The runtime's fill (hipMemsetAsync) is a launch per buffer and a narrow one:
// the 33 MB sum table of the fragment filter took 0.84 ms beside the lanes' floods, the three tables together 1.7 ms of a block's
// 22 ms chain, and every launch is host time of the one thread that queues all lanes (kernel trace of the driver's job).
constexpr int kMaxFills = 6;
struct FillList {
  uint32_t* p[kMaxFills];
  size_t words[kMaxFills];
  uint32_t value[kMaxFills];
  int n;
};
__global__ __launch_bounds__(256) void fill_list_kernel(FillList L) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < L.n; ++j) {
    uint32_t* p = L.p[j];
    const size_t w = L.words[j];
    const uint32_t v = L.value[j];
    size_t head = (size_t)((0 - (uintptr_t)p) >> 2) & 3;  // words up to 16-byte alignment
    head = head < w ? head : w;
    if (t0 < head) p[t0] = v;
    uint4* q = (uint4*)(p + head);
    const size_t nv = (w - head) >> 2;
    for (size_t i = t0; i < nv; i += stride) q[i] = make_uint4(v, v, v, v);
    const size_t done = head + 4 * nv;
    if (t0 < w - done) p[done + t0] = v;
  }
}
struct Fills {
  FillList L{};
  size_t most = 0;
  void add(void* p, size_t bytes, uint32_t value = 0) {  // 4-byte aligned, whole words (every table here is)
    L.p[L.n] = (uint32_t*)p; L.words[L.n] = bytes / 4; L.value[L.n] = value; ++L.n;
    most = std::max(most, bytes / 16);
  }
  void launch(hipStream_t s) const {
    const unsigned grid = (unsigned)std::min<size_t>((most + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(fill_list_kernel, dim3(grid), dim3(256), 0, s, L);
  }
};

int seg_scan_grid() {
  static const int g = [] { const char* e = getenv("BSMI_SEG_SCAN_GRID"); const int v = e ? atoi(e) : 32; return v < 1 ? 1 : v; }();
  return g;
}

}  // namespace bsmi

using namespace bsmi;

constexpr int kMaxThresholds = 16;

struct bsmi_seg {
  int device = 0;
  int64_t max_shape[3] = {0, 0, 0};
  size_t max_vox = 0;
  std::vector<void*> allocs;
  WsScratch ws{};
  uint64_t* flood_spill = nullptr;
  size_t flood_spill_stride = 0;
  AggWs agg{};
  FragWs frag{};
  uint64_t* crop_tmp = nullptr;  // [max_vox] cropped fragments before relabelling
  void* sort_tmp = nullptr;      // hipcub radix-sort scratch
  size_t sort_tmp_bytes = 0;
  int32_t* seedlab = nullptr;      // [max_vox] unmasked seed labels (bsmi_ws_fragments_seeds_u8), allocated on first use
  uint64_t* rag_counts = nullptr;  // [4] ne, nm, nn of the last RAG call
  float* thr_dev = nullptr;
  int* status_dev = nullptr;
  bool host_flood3 = false;  // bsmi_seg_set_host_flood: the 3-D watershed's flood runs on the host (flood_host.cpp)
};

namespace bsmi {
void host_flood3(int D, int H, int W, const uint8_t* mask, const int32_t* d2, int32_t* lab);  // flood_host.cpp
}

namespace bsmi {
template <typename T>
static int dalloc(bsmi_seg* h, T** p, size_t count) {
  void* q = nullptr;
  BSMI_HIP(hipMalloc(&q, count * sizeof(T) + 16));
  h->allocs.push_back(q);
  *p = (T*)q;
  return BSMI_OK;
}
static uint32_t next_pow2(uint64_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
}  // namespace bsmi

extern "C" {

int bsmi_seg_create(int device, const int64_t max_shape[3], bsmi_seg** out) {
  if (!max_shape || !out) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  for (int d = 0; d < 3; ++d)
    if (max_shape[d] < 1 || max_shape[d] > 4096) BSMI_FAIL(BSMI_ERR_INVALID, "bad max_shape");
  if (max_shape[1] * max_shape[2] >= (1 << 20))
    BSMI_FAIL(BSMI_ERR_INVALID, "slices larger than 2^20 voxels are not supported by the flood kernel");
  BSMI_HIP(hipSetDevice(device));
  bsmi_seg* h = new bsmi_seg;
  h->device = device;
  for (int d = 0; d < 3; ++d) h->max_shape[d] = max_shape[d];
  const size_t nv = (size_t)max_shape[0] * max_shape[1] * max_shape[2];
  const size_t ns = (size_t)max_shape[1] * max_shape[2];
  h->max_vox = nv;
  int rc = 0;
#define A(ptr, cnt) if (!rc) rc = dalloc(h, &(ptr), (cnt))
  A(h->ws.mask, nv); A(h->ws.g, nv); A(h->ws.d2, nv); A(h->ws.mf, nv); A(h->ws.par, nv); A(h->ws.lab, nv);
  A(h->ws.nseeds, (size_t)max_shape[0]); A(h->ws.offs, (size_t)max_shape[0]);
  h->ws.seedlab = nullptr;
  h->flood_spill_stride = ns;
  A(h->flood_spill, nv);
  AggWs& g = h->agg;
  static std::atomic<int> next_xcd{0};
  g.xcd_hint = next_xcd.fetch_add(1) & 7;
  g.id_cap = (uint32_t)std::min<size_t>(nv + 2, (size_t)1 << 27);
  g.node_cap = (uint32_t)std::min<size_t>(nv / 8 + 1024, (size_t)1 << 24);
  g.hcap = next_pow2(std::max<size_t>(nv / 2, 1024));
  g.edge_cap = g.hcap / 2;
  A(g.rank_of_id, (size_t)g.id_cap); A(g.ids, (size_t)g.node_cap); A(g.counters, 8);
  A(g.hkeys, (size_t)g.hcap); A(g.hvals, (size_t)g.hcap); A(g.hsum, (size_t)g.hcap); A(g.hcnt, (size_t)g.hcap);
  A(g.eu, (size_t)g.edge_cap); A(g.ev, (size_t)g.edge_cap); A(g.ekey0, (size_t)g.edge_cap);
  A(g.esum, (size_t)g.edge_cap); A(g.ecnt, (size_t)g.edge_cap); A(g.enextu, (size_t)g.edge_cap);
  A(g.enextv, (size_t)g.edge_cap); A(g.eflags, (size_t)g.edge_cap);
  A(g.head, (size_t)g.node_cap); A(g.parent, (size_t)g.node_cap);
  A(g.roots, (size_t)g.node_cap * kMaxThresholds); A(g.heap_spill, (size_t)g.edge_cap); A(g.maxid, 1);
  A(g.sticky, 1);
  A(g.escore, (size_t)g.edge_cap); A(g.etime, (size_t)g.edge_cap); A(g.ntime, (size_t)g.node_cap);
  A(h->thr_dev, kMaxThresholds); A(h->status_dev, 4);
  FragWs& f = h->frag;
  f.id_cap = (uint32_t)std::min<size_t>(nv + 2, (size_t)1 << 27);
  A(f.lsum, (size_t)f.id_cap); A(f.lcnt, (size_t)f.id_cap); A(f.rank, nv); A(f.blk, nv / 1024 + 2); A(f.flags, 4);
  f.par = h->ws.par;
  A(h->crop_tmp, nv);
  g.icap = next_pow2((size_t)g.node_cap * 2);
  A(g.idkeys, (size_t)g.icap); A(g.idvals, (size_t)g.icap); A(g.idu, (size_t)g.node_cap);
  A(g.skeys, (size_t)g.hcap); A(g.sslot, (size_t)g.hcap); A(g.iota, (size_t)g.hcap); A(g.qnext, (size_t)g.edge_cap);
  A(g.tnext, (size_t)g.node_cap * 2); A(g.tscore, (size_t)g.node_cap * 2); A(g.cur, (size_t)g.node_cap);
  A(g.ha, (size_t)g.node_cap); A(g.hb, (size_t)g.node_cap); A(h->rag_counts, 4);
  A(g.tcount, (size_t)g.icap); A(g.tzmin, (size_t)g.icap); A(g.tzmax, (size_t)g.icap);
  if (!rc) {
    size_t b1 = 0, b2 = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, b1, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)g.node_cap) != hipSuccess ||
        hipcub::DeviceRadixSort::SortPairs(nullptr, b2, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)g.hcap) != hipSuccess) {
      bsmi::set_error("hipcub temp-storage query failed");
      rc = BSMI_ERR_HIP;
    } else {
      h->sort_tmp_bytes = std::max(b1, b2) + 256;
      uint8_t* t = nullptr;
      rc = dalloc(h, &t, h->sort_tmp_bytes);
      h->sort_tmp = t;
    }
  }
#undef A
  if (rc) {
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return rc;
  }
  BSMI_HIP(hipMemset(g.counters, 0, 8 * sizeof(uint32_t)));
  BSMI_HIP(hipMemset(g.sticky, 0, sizeof(uint32_t)));
  BSMI_HIP(hipMemset(h->frag.flags, 0, 4 * sizeof(uint32_t)));
  BSMI_HIP(hipDeviceSynchronize());  // the fills ran on the null stream; the lanes' streams are non-blocking and would not wait for them
  *out = h;
  return BSMI_OK;
}

int bsmi_seg_destroy(bsmi_seg* h) {
  if (!h) return BSMI_OK;
  (void)hipSetDevice(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return BSMI_OK;
}

static int check_seg_shape(bsmi_seg* h, const int64_t shape[3]) {
  if (!h || !shape) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  for (int d = 0; d < 3; ++d)
    if (shape[d] < 1) BSMI_FAIL(BSMI_ERR_INVALID, "bad shape");
  if ((size_t)shape[0] * shape[1] * shape[2] > h->max_vox || shape[0] > h->max_shape[0] ||
      shape[1] * shape[2] > h->max_shape[1] * h->max_shape[2])
    BSMI_FAIL(BSMI_ERR_INVALID, "shape (%lld,%lld,%lld) exceeds the handle's max_shape (%lld,%lld,%lld)",
              (long long)shape[0], (long long)shape[1], (long long)shape[2], (long long)h->max_shape[0],
              (long long)h->max_shape[1], (long long)h->max_shape[2]);
  return BSMI_OK;
}

int bsmi_ws_fragments_u8(bsmi_seg* h, const uint8_t* affs_dev, const int64_t shape[3], int fragments_in_xy,
                         int min_seed_distance, uint64_t* frags_dev, uint64_t* max_id_dev, void* stream) {
  return bsmi_ws_fragments_seeds_u8(h, affs_dev, shape, fragments_in_xy, min_seed_distance, frags_dev, max_id_dev, nullptr, stream);
}

int bsmi_ws_fragments_seeds_u8(bsmi_seg* h, const uint8_t* affs_dev, const int64_t shape[3], int fragments_in_xy,
                               int min_seed_distance, uint64_t* frags_dev, uint64_t* max_id_dev, uint64_t* seeds_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !max_id_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (min_seed_distance < 1 || min_seed_distance > 64) BSMI_FAIL(BSMI_ERR_INVALID, "min_seed_distance out of range");
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  if (!fragments_in_xy) {
    const size_t n = (size_t)D * H * W;
    if (n >= ((size_t)1 << 23) || (size_t)D * D + (size_t)H * H + (size_t)W * W + 2 * D + 1 >= ((size_t)1 << 18))
      BSMI_FAIL(BSMI_ERR_INVALID, "3-D watershed: volumes of 2^23 voxels or more are not supported by the flood's queue entries");
    WsScratch& w = h->ws;
    FragWs& f = h->frag;
    const int bs = 256;
    const int grid = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
    int* any_bg = (int*)h->status_dev;
    BSMI_HIP(hipMemsetAsync(any_bg, 0, sizeof(int), s));
    hipLaunchKernelGGL(ws3_mask_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, n, w.mask, any_bg);
    hipLaunchKernelGGL(ws3_edt_x_kernel, dim3((D * H + 63) / 64), dim3(64), 0, s, (const uint8_t*)w.mask, D * H, W, w.g);
    hipLaunchKernelGGL(ws3_edt_axis_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.g, D, H, W, 1, w.mf);
    hipLaunchKernelGGL(ws3_edt_axis_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.mf, D, H, W, 0, w.d2);
    hipLaunchKernelGGL(ws3_edt_nobg_kernel, dim3(grid), dim3(bs), 0, s, D, H, W, (const int*)any_bg, w.d2);
    hipLaunchKernelGGL(ws3_maxfilter_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.d2, D, H, W, 2, min_seed_distance, w.g);
    hipLaunchKernelGGL(ws3_maxfilter_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.g, D, H, W, 1, min_seed_distance, w.mf);
    hipLaunchKernelGGL(ws3_maxfilter_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.mf, D, H, W, 0, min_seed_distance, w.g);
    f.par = w.par;
    hipLaunchKernelGGL(ws3_maxima_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.d2, (const int32_t*)w.g, n, h->crop_tmp, f);
    hipLaunchKernelGGL(cc6_union_kernel, dim3(grid), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, D, H, W, f);
    const uint32_t nblk = (uint32_t)((n + 1023) / 1024);
    hipLaunchKernelGGL(cc26_count_kernel, dim3(nblk), dim3(1024), 0, s, n, f);
    hipLaunchKernelGGL(cc26_scan_kernel, dim3(1), dim3(1024), 0, s, nblk, f, max_id_dev);
    hipLaunchKernelGGL(cc26_rank_kernel, dim3(nblk), dim3(1024), 0, s, n, f);
    hipLaunchKernelGGL(ws3_markers_kernel, dim3(grid), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, (const uint8_t*)w.mask, n, f, w.lab, seeds_dev);
    if (h->host_flood3) {
      // the caller waits for this one result: mask, distances and markers to the host, the sequential flood there, labels back
      BSMI_HIP(hipGetLastError());
      std::vector<uint8_t> hmask(n);
      std::vector<int32_t> hd2(n), hlab(n);
      BSMI_HIP(hipMemcpyAsync(hmask.data(), w.mask, n, hipMemcpyDeviceToHost, s));
      BSMI_HIP(hipMemcpyAsync(hd2.data(), w.d2, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      BSMI_HIP(hipMemcpyAsync(hlab.data(), w.lab, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      BSMI_HIP(hipStreamSynchronize(s));
      host_flood3(D, H, W, hmask.data(), hd2.data(), hlab.data());
      BSMI_HIP(hipMemcpyAsync(w.lab, hlab.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(ws3_labels_out_kernel, dim3(grid), dim3(bs), 0, s, (const int32_t*)w.lab, n, frags_dev);
      BSMI_HIP(hipGetLastError());
      BSMI_HIP(hipStreamSynchronize(s));  // (the host buffers go out of scope)
      return BSMI_OK;
    }
    hipLaunchKernelGGL(ws3_flood_kernel, dim3(1), dim3(64), 0, s, D, H, W, w, h->flood_spill, frags_dev);
    BSMI_HIP(hipGetLastError());
    return BSMI_OK;
  }
  WsScratch wsx = h->ws;
  bool compact = false;  // the flood's per-voxel state as one 32-bit record (set with the seeds kernel's LDS path below)
  wsx.seedlab = nullptr;
  if (seeds_dev) {
    if (!h->seedlab) {  // allocated on first use
      void* q = nullptr;
      BSMI_HIP(hipMalloc(&q, h->max_vox * sizeof(int32_t)));
      h->allocs.push_back(q);
      h->seedlab = (int32_t*)q;
    }
    wsx.seedlab = h->seedlab;
  }
  {
    // squared distances must fit the uint16 intermediates of the LDS path: H^2 + W^2 < 65535
    // sg u16 [H][W+2] | sd2 u16 [H*W] | smask u8 [H*W] | sflag u8 [H*W], each 16-byte aligned (ws_seeds_kernel)
    const size_t hw16 = ((size_t)H * W + 15) & ~(size_t)15;
    const size_t lds = (((size_t)H * (W + 2) * 2 + 15) & ~(size_t)15) + (((size_t)H * W * 2 + 15) & ~(size_t)15) + 2 * hw16;
    const bool use_lds = lds <= 158 * 1024 && (size_t)H * H + (size_t)W * W < 65535 && (H + 1) * (H + 1) + W * W < 65535;
    // (then also H * W < 32768: a slice's marker labels and voxel indices fit the compact record's 16 bits)
    static const bool compact_ok = [] { const char* e = getenv("BSMI_FLOOD_COMPACT"); return !(e && e[0] == '0'); }();
    compact = use_lds && compact_ok;
    if (use_lds) {
      static DeviceOnce once;
      const int rc_once = once.run([&]() -> int {
        BSMI_HIP(hipFuncSetAttribute((const void*)ws_seeds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        return BSMI_OK;
      });
      if (rc_once) return rc_once;
      hipLaunchKernelGGL(ws_seeds_kernel<true>, dim3(D), dim3(WS_T), lds, s, affs_dev, D, H, W, min_seed_distance, wsx, compact ? 1 : 0);
    } else {
      hipLaunchKernelGGL(ws_seeds_kernel<false>, dim3(D), dim3(WS_T), 0, s, affs_dev, D, H, W, min_seed_distance, wsx, 0);
    }
  }
  hipLaunchKernelGGL(ws_offsets_kernel, dim3(1), dim3(64), 0, s, D, h->ws, max_id_dev);
  if (seeds_dev)
    hipLaunchKernelGGL(ws_seeds_out_kernel, dim3((unsigned)std::min<size_t>(((size_t)D * H * W + 255) / 256, 4096)), dim3(256), 0, s, D,
                       (size_t)H * W, wsx, seeds_dev);
  if (compact)
    hipLaunchKernelGGL(ws_flood_kernel<true>, dim3((D + FLOOD_WAVES - 1) / FLOOD_WAVES), dim3(64 * FLOOD_WAVES), 0, s, D, H, W, h->ws, h->flood_spill,
                       h->flood_spill_stride, frags_dev, h->status_dev);
  else
    hipLaunchKernelGGL(ws_flood_kernel<false>, dim3((D + FLOOD_WAVES - 1) / FLOOD_WAVES), dim3(64 * FLOOD_WAVES), 0, s, D, H, W, h->ws, h->flood_spill,
                       h->flood_spill_stride, frags_dev, h->status_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// BSMI_AGG_FAST=0 (tests): the general forms of the merge loops (state in the workspace's HBM arrays)
static bool agg_fast_enabled() {
  static const bool fast = [] { const char* e = getenv("BSMI_AGG_FAST"); return !(e && e[0] == '0'); }();
  return fast;
}

int bsmi_agglomerate_mean_u8(bsmi_seg* h, const uint8_t* affs_dev, const uint64_t* frags_dev, const int64_t shape[3],
                             const float* thresholds_host, int n_thresholds, uint64_t* segs_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !thresholds_host || !segs_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_thresholds < 1 || n_thresholds > kMaxThresholds) BSMI_FAIL(BSMI_ERR_INVALID, "1..%d thresholds supported", kMaxThresholds);
  for (int i = 0; i < n_thresholds; ++i) {
    if (!(thresholds_host[i] >= 0.f)) BSMI_FAIL(BSMI_ERR_INVALID, "thresholds must be >= 0");
    if (i && thresholds_host[i] < thresholds_host[i - 1]) BSMI_FAIL(BSMI_ERR_INVALID, "thresholds must be ascending");
  }
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  const size_t n = (size_t)D * H * W;
  AggWs& g = h->agg;
  AggThresholds thr;
  static_assert(kMaxThresholds <= 16, "AggThresholds holds 16 values");
  for (int i = 0; i < 16; ++i) thr.v[i] = i < n_thresholds ? thresholds_host[i] : 0.f;
  BSMI_HIP(hipMemsetAsync(g.counters, 0, 8 * sizeof(uint32_t), s));
  BSMI_HIP(hipMemsetAsync(g.maxid, 0, sizeof(uint64_t), s));
  // The scans run as FEW, FAT workgroups (kScanGrid x 1024 threads, grid-stride loops): a lane shares the GPU with
  // the U-Net, whose persistent conv workgroups each need a completely free CU; a 2048-workgroup scan (or a runtime
  // fill kernel) puts a wave on every CU and keeps them all away until it has drained.
  const int bs = 1024;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, (size_t)seg_scan_grid());
  hipLaunchKernelGGL(seg_clear_kernel, dim3(grid), dim3(bs), 0, s, g);
  hipLaunchKernelGGL(agg_maxid_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, g);
  hipLaunchKernelGGL(agg_mark_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, g);
  hipLaunchKernelGGL(agg_rank_kernel, dim3(1), dim3(1024), 0, s, g);
  hipLaunchKernelGGL(agg_edges_kernel<false>, dim3(grid), dim3(bs), 0, s, affs_dev, frags_dev, D, H, W, g);
  hipLaunchKernelGGL(agg_compact_kernel, dim3(grid), dim3(bs), 0, s, g);
  {
    static DeviceOnce once;
    const bool attr_set = once.run([&]() -> int {
      BSMI_HIP(hipFuncSetAttribute((const void*)agg_edge_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRankMax * sizeof(uint64_t))));
      return BSMI_OK;
    }) == BSMI_OK;
    // BSMI_AGG_FAST=0 (tests): leave the edges unranked, i.e. take the general form of the merge loop
    if (attr_set && agg_fast_enabled()) hipLaunchKernelGGL(agg_edge_rank_kernel, dim3(1), dim3(1024), kRankMax * sizeof(uint64_t), s, g);
  }
  hipLaunchKernelGGL(agg_merge_kernel, dim3(8), dim3(64), 0, s, g, thr, n_thresholds);
  hipLaunchKernelGGL(agg_relabel_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, n_thresholds, g, segs_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_agglomerate_hist_u8(bsmi_seg* h, const uint8_t* affs_dev, const uint64_t* frags_dev, const int64_t shape[3],
                             const float* thresholds_host, int n_thresholds, int quantile, int init_with_max, uint64_t* segs_dev,
                             void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !thresholds_host || !segs_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_thresholds < 1 || n_thresholds > kMaxThresholds) BSMI_FAIL(BSMI_ERR_INVALID, "1..%d thresholds supported", kMaxThresholds);
  if (quantile < 0 || quantile > 100) BSMI_FAIL(BSMI_ERR_INVALID, "quantile %d outside 0..100", quantile);
  for (int i = 0; i < n_thresholds; ++i) {
    if (!(thresholds_host[i] >= 0.f)) BSMI_FAIL(BSMI_ERR_INVALID, "thresholds must be >= 0");
    if (i && thresholds_host[i] < thresholds_host[i - 1]) BSMI_FAIL(BSMI_ERR_INVALID, "thresholds must be ascending");
  }
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  const size_t n = (size_t)D * H * W;
  AggWs& g = h->agg;
  // region graph on the device, as in bsmi_agglomerate_mean_u8
  BSMI_HIP(hipMemsetAsync(g.counters, 0, 8 * sizeof(uint32_t), s));
  BSMI_HIP(hipMemsetAsync(g.maxid, 0, sizeof(uint64_t), s));
  const int bs = 1024;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, (size_t)seg_scan_grid());
  hipLaunchKernelGGL(seg_clear_kernel, dim3(grid), dim3(bs), 0, s, g);
  hipLaunchKernelGGL(agg_maxid_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, g);
  hipLaunchKernelGGL(agg_mark_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, g);
  hipLaunchKernelGGL(agg_rank_kernel, dim3(1), dim3(1024), 0, s, g);
  hipLaunchKernelGGL(agg_edges_kernel<false>, dim3(grid), dim3(bs), 0, s, affs_dev, frags_dev, D, H, W, g);
  hipLaunchKernelGGL(agg_compact_kernel, dim3(grid), dim3(bs), 0, s, g);
  BSMI_HIP(hipGetLastError());
  uint32_t c[8];
  BSMI_HIP(hipMemcpyAsync(c, g.counters, sizeof c, hipMemcpyDeviceToHost, s));
  BSMI_HIP(hipStreamSynchronize(s));
  if (c[3]) BSMI_FAIL(BSMI_ERR_OVERFLOW, "agglomeration workspace overflow (flags 0x%x: 1 id range, 2 nodes, 4 hash, 8 edges)", c[3]);
  const uint32_t nn = c[0], ne = c[1];
  // per-edge histograms: second scan, then the merge loop on the host (agglo_host.cpp), then the relabel on the device
  // 1 KiB of histogram per edge, on the device and again on the host: say so before a few million edges end in a bare
  // out-of-memory error (this entry point serves the whole-ROI simple_watershed; the block pipeline never builds histograms)
  const size_t hist_bytes = (size_t)ne * 256 * sizeof(uint32_t);
  {
    size_t free_b = 0, total_b = 0;
    BSMI_HIP(hipMemGetInfo(&free_b, &total_b));
    if (hist_bytes > free_b - free_b / 8)
      BSMI_FAIL(BSMI_ERR_OVERFLOW, "histogram-quantile agglomeration of %u edges needs %.1f GB of histograms (1 KiB per edge) on the device and the host, %.1f GB of device memory are free: segment the volume blockwise, or with the mean scorer", ne,
                hist_bytes / 1e9, free_b / 1e9);
  }
  std::vector<uint32_t> eu, ev, hist, roots;
  try {
    eu.resize(ne); ev.resize(ne); hist.resize((size_t)ne * 256); roots.resize((size_t)n_thresholds * std::max(nn, 1u));
  } catch (const std::bad_alloc&) {
    BSMI_FAIL(BSMI_ERR_OVERFLOW, "histogram-quantile agglomeration of %u edges: no %.1f GB of host memory for the histograms", ne, hist_bytes / 1e9);
  }
  if (ne) {
    uint32_t* hist_dev = nullptr;
    BSMI_HIP(hipMalloc((void**)&hist_dev, (size_t)ne * 256 * sizeof(uint32_t)));
    hipError_t err = hipMemsetAsync(hist_dev, 0, (size_t)ne * 256 * sizeof(uint32_t), s);
    if (err == hipSuccess) {
      hipLaunchKernelGGL(agg_hist_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, frags_dev, D, H, W, g, hist_dev);
      err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(hist.data(), hist_dev, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipMemcpyAsync(eu.data(), g.eu, ne * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipMemcpyAsync(ev.data(), g.ev, ne * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    (void)hipFree(hist_dev);
    BSMI_HIP(err);
  }
  host_agglomerate_hist(nn, ne, eu.data(), ev.data(), hist.data(), quantile, init_with_max, thresholds_host, n_thresholds, roots.data());
  for (int t = 0; t < n_thresholds && nn; ++t)
    BSMI_HIP(hipMemcpyAsync(g.roots + (size_t)t * g.node_cap, roots.data() + (size_t)t * nn, nn * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(agg_relabel_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, n_thresholds, g, segs_dev);
  BSMI_HIP(hipGetLastError());
  BSMI_HIP(hipStreamSynchronize(s));  // `roots` (host memory) must outlive the copies
  return BSMI_OK;
}

int bsmi_frag_postprocess_u8(bsmi_seg* h, const uint8_t* affs_dev, uint64_t* frags_dev, const int64_t shape[3],
                             double filter_value, int64_t min_size, const int64_t crop_offset[3],
                             const int64_t crop_shape[3], uint64_t id_offset, uint64_t* out_dev, uint64_t* num_labels_dev,
                             void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !crop_offset || !crop_shape || !out_dev || !num_labels_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  for (int d = 0; d < 3; ++d)
    if (crop_offset[d] < 0 || crop_shape[d] < 1 || crop_offset[d] + crop_shape[d] > shape[d])
      BSMI_FAIL(BSMI_ERR_INVALID, "crop outside the fragment volume");
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  const size_t nc = (size_t)crop_shape[0] * crop_shape[1] * crop_shape[2];
  FragWs& f = h->frag;
  const int bs = 256;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
  const int gridc = (int)std::min<size_t>((nc + bs - 1) / bs, 4096);
  {
    Fills fl;
    fl.add(f.flags, 4 * sizeof(uint32_t));
    if (filter_value > 0.0 || min_size > 0) {
      fl.add(f.lsum, (size_t)f.id_cap * sizeof(unsigned long long));
      fl.add(f.lcnt, (size_t)f.id_cap * sizeof(uint32_t));
    }
    fl.launch(s);
  }
  if (filter_value > 0.0 || min_size > 0) {
    hipLaunchKernelGGL(frag_stats_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, (const uint64_t*)frags_dev, n, f);
    hipLaunchKernelGGL(frag_decide_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, (const uint64_t*)frags_dev, n, f, filter_value,
                       (long long)min_size);
    hipLaunchKernelGGL(frag_filter_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, f);
  }
  hipLaunchKernelGGL(crop_u64_kernel, dim3(gridc), dim3(bs), 0, s, (const uint64_t*)frags_dev, (int)shape[1], (int)shape[2],
                     (int)crop_offset[0], (int)crop_offset[1], (int)crop_offset[2], (int)crop_shape[0], (int)crop_shape[1],
                     (int)crop_shape[2], h->crop_tmp);
  const uint32_t nblk = (uint32_t)((nc + 1023) / 1024);
  hipLaunchKernelGGL(cc26_init_kernel, dim3(gridc), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, nc, f);
  hipLaunchKernelGGL(cc26_union_kernel, dim3(gridc), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, (int)crop_shape[0],
                     (int)crop_shape[1], (int)crop_shape[2], f);
  hipLaunchKernelGGL(cc26_count_kernel, dim3(nblk), dim3(1024), 0, s, nc, f);
  hipLaunchKernelGGL(cc26_scan_kernel, dim3(1), dim3(1024), 0, s, nblk, f, num_labels_dev);
  hipLaunchKernelGGL(cc26_rank_kernel, dim3(nblk), dim3(1024), 0, s, nc, f);
  hipLaunchKernelGGL(cc26_write_kernel, dim3(gridc), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, nc, f, id_offset, out_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_label_stats(bsmi_seg* h, const uint64_t* labels_dev, const int64_t shape[3], uint64_t id_offset, uint64_t num,
                     uint64_t* size_dev, uint64_t* sums_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!labels_dev || !size_dev || !sums_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  {
    Fills fl;
    fl.add(size_dev, num * sizeof(uint64_t));
    fl.add(sums_dev, 3 * num * sizeof(uint64_t));
    fl.launch(s);
  }
  const int bs = 256;
  hipLaunchKernelGGL(label_stats_kernel, dim3((int)std::min<size_t>((n + bs - 1) / bs, 4096)), dim3(bs), 0, s, labels_dev,
                     (int)shape[0], (int)shape[1], (int)shape[2], id_offset, num, (unsigned long long*)size_dev,
                     (unsigned long long*)sums_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// ids -> ranks, region graph, bin-queue merge loop up to `threshold` (the common front of the two RAG entry points)
static int rag_build_and_merge(bsmi_seg* h, const uint8_t* affs_dev, const uint64_t* frags_dev, const int64_t shape[3], float threshold,
                               int discretize_queue, uint64_t* counts_dev, hipStream_t s) {
  BSMI_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  AggWs& g = h->agg;
  {
    Fills fl;
    fl.add(g.counters, 8 * sizeof(uint32_t));
    if (counts_dev) fl.add(counts_dev, 3 * sizeof(uint64_t));
    fl.add(g.idkeys, (size_t)g.icap * sizeof(uint64_t), 0xffffffffu);
    fl.add(g.hkeys, (size_t)g.hcap * sizeof(uint64_t), 0xffffffffu);
    fl.add(g.hsum, (size_t)g.hcap * sizeof(unsigned long long));
    fl.add(g.hcnt, (size_t)g.hcap * sizeof(uint32_t));
    fl.launch(s);
  }
  const int bs = 256;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
  hipLaunchKernelGGL(rag_ids_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, W, g);
  hipLaunchKernelGGL(rag_pad_kernel, dim3(256), dim3(bs), 0, s, g);
  size_t tb = h->sort_tmp_bytes;
  BSMI_HIP(hipcub::DeviceRadixSort::SortKeys(h->sort_tmp, tb, (const uint64_t*)g.idu, g.ids, (int)g.node_cap, 0, 64, s));
  hipLaunchKernelGGL(rag_rank_kernel, dim3(256), dim3(bs), 0, s, g);
  hipLaunchKernelGGL(agg_edges_kernel<true>, dim3(grid), dim3(bs), 0, s, affs_dev, frags_dev, D, H, W, g);
  hipLaunchKernelGGL(rag_iota_kernel, dim3(1024), dim3(bs), 0, s, g);
  tb = h->sort_tmp_bytes;
  BSMI_HIP(hipcub::DeviceRadixSort::SortPairs(h->sort_tmp, tb, (const uint64_t*)g.hkeys, g.skeys, (const uint32_t*)g.iota, g.sslot,
                                              (int)g.hcap, 0, 64, s));
  hipLaunchKernelGGL(rag_compact_kernel, dim3(std::min<uint32_t>(g.hcap / bs, 2048u)), dim3(bs), 0, s, g);
  hipLaunchKernelGGL(rag_merge_kernel, dim3(8), dim3(64), 0, s, g, threshold, discretize_queue, agg_fast_enabled() ? 1 : 0);
  return BSMI_OK;
}

int bsmi_rag_graph_u8(bsmi_seg* h, const uint8_t* affs_dev, const uint64_t* frags_dev, const int64_t shape[3], uint64_t* edges_dev,
                      uint64_t* sums_dev, uint32_t* pair_counts_dev, uint64_t edge_capacity, uint64_t* counts_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !edges_dev || !sums_dev || !pair_counts_dev || !counts_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  BSMI_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  AggWs& g = h->agg;
  {
    Fills fl;
    fl.add(g.counters, 8 * sizeof(uint32_t));
    fl.add(counts_dev, 3 * sizeof(uint64_t));
    fl.add(g.idkeys, (size_t)g.icap * sizeof(uint64_t), 0xffffffffu);
    fl.add(g.hkeys, (size_t)g.hcap * sizeof(uint64_t), 0xffffffffu);
    fl.add(g.hsum, (size_t)g.hcap * sizeof(unsigned long long));
    fl.add(g.hcnt, (size_t)g.hcap * sizeof(uint32_t));
    fl.launch(s);
  }
  const int bs = 256;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
  hipLaunchKernelGGL(rag_ids_kernel, dim3(grid), dim3(bs), 0, s, frags_dev, n, (int)shape[2], g);
  hipLaunchKernelGGL(agg_edges_kernel<true>, dim3(grid), dim3(bs), 0, s, affs_dev, frags_dev, (int)shape[0], (int)shape[1], (int)shape[2], g);
  hipLaunchKernelGGL(rag_graph_hash_out_kernel, dim3(std::min<uint32_t>(g.hcap / bs, 2048u)), dim3(bs), 0, s, g, edges_dev, sums_dev, pair_counts_dev,
                     edge_capacity, counts_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_rag_merge_scores_u8(bsmi_seg* h, const uint8_t* affs_dev, const uint64_t* frags_dev, const int64_t shape[3],
                             float threshold, int discretize_queue, uint64_t* edges_dev, float* scores_dev,
                             uint64_t edge_capacity, uint64_t* merges_dev, float* merge_scores_dev, uint64_t* counts_dev,
                             void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !edges_dev || !scores_dev || !counts_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (discretize_queue < 1 || discretize_queue > kMaxQueueBins)
    BSMI_FAIL(BSMI_ERR_INVALID, "discretize_queue must be in [1, %d] (the exact-order queue is bsmi_agglomerate_mean_u8)", kMaxQueueBins);
  if (!(threshold > 0.f)) BSMI_FAIL(BSMI_ERR_INVALID, "threshold must be positive");
  hipStream_t s = (hipStream_t)stream;
  rc = rag_build_and_merge(h, affs_dev, frags_dev, shape, threshold, discretize_queue, counts_dev, s);
  if (rc) return rc;
  const int bs = 256;
  hipLaunchKernelGGL(rag_scores_kernel, dim3(1024), dim3(bs), 0, s, h->agg, edges_dev, scores_dev, edge_capacity, merges_dev,
                     merge_scores_dev, counts_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_rag_agglomerate_u8(bsmi_seg* h, const uint8_t* affs_dev, uint64_t* frags_dev, const int64_t shape[3], float threshold,
                            int discretize_queue, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (discretize_queue < 1 || discretize_queue > kMaxQueueBins) BSMI_FAIL(BSMI_ERR_INVALID, "discretize_queue must be in [1, %d]", kMaxQueueBins);
  if (!(threshold > 0.f)) BSMI_FAIL(BSMI_ERR_INVALID, "threshold must be positive");
  hipStream_t s = (hipStream_t)stream;
  rc = rag_build_and_merge(h, affs_dev, frags_dev, shape, threshold, discretize_queue, nullptr, s);
  if (rc) return rc;
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  hipLaunchKernelGGL(rag_relabel_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, frags_dev, n, h->agg);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_rag_edge_stats(bsmi_seg* h, uint64_t* sums_dev, uint64_t* counts_dev, uint64_t capacity, void* stream) {
  if (!h || !sums_dev || !counts_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  BSMI_HIP(hipSetDevice(h->device));
  hipLaunchKernelGGL(rag_edge_stats_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, h->agg, sums_dev, counts_dev, capacity);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_lut_relabel(int device, const uint64_t* in_dev, uint64_t n, const uint64_t* keys_dev, const uint64_t* vals_dev, uint64_t m,
                     uint64_t* out_dev, void* stream) {
  if (!in_dev || !out_dev || (m && (!keys_dev || !vals_dev))) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  BSMI_HIP(hipSetDevice(device));
  if (!n) return BSMI_OK;
  const int bs = 256;
  hipLaunchKernelGGL(lut_relabel_kernel, dim3((int)std::min<uint64_t>((n + bs - 1) / bs, 8192)), dim3(bs), 0, (hipStream_t)stream,
                     in_dev, (size_t)n, keys_dev, vals_dev, m, out_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_lut_relabel_multi(int device, const uint64_t* in_dev, uint64_t n, const uint64_t* keys_dev, const uint64_t* vals_dev, uint64_t m,
                           int n_columns, uint64_t* out_dev, void* stream) {
  if (!in_dev || !out_dev || (m && (!keys_dev || !vals_dev))) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_columns < 1 || n_columns > kLutMaxColumns) BSMI_FAIL(BSMI_ERR_INVALID, "1 to %d value columns", kLutMaxColumns);
  BSMI_HIP(hipSetDevice(device));
  if (!n) return BSMI_OK;
  LutColumns c{};
  for (int t = 0; t < n_columns; ++t) {
    c.vals[t] = vals_dev ? vals_dev + (size_t)t * m : nullptr;
    c.out[t] = out_dev + (size_t)t * n;
  }
  const int bs = 256;
  const uint64_t nruns = (n + 7) / 8;
  hipLaunchKernelGGL(lut_relabel_multi_kernel, dim3((int)std::min<uint64_t>((nruns + bs - 1) / bs, 16384)), dim3(bs), 0, (hipStream_t)stream,
                     in_dev, (size_t)n, keys_dev, m, n_columns, c);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// Host-side union-find over the scored RAG (reference post/watershed.py:182,
// funlib.segment.graphs.impl.connected_components [EXT]); same contract as oracle seg_connected_components:
// score <= threshold joins, a component is named by its smallest node id, nodes ascending.
int bsmi_connected_components_multi(const uint64_t* nodes, uint64_t n, const uint64_t* edges, const float* scores, uint64_t m,
                                    const float* thresholds, int n_thresholds, uint64_t* components) {
  if ((n && (!nodes || !components)) || (m && (!edges || !scores)) || !thresholds || n_thresholds < 1)
    BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n >= 0xffffffffull) BSMI_FAIL(BSMI_ERR_INVALID, "%llu nodes: the union-find indices are 32-bit", (unsigned long long)n);
  for (uint64_t i = 1; i < n; ++i)
    if (nodes[i] <= nodes[i - 1]) BSMI_FAIL(BSMI_ERR_INVALID, "nodes must be strictly ascending");
  // node ids -> indices, once for all thresholds, on a few host threads (two binary searches per edge)
  constexpr uint32_t kNone = 0xffffffffu;
  const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  std::vector<uint32_t> iu(m), iv(m);
  {
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const unsigned T = m < 20000 ? 1u : hw;  // 143 000 edges on one thread: 5 ms of the job's last 9
    // two-level search: every 64th id in a table that stays in cache, then the 64 ids of one run (8 cache lines)
    constexpr uint64_t kRun = 64;
    std::vector<uint64_t> coarse((n + kRun - 1) / kRun);
    for (size_t c = 0; c < coarse.size(); ++c) coarse[c] = nodes[c * kRun];
    auto index_of = [&](uint64_t id) -> uint32_t {
      if (!n || id < nodes[0]) return kNone;
      const size_t c = (size_t)(std::upper_bound(coarse.begin(), coarse.end(), id) - coarse.begin()) - 1;
      const uint64_t* lo = nodes + c * kRun;
      const uint64_t* hi = nodes + std::min<uint64_t>(n, (c + 1) * kRun);
      const uint64_t* p = std::lower_bound(lo, hi, id);
      return (p != hi && *p == id) ? (uint32_t)(p - nodes) : kNone;
    };
    auto work = [&](uint64_t e0, uint64_t e1) {
      for (uint64_t e = e0; e < e1; ++e) {
        const uint32_t a = index_of(edges[2 * e]), b = index_of(edges[2 * e + 1]);
        const bool ok = a != kNone && b != kNone;
        iu[e] = ok ? a : kNone;
        iv[e] = ok ? b : kNone;
      }
    };
    if (T == 1) {
      work(0, m);
    } else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t) th.emplace_back(work, m * t / T, m * (t + 1) / T);
      for (auto& x : th) x.join();
    }
  }
  const bool dbg = getenv("BSMI_CC_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_map = now();
  if (dbg) fprintf(stderr, "[cc] look-up of %llu edges: %.3f s\n", (unsigned long long)m, t_map - t_begin);
  std::vector<uint32_t> parent(n);
  for (uint64_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
  auto find = [&](uint32_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  };
  // thresholds in ascending order: the edges of a lower threshold are among those of a higher one, and with the smaller
  // index as the root a component's name does not depend on the order of the unions
  std::vector<int> order(n_thresholds);
  for (int k = 0; k < n_thresholds; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return thresholds[a] < thresholds[b]; });
  float prev = -INFINITY;
  bool first = true;
  for (int k : order) {
    const float t = thresholds[k];
    for (uint64_t e = 0; e < m; ++e) {
      const float sc = scores[e];
      if (!(sc <= t) || (!first && sc <= prev) || iu[e] == kNone) continue;
      const uint32_t a = find(iu[e]), b = find(iv[e]);
      if (a == b) continue;
      if (a < b) parent[b] = a; else parent[a] = b;
    }
    const double t_un = now();
    uint64_t* out = components + (size_t)k * n;
    for (uint64_t i = 0; i < n; ++i) out[i] = nodes[find((uint32_t)i)];
    if (dbg) fprintf(stderr, "[cc] threshold %g: unions until %.3f, snapshot %.3f s\n", t, t_un - t_map, now() - t_un);
    prev = t;
    first = false;
  }
  return BSMI_OK;
}

int bsmi_connected_components(const uint64_t* nodes, uint64_t n, const uint64_t* edges, const float* scores, uint64_t m,
                              float threshold, uint64_t* components) {
  return bsmi_connected_components_multi(nodes, n, edges, scores, m, &threshold, 1, components);
}

int bsmi_cc_affs_u8(bsmi_seg* h, const uint8_t* affs_dev, const int64_t shape[3], int cut, int64_t min_size, uint64_t* frags_dev,
                    uint64_t* seg_dev, uint64_t* num_labels_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!affs_dev || !frags_dev || !num_labels_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (cut < -1 || cut > 255) BSMI_FAIL(BSMI_ERR_INVALID, "cut must be in [-1, 255]");
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)shape[0] * shape[1] * shape[2];
  FragWs& f = h->frag;
  const int bs = 256;
  const int grid = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
  const uint32_t nblk = (uint32_t)((n + 1023) / 1024);
  BSMI_HIP(hipMemsetAsync(f.flags, 0, 4 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(ccaff_init_kernel, dim3(grid), dim3(bs), 0, s, n, f, h->crop_tmp);
  hipLaunchKernelGGL(ccaff_union_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, (int)shape[0], (int)shape[1], (int)shape[2], cut, f, h->crop_tmp);
  hipLaunchKernelGGL(ccaff_finalize_kernel, dim3(grid), dim3(bs), 0, s, n, f, (const uint64_t*)h->crop_tmp);
  hipLaunchKernelGGL(cc26_count_kernel, dim3(nblk), dim3(1024), 0, s, n, f);
  hipLaunchKernelGGL(cc26_scan_kernel, dim3(1), dim3(1024), 0, s, nblk, f, num_labels_dev);
  hipLaunchKernelGGL(cc26_rank_kernel, dim3(nblk), dim3(1024), 0, s, n, f);
  hipLaunchKernelGGL(cc26_write_kernel, dim3(grid), dim3(bs), 0, s, (const uint64_t*)h->crop_tmp, n, f, (uint64_t)0, frags_dev);
  if (seg_dev) {
    BSMI_HIP(hipMemcpyAsync(seg_dev, frags_dev, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    if (min_size > 0) {  // skimage remove_small_objects on the labels (post/connected_components.py:97-101)
      BSMI_HIP(hipMemsetAsync(f.lsum, 0, (size_t)f.id_cap * sizeof(unsigned long long), s));
      BSMI_HIP(hipMemsetAsync(f.lcnt, 0, (size_t)f.id_cap * sizeof(uint32_t), s));
      hipLaunchKernelGGL(frag_stats_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, (const uint64_t*)seg_dev, n, f);
      hipLaunchKernelGGL(frag_decide_kernel, dim3(grid), dim3(bs), 0, s, affs_dev, (const uint64_t*)seg_dev, n, f, 0.0, (long long)min_size);
      hipLaunchKernelGGL(frag_filter_kernel, dim3(grid), dim3(bs), 0, s, seg_dev, n, f);
    }
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_label_table_u64(bsmi_seg* h, const uint64_t* labels_dev, const int64_t shape[3], int64_t z0, uint64_t* ids_dev, uint64_t* counts_dev,
                         int32_t* zmin_dev, int32_t* zmax_dev, uint64_t capacity, uint64_t* n_dev, void* stream) {
  int rc = check_seg_shape(h, shape);
  if (rc) return rc;
  if (!labels_dev || !ids_dev || !counts_dev || !zmin_dev || !zmax_dev || !n_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  AggWs& g = h->agg;
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  BSMI_HIP(hipMemsetAsync(g.counters, 0, 8 * sizeof(uint32_t), s));
  BSMI_HIP(hipMemsetAsync(n_dev, 0, sizeof(uint64_t), s));
  hipLaunchKernelGGL(ltab_clear_kernel, dim3(512), dim3(256), 0, s, g);
  const size_t nrows = (size_t)D * H;
  hipLaunchKernelGGL(ltab_scan_kernel, dim3((unsigned)std::min<size_t>((nrows + 63) / 64, 8192)), dim3(64), 0, s, labels_dev, D, H, W, (int)z0, g);
  hipLaunchKernelGGL(ltab_compact_kernel, dim3(512), dim3(256), 0, s, g);
  hipLaunchKernelGGL(ltab_pad_kernel, dim3(256), dim3(256), 0, s, g);
  size_t tb = h->sort_tmp_bytes;
  BSMI_HIP(hipcub::DeviceRadixSort::SortPairs(h->sort_tmp, tb, (const uint64_t*)g.idu, g.ids, (const uint32_t*)g.ha, g.hb, (int)g.node_cap, 0, 64, s));
  hipLaunchKernelGGL(ltab_gather_kernel, dim3(256), dim3(256), 0, s, g, ids_dev, counts_dev, zmin_dev, zmax_dev, capacity, n_dev);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_seg_set_host_flood(bsmi_seg* h, int on) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  h->host_flood3 = on != 0;
  return BSMI_OK;
}

int bsmi_seg_status(bsmi_seg* h, void* stream) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  BSMI_HIP(hipSetDevice(h->device));
  uint32_t c[8], sticky = 0;
  BSMI_HIP(hipMemcpyAsync(c, h->agg.counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t)stream));
  BSMI_HIP(hipMemcpyAsync(&sticky, h->agg.sticky, sizeof sticky, hipMemcpyDeviceToHost, (hipStream_t)stream));
  BSMI_HIP(hipStreamSynchronize((hipStream_t)stream));
  uint32_t ff[4];
  BSMI_HIP(hipMemcpy(ff, h->frag.flags, sizeof ff, hipMemcpyDeviceToHost));
  if (ff[0]) BSMI_FAIL(BSMI_ERR_OVERFLOW, "fragment id exceeds the post-processing table (ids must stay below %u)", h->frag.id_cap);
  // counters[3]: the last call; sticky: every call on this handle since the previous status check (cleared here)
  const uint32_t flags = c[3] | sticky;
  if (sticky) BSMI_HIP(hipMemsetAsync(h->agg.sticky, 0, sizeof(uint32_t), (hipStream_t)stream));
  if (flags)
    BSMI_FAIL(BSMI_ERR_OVERFLOW, "agglomeration workspace overflow (flags 0x%x: 1 id range, 2 nodes, 4 hash, 8 edges, 16 hash churn, 32 edge buffer too small)", flags);
  return BSMI_OK;
}

}  // extern "C"
