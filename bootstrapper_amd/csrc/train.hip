// Training step of the U-Net engine: loss, backward pass, Adam, on the device in fp32 (the reference trains
// in fp32: models/3d_affs/train.py, training.py:96-137).
//
// Reference being replaced (paths relative to /root/reference/bootstrapper):
//   models/3d_affs/train.py:152-159   training_step = loss(model(raw), gt, weights); Adam(lr = 0.5e-4)
//   models/3d_affs/model.py:67-92     WeightedMSELoss (masked mean of w (p - t)^2)
//   models/3d_affs/unet.py            autograd of ConvPass / Downsample / Upsample, restated here as explicit kernels
//
// The forward pass is the inference engine's own (bsmi_unet_forward in BSMI_PREC_F32: exact f32 MFMA, every
// activation of the block stays resident), so the backward pass walks the same launch plan in reverse:
//   conv stage  g = dY * [Y > 0] into a zero-bordered tensor; bias gradient = column sums; weight gradient by
//               v_mfma_f32_32x32x2_f32 straight from global memory (g^T x, one kernel tap per workgroup); input
//               gradient = the SAME implicit-GEMM kernel as the forward pass run over the padded g with flipped,
//               transposed weights -- the cropped 1x1x1 residual branch rides along as one more K-step source,
//               exactly as in the forward launch
//   max-pool    gradient routed to the first maximum of each window; trilinear upsample: transposed interpolation
//   head        two 1x1x1 convolutions + sigmoid, fused with the loss gradient
// Parameters, gradients and Adam moments are flat fp32 device buffers in state_dict order (the gradient buffer
// is what the data-parallel all-reduce runs on); after an optimizer step the packed weight images of the
// forward and backward launches are rewritten on the device.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "unet_internal.h"
#include "unet_ops.h"

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

namespace bsmi {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------

// ---- split-bf16 helpers (weight gradients, input gradients) ----
typedef __bf16 wg_bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float wg_f32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) char* wg_gptr_t;
typedef __attribute__((address_space(3))) char* wg_lptr_t;

__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const wg_bf16x2_t h = __builtin_convertvector(wg_f32x2_t{a, b}, wg_bf16x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
  const wg_bf16x2_t l = __builtin_convertvector(wg_f32x2_t{a - ha, b - hb}, wg_bf16x2_t);
  lo = __builtin_bit_cast(uint32_t, l);
}

// one unit (8 floats of K) of a packed weight image: dst[(u / 2) * Npad + n][(u % 2) * 8 + kk] =
// src[wbase + n * sn + (c0 + kk) * sc + tap] for n < nreal, c0 + kk < creal; zero elsewhere
struct PackUnit {
  long long wbase;  // float offset into the flat parameter buffer, -1: padding unit
  int sn, sc, tap, c0, creal, pad;
};

__global__ void pack_weights_kernel(const float* __restrict__ params, const PackUnit* __restrict__ units, int nunits, int Npad,
                                    int nreal, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (unit, n)
  if (i >= (size_t)nunits * Npad) return;
  const int u = (int)(i / Npad), n = (int)(i - (size_t)u * Npad);
  const PackUnit pu = units[u];
  float v[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    v[kk] = 0.f;
    if (pu.wbase >= 0 && n < nreal && pu.c0 + kk < pu.creal) v[kk] = params[pu.wbase + (long long)n * pu.sn + (long long)(pu.c0 + kk) * pu.sc + pu.tap];
  }
  float* d = dst + ((size_t)(u >> 1) * Npad + n) * 16 + (u & 1) * 8;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) d[kk] = v[kk];
}

// the same for a fused split-bf16 launch: units of 16 channels, rows of 32 bf16, a hi image and a lo image (conv_igemm.h)
// Thread -> (unit, n): the units of a window of `ugw` consecutive units (the taps x 2 units of a 32-channel chunk) vary
// fastest, then n: consecutive lanes then read consecutive taps of one (n, c) -- and the next n or c continues the run --
// instead of one cache line per lane (n fastest: 2.9 ms per step for the two images of every layer).
__global__ void pack_weights_x3_kernel(const float* __restrict__ params, const PackUnit* __restrict__ units, int nunits, int Npad, int nreal,
                                       int ugw, uint32_t* __restrict__ hi_img, uint32_t* __restrict__ lo_img) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_window = (size_t)ugw * Npad;
  const int win = (int)(i / per_window);
  const size_t r = i - (size_t)win * per_window;
  const int n = (int)(r / ugw), u = win * ugw + (int)(r - (size_t)n * ugw);
  if (u >= nunits) return;
  const PackUnit pu = units[u];
  const size_t d = (((size_t)(u >> 1) * Npad + n) * 32 + (u & 1) * 16) / 2;  // in bf16 pairs
#pragma unroll
  for (int kk = 0; kk < 16; kk += 2) {
    float v0 = 0.f, v1 = 0.f;
    if (pu.wbase >= 0 && n < nreal) {
      if (pu.c0 + kk < pu.creal) v0 = params[pu.wbase + (long long)n * pu.sn + (long long)(pu.c0 + kk) * pu.sc + pu.tap];
      if (pu.c0 + kk + 1 < pu.creal) v1 = params[pu.wbase + (long long)n * pu.sn + (long long)(pu.c0 + kk + 1) * pu.sc + pu.tap];
    }
    uint32_t h, l;
    split_pair(v0, v1, h, l);
    hi_img[d + kk / 2] = h;
    lo_img[d + kk / 2] = l;
  }
}

// The same through LDS, for windows of a 3 x 3 x 3 layer (ugw = 2 x 27 units of one 32-channel chunk, ordered [tap][half]): a
// workgroup takes PK_NB output channels of one window, reads their 32 x 27 weights in the order they lie in the parameter
// buffer (one run of 3 456 bytes per output channel; the kernel above has every lane walk its own 16 channels, 108 bytes
// apart) and writes the K-steps' rows PK_NB at a time (512 contiguous bytes per image and K-step instead of 32).  The unit
// fields are used as they are: a window that is not of that form (a residual's, padding) is packed correctly, only slower.
constexpr int PK_NB = 8;
template <int NT>  // taps per window (27: constant divisors); 0: ugw / 2 at run time
__global__ __launch_bounds__(256) void pack_weights_x3_t_kernel(const float* __restrict__ params, const PackUnit* __restrict__ units, int nunits,
                                                                int Npad, int nreal, int ugw, uint32_t* __restrict__ hi_img,
                                                                uint32_t* __restrict__ lo_img) {
  extern __shared__ __attribute__((aligned(16))) float pk_sv[];  // [PK_NB][32 channels][taps] values (the order of an OIDHW weight), then the window's units
  PackUnit* su = (PackUnit*)(pk_sv + (size_t)PK_NB * ugw * 16);
  const int tid = threadIdx.x;
  const int nt = NT ? NT : ugw / 2;
  const int u0 = blockIdx.x * ugw, n0 = blockIdx.y * PK_NB;
  for (int i = tid; i < ugw; i += 256) {  // (straight into LDS: a local copy of the struct was a scratch segment)
    if (u0 + i < nunits) {
      su[i] = units[u0 + i];
    } else {
      su[i] = PackUnit{};
      su[i].wbase = -1;
    }
  }
  __syncthreads();
  const int per_n = 32 * nt;
  // forward image: n is the weight's output channel (the outermost index of OIDHW): tap fastest, then the chunk's 32 channels;
  // input-gradient image: n is the weight's INPUT channel (make_dgrad: sn = taps): tap fastest, then the PK_NB values of n
  const bool n_inner = su[0].wbase >= 0 && su[0].sn < su[0].sc;
  for (int e = tid; e < PK_NB * per_n; e += 256) {
    int nl, c, tap;
    if (n_inner) {
      c = e / (PK_NB * nt);
      const int r = e - c * (PK_NB * nt);
      nl = r / nt;
      tap = r - nl * nt;
    } else {
      nl = e / per_n;
      const int r = e - nl * per_n;
      c = r / nt;
      tap = r - c * nt;
    }
    const PackUnit& pu = su[tap * 2 + (c >> 4)];
    const int n = n0 + nl, kk = c & 15;
    float v = 0.f;
    if (pu.wbase >= 0 && n < nreal && pu.c0 + kk < pu.creal) v = params[pu.wbase + (long long)n * pu.sn + (long long)(pu.c0 + kk) * pu.sc + pu.tap];
    pk_sv[nl * per_n + c * nt + tap] = v;  // consecutive lanes, consecutive words (a [unit][16] layout: one bank for the whole wave)
  }
  __syncthreads();
  for (int e = tid; e < ugw * PK_NB * 8; e += 256) {
    const int pr = e & 7, half = (e >> 3) & 1, nl = (e >> 4) % PK_NB, ksl = e / (16 * PK_NB);
    const int ul = ksl * 2 + half, u = u0 + ul, n = n0 + nl;
    if (u >= nunits || n >= Npad) continue;
    const float* v = pk_sv + nl * per_n + (half * 16 + 2 * pr) * nt + ksl;
    uint32_t h, l;
    split_pair(v[0], v[nt], h, l);
    const size_t d = (((size_t)(u >> 1) * Npad + n) * 32 + (u & 1) * 16) / 2 + pr;
    hi_img[d] = h;
    lo_img[d] = l;
  }
}

// bias image of a forward launch: b[n] = params[b0 + n] (+ params[b1 + n])
__global__ void pack_bias_kernel(const float* __restrict__ params, long long b0, long long b1, int nreal, int Npad, float* __restrict__ dst) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npad) return;
  float v = 0.f;
  if (n < nreal) v = params[b0 + n] + (b1 >= 0 ? params[b1 + n] : 0.f);
  dst[n] = v;
}

// head image [cout][2][cin] / [cout][2] from the two 1x1x1 weights
__global__ void pack_head_kernel(const float* __restrict__ params, long long wc, long long wr, long long bc, long long br, int cout, int cin,
                                 float* __restrict__ hw, float* __restrict__ hb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * cin) {
    const int o = i / cin, c = i - o * cin;
    hw[(o * 2 + 0) * cin + c] = params[wc + i];
    hw[(o * 2 + 1) * cin + c] = params[wr + i];
  }
  if (i < cout) {
    hb[i * 2 + 0] = params[bc + i];
    hb[i * 2 + 1] = params[br + i];
  }
}

// WeightedMSELoss, pass 1: sums[0] += sum of w (p - t)^2 over w > 0, sums[1] += count(w > 0), sums[2] += sum over all,
// sums[3] += count(scale != 0)
// `part` (deterministic mode): instead of the atomics every workgroup leaves its four sums in part[block][4] (its waves folded
// in wave order) and fold_kernel adds the workgroups in index order.
__global__ void loss_sums_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ w, size_t n,
                                 double* __restrict__ sums, double* __restrict__ part) {
  __shared__ double wave_sums[16][4];
  double s_mask = 0, s_all = 0;
  unsigned long long c_mask = 0, c_nz = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = p[i] - t[i];
    const float sc = w[i] * (d * d);
    s_all += sc;
    if (w[i] > 0.f) { s_mask += sc; ++c_mask; }
    if (sc != 0.f) ++c_nz;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s_mask += __shfl_down(s_mask, o);
    s_all += __shfl_down(s_all, o);
    c_mask += __shfl_down(c_mask, o);
    c_nz += __shfl_down(c_nz, o);
  }
  if (part) {
    if ((threadIdx.x & 63) == 0) {
      double* ws = wave_sums[threadIdx.x >> 6];
      ws[0] = s_mask; ws[1] = (double)c_mask; ws[2] = s_all; ws[3] = (double)c_nz;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double acc = 0;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) acc += wave_sums[wv][threadIdx.x];
      part[(size_t)blockIdx.x * 4 + threadIdx.x] = acc;
    }
    return;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sums[0], s_mask);
    atomicAdd(&sums[1], (double)c_mask);
    atomicAdd(&sums[2], s_all);
    atomicAdd(&sums[3], (double)c_nz);
  }
}

// Ordered fold of per-workgroup partial sums (deterministic mode): out[i] (+)= part[0][i] + part[1][i] + ... in that order, one
// thread per column i < width; rows are `stride` values apart.  The same bits whatever order the workgroups ran in.
template <typename T>
__global__ __launch_bounds__(1024) void fold_kernel(const T* __restrict__ part, int nparts, int stride, int width, T* __restrict__ out0,
                                                    T* __restrict__ out1, int assign) {
  // a workgroup = 32 columns x 32 chunk lanes: lane l adds rows l, l + 32, l + 64, ... in that order, then the 32 lanes' sums are
  // added in lane order -- a fixed tree, whatever the order the partial results were produced in
  __shared__ T lanes[32][33];
  const int col = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  T acc = 0;
  if (i < width) {
    int pidx = l;
    for (; pidx + 96 < nparts; pidx += 128) {  // four loads in flight, added in row order
      const T a0 = part[(size_t)pidx * stride + i], a1 = part[(size_t)(pidx + 32) * stride + i];
      const T a2 = part[(size_t)(pidx + 64) * stride + i], a3 = part[(size_t)(pidx + 96) * stride + i];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; pidx < nparts; pidx += 32) acc += part[(size_t)pidx * stride + i];
  }
  lanes[l][col] = acc;
  __syncthreads();
  if (l != 0 || i >= width) return;
  T sum = 0;
  for (int k = 0; k < 32; ++k) sum += lanes[k][col];
  if (assign) {
    out0[i] = sum;
    if (out1) out1[i] = sum;
  } else {
    out0[i] += sum;
    if (out1) out1[i] += sum;
  }
}

// pass 2: loss value and dL/dp; dp = 2 w (p - t) / N with N = count(w > 0) if any weighted error is non-zero, else numel
__global__ void loss_grad_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ w, size_t n,
                                 const double* __restrict__ sums, float* __restrict__ dp, float* __restrict__ loss_accum) {
  const bool masked = sums[3] != 0.0;
  const double denom = masked ? sums[1] : (double)n;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(loss_accum, (float)((masked ? sums[0] : sums[2]) / denom));
  const float inv = (float)(1.0 / denom);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = 2.f * w[i] * (p[i] - t[i]) * inv;
    dp[i] = (masked && !(w[i] > 0.f)) ? 0.f : g;
  }
}

// head backward: p = sigmoid((Wc + Wr) z + bc + br).  dlogit = dp p (1 - p); dz (channels-last) += (Wc + Wr)^T dlogit;
// dWc, dWr += dlogit z^T; dbc, dbr += dlogit.  One thread per voxel, block-level reduction of the weight gradients.
__global__ void head_bwd_kernel(const float* __restrict__ z, int zc, const float* __restrict__ p, const float* __restrict__ dp, size_t nvox,
                                int cin, int cout, const float* __restrict__ hw, float* __restrict__ dz, float* __restrict__ gwc,
                                float* __restrict__ gwr, float* __restrict__ gbc, float* __restrict__ gbr, float* __restrict__ part) {
  extern __shared__ float red[];  // [cout * cin + cout]; deterministic mode: one such row per wave
  const int nred = cout * cin + cout;
  for (int i = threadIdx.x; i < nred; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (part) {
    // deterministic mode: every product is summed over the wave by a fixed shuffle tree (lanes past the end hold zeros), the
    // waves in wave order, and the workgroup's row goes to part[block][nred] for fold_kernel: no atomics anywhere
    const bool live = v < nvox;
    const size_t vv = live ? v : 0;
    float* mine = red + (threadIdx.x >> 6) * nred;
    for (int c = 0; c < cin; ++c) {
      float acc = 0.f;
      for (int o = 0; o < cout; ++o) {
        const float pp = p[(size_t)o * nvox + vv];
        acc += (hw[(o * 2 + 0) * cin + c] + hw[(o * 2 + 1) * cin + c]) * (dp[(size_t)o * nvox + vv] * pp * (1.f - pp));
      }
      if (live) dz[v * zc + c] += acc;
    }
    for (int o = 0; o < cout; ++o) {
      const float pp = p[(size_t)o * nvox + vv];
      const float dlo = live ? dp[(size_t)o * nvox + vv] * pp * (1.f - pp) : 0.f;
      for (int c = 0; c <= cin; ++c) {  // c == cin: the bias column
        float t = c < cin ? dlo * z[vv * zc + c] : dlo;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
        if ((threadIdx.x & 63) == 0) mine[c < cin ? o * cin + c : cout * cin + o] = t;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += blockDim.x) {
      float acc = 0.f;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) acc += red[wv * nred + i];
      part[(size_t)blockIdx.x * nred + i] = acc;
    }
    return;
  }
  if (v < nvox) {
    float zz[32], dl[16];
    for (int c = 0; c < cin; ++c) zz[c] = z[v * zc + c];
    for (int o = 0; o < cout; ++o) {
      const float pp = p[(size_t)o * nvox + v];
      dl[o] = dp[(size_t)o * nvox + v] * pp * (1.f - pp);
    }
    for (int c = 0; c < cin; ++c) {
      float acc = 0.f;
      for (int o = 0; o < cout; ++o) acc += (hw[(o * 2 + 0) * cin + c] + hw[(o * 2 + 1) * cin + c]) * dl[o];
      dz[v * zc + c] += acc;
    }
    for (int o = 0; o < cout; ++o) {
      for (int c = 0; c < cin; ++c) atomicAdd(&red[o * cin + c], dl[o] * zz[c]);
      atomicAdd(&red[cout * cin + o], dl[o]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cout * cin; i += blockDim.x) {
    atomicAdd(&gwc[i], red[i]);
    atomicAdd(&gwr[i], red[i]);
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) {
    atomicAdd(&gbc[i], red[cout * cin + i]);
    atomicAdd(&gbr[i], red[cout * cin + i]);
  }
}

// g = dY * [Y > 0], written into the interior of a zero-bordered tensor [D + 2pz][H + 2py][W + 2px][C]
// `outs` (optional): the same tensor once more in the split-bf16 activation layout (conv_dev.h act_index: per 8 channels 16
// bytes of hi = bf16(v) then 16 bytes of lo = bf16(v - hi)), the A operand of the split-bf16 input-gradient launch
// `cs0` (optional): the column sums of g -- the bias gradient -- are added to cs0[c] (and cs1[c]) for c < nreal: per workgroup
// in LDS (dynamic, C floats), one global atomic per channel and workgroup at the end (colsum_kernel read the tensor again).
__global__ void relu_bwd_pad_kernel(const float* __restrict__ dy, const float* __restrict__ y, int D, int H, int W, int C, int pz, int py,
                                    int px, float* __restrict__ out, uint16_t* __restrict__ outs, int nreal, float* __restrict__ cs0,
                                    float* __restrict__ cs1) {
  extern __shared__ float rb_sum[];  // [C] when cs0
  if (cs0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) rb_sum[c] = 0.f;
    __syncthreads();
  }
  const size_t total = (size_t)D * H * W * (C / 4);
  const int Hp = H + 2 * py, Wp = W + 2 * px;
  // the channel group of a thread is fixed when the grid's stride is a multiple of C / 4 (the launcher sees to it): sums in registers
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool fixed_c = ((size_t)gridDim.x * blockDim.x) % (size_t)(C / 4) == 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    size_t v = i / (C / 4);
    const int x = (int)(v % W); v /= W;
    const int yy = (int)(v % H);
    const int zz = (int)(v / H);
    const size_t src = (((size_t)zz * H + yy) * W + x) * C + c4 * 4;
    const float4 g = *(const float4*)(dy + src), a = *(const float4*)(y + src);
    float4 r;
    r.x = a.x > 0.f ? g.x : 0.f; r.y = a.y > 0.f ? g.y : 0.f; r.z = a.z > 0.f ? g.z : 0.f; r.w = a.w > 0.f ? g.w : 0.f;
    const size_t row = (((size_t)(zz + pz) * Hp + (yy + py)) * Wp + (x + px)) * C;
    *(float4*)(out + row + c4 * 4) = r;
    if (cs0) {
      if (fixed_c) {
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
      } else {
        atomicAdd(&rb_sum[c4 * 4 + 0], r.x); atomicAdd(&rb_sum[c4 * 4 + 1], r.y);
        atomicAdd(&rb_sum[c4 * 4 + 2], r.z); atomicAdd(&rb_sum[c4 * 4 + 3], r.w);
      }
    }
    if (outs) {
      const int n = c4 * 4;
      uint32_t h0, l0, h1, l1;
      split_pair(r.x, r.y, h0, l0);
      split_pair(r.z, r.w, h1, l1);
      uint16_t* d = outs + 2 * row + ((n >> 3) << 4) + (n & 7);
      *(uint2*)d = make_uint2(h0, h1);
      *(uint2*)(d + 8) = make_uint2(l0, l1);
    }
  }
  if (cs0) {
    if (fixed_c) {
      const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
      if (i0 < total) {
        const int c4 = (int)(i0 % (C / 4));
        atomicAdd(&rb_sum[c4 * 4 + 0], acc.x); atomicAdd(&rb_sum[c4 * 4 + 1], acc.y);
        atomicAdd(&rb_sum[c4 * 4 + 2], acc.z); atomicAdd(&rb_sum[c4 * 4 + 3], acc.w);
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C && c < nreal; c += blockDim.x) {
      const float v = rb_sum[c];
      if (v != 0.f) {
        atomicAdd(&cs0[c], v);
        if (cs1) atomicAdd(&cs1[c], v);
      }
    }
  }
}

// split-bf16 copy of an f32 tensor (groups of 8 channels: 16 bytes of hi, 16 bytes of lo)
__global__ void f32_to_split_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, size_t ngroups8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = src[2 * i], b = src[2 * i + 1];
    uint4 h, l;
    split_pair(a.x, a.y, h.x, l.x);
    split_pair(a.z, a.w, h.y, l.y);
    split_pair(b.x, b.y, h.z, l.z);
    split_pair(b.z, b.w, h.w, l.w);
    dst[2 * i] = h;
    dst[2 * i + 1] = l;
  }
}

// f32 tensor out of a split-bf16 one (the split-bf16 input-gradient launch writes its result in the activation layout)
__global__ void split_to_f32_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, size_t ngroups8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups8; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 h = src[2 * i], l = src[2 * i + 1];
    float4 a, b;
    a.x = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
    a.y = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
    a.z = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
    a.w = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
    b.x = __uint_as_float(h.z << 16) + __uint_as_float(l.z << 16);
    b.y = __uint_as_float(h.z & 0xffff0000u) + __uint_as_float(l.z & 0xffff0000u);
    b.z = __uint_as_float(h.w << 16) + __uint_as_float(l.w << 16);
    b.w = __uint_as_float(h.w & 0xffff0000u) + __uint_as_float(l.w & 0xffff0000u);
    dst[2 * i] = a;
    dst[2 * i + 1] = b;
  }
}

// column sums of the interior of a padded tensor: out[c0 + c] += sum over voxels of g[..][c0 + c], c < Cc (Cc <= 1024)
__global__ void colsum_kernel(const float* __restrict__ g, int D, int H, int W, int C, int c0, int Cc, int pz, int py, int px, int nreal,
                              float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ part) {
  const int Hp = H + 2 * py, Wp = W + 2 * px;
  const int lanes = blockDim.x / Cc;  // voxels handled side by side
  if ((int)threadIdx.x >= lanes * Cc) return;  // (none: the launcher's block size is a multiple of Cc)
  const int c = c0 + (int)threadIdx.x % Cc;
  // one line of the interior per lane group and trip: no division per element (64-bit ones cost more than the load)
  const int nrows = D * H;
  float acc = 0.f;
  for (int row = (int)blockIdx.x * lanes + (int)threadIdx.x / Cc; row < nrows; row += (int)gridDim.x * lanes) {
    const int zz = row / H, yy = row - zz * H;
    const float* gl = g + (((size_t)(zz + pz) * Hp + (yy + py)) * Wp + px) * C + c;
    float a0 = 0.f, a1 = 0.f;
    int x = 0;
    for (; x + 1 < W; x += 2) {
      a0 += gl[(size_t)x * C];
      a1 += gl[(size_t)(x + 1) * C];
    }
    if (x < W) a0 += gl[(size_t)x * C];
    acc += a0 + a1;
  }
  if (part) {  // deterministic mode: the lane groups in index order, then part[block][C] for fold_kernel
    extern __shared__ float cs_red[];  // [lanes][Cc]
    cs_red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < Cc) {
      float sum = 0.f;
      for (int lg = 0; lg < lanes; ++lg) sum += cs_red[lg * Cc + threadIdx.x];
      part[(size_t)blockIdx.x * C + c] = sum;
    }
    return;
  }
  if (c < nreal && acc != 0.f) {
    atomicAdd(&out0[c], acc);
    if (out1) atomicAdd(&out1[c], acc);
  }
}

// Weight gradient of one kernel tap: dW[n][cbase + c][tap] += sum over output voxels m of g[m][n] x[m + tap][c].
// One wave per (32 x 32 block of (n, c), tap, chunk of output lines); v_mfma_f32_32x32x2_f32 contracts two voxels per
// instruction, and since every lane of that instruction supplies ONE element (row lane % 32, k = lane / 32) both operands
// are read straight from the channels-last tensors: 32 lanes = 128 contiguous bytes of one voxel.
struct WgradArgs {
  const float* g; long long gsz, gsy, gsx;  // interior of the padded gradient: origin pointer and strides (floats)
  const float* x; long long xsz, xsy, xsx;  // source tensor at (slot origin + tap origin) and strides (floats)
  int Do, Ho, Wo;
  int N, C;            // real output / input channels of this slot
  int kz, ky, kx;      // taps of this launch (1,1,1 for the residual)
  float* dw;           // gradient of the weight [N][Cin_total][ntap]
  float* dwt;          // split-bf16 form: tap-major workspace [ntap][N][Cin_total] (wgrad_finish_kernel adds it into dw)
  int cin_total, cbase, ntap;
  int lines_per_block;
};

// KX = kx taps of one (kz, ky) tap row are accumulated by the same wave: one g value and KX pairs of x values per
// voxel pair feed 2 * KX MFMAs (a 32 x 64 block of (n, c) per tap); 96 accumulator registers, so several waves share
// a SIMD and hide each other's load latency.
template <int KX>
__global__ __launch_bounds__(64) void wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
  const int nblocks_c = (a.C + 63) / 64;
  const int nb = blockIdx.x / nblocks_c, cb = blockIdx.x - nb * nblocks_c;
  const int trow = blockIdx.y;  // (tz, ty)
  const int tz = trow / a.ky, ty = trow - tz * a.ky;
  const int n0 = nb * 32 + lr, c0 = cb * 64 + lr;
  const bool nok0 = n0 < a.N, cok0 = c0 < a.C, cok1 = c0 + 32 < a.C;
  const float* gp = a.g + (nok0 ? n0 : 0);
  const float* xp = a.x + (cok0 ? c0 : 0) + tz * a.xsz + ty * a.xsy;
  f32x16_t acc[KX][2];
#pragma unroll
  for (int t = 0; t < KX; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  const int nlines = a.Do * a.Ho;
  const int l0 = blockIdx.z * a.lines_per_block, l1 = min(nlines, l0 + a.lines_per_block);
  for (int l = l0; l < l1; ++l) {
    const int z = l / a.Ho, y = l - z * a.Ho;
    const float* gl = gp + z * a.gsz + y * a.gsy;
    const float* xl = xp + z * a.xsz + y * a.xsy;
#pragma unroll 2
    for (int x0 = 0; x0 < a.Wo; x0 += 2) {
      const int xx = x0 + lh;
      const bool ok = xx < a.Wo;
      const float g0 = (ok && nok0) ? gl[xx * a.gsx] : 0.f;
      float x0v[KX], x1v[KX];
#pragma unroll
      for (int t = 0; t < KX; ++t) {
        x0v[t] = (ok && cok0) ? xl[(xx + t) * a.xsx] : 0.f;
        x1v[t] = (ok && cok1) ? xl[(xx + t) * a.xsx + 32] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < KX; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x0v[t], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x1v[t], acc[t][1], 0, 0, 0);
      }
    }
  }
  // acc[r]: row (n) = (r & 3) + 8 (r >> 2) + 4 lh, column (c) = lr
#pragma unroll
  for (int t = 0; t < KX; ++t) {
    const int tap = trow * KX + t;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cb * 64 + j * 32 + lr;
      if (c >= a.C) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nn < a.N && acc[t][j][r] != 0.f)
          atomicAdd(&a.dw[((size_t)nn * a.cin_total + a.cbase + c) * a.ntap + tap], acc[t][j][r]);
      }
    }
  }
}

// LDS-tiled form for the wide layers: a workgroup of 4 waves (2 x 2) owns a 128 x 128 block of (n, c) for the KX taps of
// one (kz, ky) tap row.  32 voxels of g ([32][128] floats) and the 32 + KX - 1 voxels of x they meet are staged in LDS
// once and shared by the four waves (a 3x smaller global read volume than the per-wave form, which is bound by it);
// the next chunk is fetched into registers while the current one is multiplied.
template <int KX>
__global__ __launch_bounds__(256) void wgrad_tiled_kernel(const WgradArgs a) {
  constexpr int MB = 32, TW = 128, XR = MB + KX - 1;
  __shared__ float gs[MB][TW];
  __shared__ float xs[XR][TW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  const int nblocks_c = (a.C + TW - 1) / TW;
  const int nt = blockIdx.x / nblocks_c, ct = blockIdx.x - nt * nblocks_c;
  const int trow = blockIdx.y;
  const int tz = trow / a.ky, ty = trow - tz * a.ky;
  const int nbase = nt * TW, cbase = ct * TW;
  // which of this wave's 2 x 2 blocks hold real channels (uniform)
  bool nuse[2], cuse[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    nuse[i] = nbase + wn * 64 + i * 32 < a.N;
    cuse[i] = cbase + wc * 64 + i * 32 < a.C;
  }
  f32x16_t acc[KX][2][2];
#pragma unroll
  for (int t = 0; t < KX; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
  // staging: thread -> (row, 4-float column group); g: MB rows x 32 groups = 1024 float4 = 4 per thread; x: XR rows
  constexpr int GV = MB * (TW / 4) / 256, XV = (XR * (TW / 4) + 255) / 256;
  float4 gr[GV], xr[XV];
  const float* gp0 = a.g;
  const float* xp0 = a.x + tz * a.xsz + ty * a.xsy;
  const int nlines = a.Do * a.Ho;
  const int l0 = blockIdx.z * a.lines_per_block, l1 = min(nlines, l0 + a.lines_per_block);
  const int chunks_per_line = (a.Wo + MB - 1) / MB;
  const int nchunks = (l1 - l0) * chunks_per_line;
  auto fetch = [&](int ch) {
    const int l = l0 + ch / chunks_per_line, x0 = (ch % chunks_per_line) * MB;
    const int z = l / a.Ho, y = l - z * a.Ho;
    const float* gl = gp0 + z * a.gsz + y * a.gsy;
    const float* xl = xp0 + z * a.xsz + y * a.xsy;
#pragma unroll
    for (int v = 0; v < GV; ++v) {
      const int idx = tid + v * 256, row = idx / (TW / 4), c4 = (idx % (TW / 4)) * 4;
      const int xx = x0 + row, n = nbase + c4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xx < a.Wo && n < a.N) {  // channel counts are padded to 16, so a 4-group never straddles the tensor's row end
        val = *(const float4*)(gl + (long long)xx * a.gsx + n);
        if (n + 3 >= a.N) {
          if (n + 1 >= a.N) val.y = 0.f;
          if (n + 2 >= a.N) val.z = 0.f;
          val.w = 0.f;
        }
      }
      gr[v] = val;
    }
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int idx = tid + v * 256, row = idx / (TW / 4), c4 = (idx % (TW / 4)) * 4;
      const int xx = x0 + row, c = cbase + c4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < XR && xx < a.Wo + KX - 1 && c < a.C) {
        val = *(const float4*)(xl + (long long)xx * a.xsx + c);
        if (c + 3 >= a.C) {
          if (c + 1 >= a.C) val.y = 0.f;
          if (c + 2 >= a.C) val.z = 0.f;
          val.w = 0.f;
        }
      }
      xr[v] = val;
    }
  };
  if (nchunks > 0) fetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();  // the previous chunk has been multiplied
#pragma unroll
    for (int v = 0; v < GV; ++v) {
      const int idx = tid + v * 256;
      *(float4*)&gs[idx / (TW / 4)][(idx % (TW / 4)) * 4] = gr[v];
    }
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int idx = tid + v * 256;
      if (idx / (TW / 4) < XR) *(float4*)&xs[idx / (TW / 4)][(idx % (TW / 4)) * 4] = xr[v];
    }
    __syncthreads();
    if (ch + 1 < nchunks) fetch(ch + 1);
    const int x0 = (ch % chunks_per_line) * MB;
    const int mvalid = min(MB, a.Wo - x0);  // rows beyond hold zeros in gs (fetch), so they add nothing
#pragma unroll 4
    for (int m = 0; m < MB; m += 2) {
      if (m >= mvalid) break;
      float g2[2], x2[KX][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) g2[i] = gs[m + lh][wn * 64 + i * 32 + lr];
#pragma unroll
      for (int t = 0; t < KX; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) x2[t][j] = xs[m + lh + t][wc * 64 + j * 32 + lr];
#pragma unroll
      for (int t = 0; t < KX; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (nuse[i] && cuse[j]) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g2[i], x2[t][j], acc[t][i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < KX; ++t) {
    const int tap = trow * KX + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = cbase + wc * 64 + j * 32 + lr;
        if (!nuse[i] || c >= a.C) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nn = nbase + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (nn < a.N && acc[t][i][j][r] != 0.f) atomicAdd(&a.dw[((size_t)nn * a.cin_total + a.cbase + c) * a.ntap + tap], acc[t][i][j][r]);
        }
      }
  }
}

// ---- split-bf16 weight gradient -------------------------------------------------------------------------------------
// The same sums on the bf16 matrix pipe (16x the rate of v_mfma_f32_32x32x2_f32): every f32 operand is split into
// hi = bf16(v), lo = bf16(v - hi) and the product is hi*hi + lo*hi + hi*lo with f32 accumulation (what BSMI_PREC_BF16X3
// does in the forward pass; relative error ~2^-17 per product).  v_mfma_f32_16x16x32_bf16 contracts 32 voxels per
// instruction and wants 8 consecutive K values (voxels) of one row (channel) per lane -- the transpose of the channels-last
// tensors.  So the operands are PACKED first (wgrad_pack_kernel, one elementwise pass per operand and conv stage):
//   G[group][plane][channel][8]   group = 8 consecutive output voxels of a line (the last group of a line zero-filled),
//                                 plane = hi | lo, channels padded to the tile; one more all-zero group at the end
//   X[group][plane][vec][channel][8]   the 8 input voxels under the group and, in vec 1, the next 8 (KX > 1): the operand
//                                 of kernel tap t is the group's vector shifted by t values
// so that 64 channels of one (group, plane) are 1 KiB of contiguous memory = ONE LDS-DMA instruction, and the fragments
// are plain 16-byte LDS reads.  (The first version split the f32 tensors inside the kernel: ~10 VALU instructions per
// element loaded left the MFMA pipe idle 80 % of the time.)
// A chunk = 4 groups = the K = 32 of one MFMA (19-voxel lines fill 79 % of the slots; whole-line chunks would fill 59 %).
// A workgroup of 2 x 2 waves owns a (32 FNW) x (32 FCW) block of (n, c) for the KX taps of one (kz, ky) tap row and a
// range of output lines; wave w stages group w of every chunk; two LDS buffers: chunk ch + 1 lands while ch is multiplied.
// src: f32 tensor at its first (line, voxel) with strides in floats; lines = nz x ny lines of `width` valid voxels, `gpl`
// groups per line; dst[((group * 2 + plane) * nvec + vec) * cpad + c] = 16-byte vector of voxels 8 (xg + vec) .. + 7 of
// channel c (zeros past `width` and past `creal`); `nullg` more all-zero groups follow.
__global__ void wgrad_pack_kernel(const float* __restrict__ src, long long sz, long long sy, long long sx, int nz, int ny, int width, int creal,
                                  int cpad, int gpl, int nvec, int nullg, u32x4_t* __restrict__ dst) {
  // 32-bit index arithmetic (the host checks the item count): 64-bit divisions cost more than the rest of the body
  const uint32_t ngroups = (uint32_t)nz * ny * gpl;
  const uint32_t total = (ngroups + nullg) * cpad;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t grp = i / (uint32_t)cpad;
    const int c = (int)(i - grp * cpad);
    const uint32_t line = grp / (uint32_t)gpl;
    const int xg = (int)(grp - line * gpl);
    const int z = (int)(line / (uint32_t)ny), y = (int)(line - (uint32_t)z * ny);
    const bool ok = grp < ngroups && c < creal;
    const float* sp = src + (ok ? (long long)z * sz + (long long)y * sy + c : 0);
    for (int v = 0; v < nvec; ++v) {
      u32x4_t hi, lo;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int x0 = (xg + v) * 8 + 2 * d;
        const float f0 = (ok && x0 < width) ? sp[x0 * sx] : 0.f;
        const float f1 = (ok && x0 + 1 < width) ? sp[(x0 + 1) * sx] : 0.f;
        uint32_t h, l;
        split_pair(f0, f1, h, l);
        hi[d] = h;
        lo[d] = l;
      }
      dst[(((size_t)grp * 2 + 0) * nvec + v) * cpad + c] = hi;
      dst[(((size_t)grp * 2 + 1) * nvec + v) * cpad + c] = lo;
    }
  }
}

struct WgradPk {
  const char* gp;  // packed g: [ngroups + 1][2][Np][16 B]
  const char* xp;  // packed x: [input lines * gpl][2][XVEC][Cp][16 B]
  int Np, Cp, gpl;
  int Do, Ho, Hil;  // output lines Do x Ho; input lines per z: Ho + ky - 1
  int N, C;         // real channels
  int kz, ky;
  float* dwt;       // tap-major workspace [ntap][N][cin_total]
  int cin_total, cbase, ntap;
  int lines_per_block, zsplit;
  size_t zstride;   // deterministic mode: line range z adds into its own copy of the workspace, dwt + z * zstride (0: one copy)
};

template <int KX, int FNW, int FCW>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(const WgradPk a) {
  // (two workgroups per CU: at most 96 accumulator registers per lane.  The 128 x 128 tile's 192 did not fit beside the
  // operands: the compiler shuttled fragments through AGPRs, 350 copies per chunk, and one wave per SIMD hid nothing)
  constexpr int TN = 32 * FNW, TC = 32 * FCW;
  static_assert(KX * FNW * FCW * 4 <= 96, "accumulators");
  constexpr int XVEC = KX > 1 ? 2 : 1;
  constexpr int GBYTES = 2 * 4 * TN * 16;          // [plane][group][TN][16 B]
  constexpr int XBYTES = 2 * XVEC * 4 * TC * 16;   // [plane][vec][group][TC][16 B]
  constexpr int BUF = GBYTES + XBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // two buffers
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;
  // Workgroup -> (tile, line range, tap row), XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, so the
  // kz * ky tap rows of one (tile, line range) are made consecutive ON one XCD: they run together and share the tile's
  // g vectors and (shifted by a line or two) x vectors in that XCD's L2 (id-major order sent the 9 readers of the same
  // vectors to different XCDs at different times: 3.9 TB/s of L2 misses on the 1500 -> 1500 layer).
  const int nblocks_c = (a.C + TC - 1) / TC;
  const int ntiles = ((a.N + TN - 1) / TN) * nblocks_c;
  const int trows = a.kz * a.ky;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int trow = seq % trows;
  const int unit = (seq / trows) * 8 + xcd;
  if (unit >= ntiles * a.zsplit) return;  // (uniform; the grid is padded to 8 x trows)
  const int tile = unit % ntiles, zblk = unit / ntiles;
  // deterministic mode: this line range's own copy of the workspace -- one contributor per element, wgrad_finish_kernel adds the
  // copies in z order (the atomicAdd below then adds to a zero and is exact whatever the order of the workgroups)
  float* const dwt_z = a.dwt + (size_t)zblk * a.zstride;
  const int nt = tile / nblocks_c, ct = tile - nt * nblocks_c;
  const int tz = trow / a.ky, ty = trow - tz * a.ky;
  const int nbase = nt * TN, cbase = ct * TC;
  bool nuse[FNW], cuse[FCW];
#pragma unroll
  for (int i = 0; i < FNW; ++i) nuse[i] = nbase + (wn * FNW + i) * 16 < a.N;
#pragma unroll
  for (int j = 0; j < FCW; ++j) cuse[j] = cbase + (wc * FCW + j) * 16 < a.C;
  const bool nall = nuse[FNW - 1];
  f32x4_t acc[KX][FNW][FCW];
#pragma unroll
  for (int t = 0; t < KX; ++t)
#pragma unroll
    for (int i = 0; i < FNW; ++i)
#pragma unroll
      for (int j = 0; j < FCW; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int gpl = a.gpl;
  const int nlines = a.Do * a.Ho;
  const int l0 = zblk * a.lines_per_block, l1 = min(nlines, l0 + a.lines_per_block);
  const int ngroups = (l1 - l0) * gpl;
  const int nchunks = (ngroups + 3) >> 2;
  // this wave's group of the chunk being staged: index within the block's range, output line (z, y), group of the line
  int gi = wave, pz, py, pxg;
  {
    const int line = l0 + wave / gpl;
    pxg = wave % gpl;
    pz = line / a.Ho;
    py = line - pz * a.Ho;
  }
  const size_t gvec = (size_t)a.Np * 16, xvec = (size_t)a.Cp * 16;  // bytes of one (group, plane[, vec]) row of all channels
  const wg_gptr_t gsrc = (wg_gptr_t)a.gp + (size_t)(nbase + lane) * 16;
  const wg_gptr_t xsrc = (wg_gptr_t)a.xp + (size_t)(cbase + lane) * 16;
  const size_t gnull = (size_t)nlines * gpl;  // the all-zero group
  auto stage = [&](int buf) __attribute__((always_inline)) {
    const bool ok = gi < ngroups;
    const size_t gabs = ok ? (size_t)l0 * gpl + gi : gnull;
    const size_t xabs = ok ? ((size_t)(pz + tz) * a.Hil + (py + ty)) * gpl + pxg : 0;  // past the range: any group (g is zero)
    const wg_lptr_t lg = (wg_lptr_t)(smem + buf * BUF);
    const wg_lptr_t lx = (wg_lptr_t)(smem + buf * BUF + GBYTES);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      // one instruction = 64 channels of a (group, plane); a 32-channel tile uses the lower half of the lanes
#pragma unroll
      for (int hf = 0; hf < (TN + 63) / 64; ++hf)
        if (TN >= 64 || lane < TN)
          __builtin_amdgcn_global_load_lds(gsrc + (gabs * 2 + pl) * gvec + hf * 1024, lg + ((pl * 4 + wave) * TN + hf * 64) * 16, 16, 0, 0);
#pragma unroll
      for (int v = 0; v < XVEC; ++v)
#pragma unroll
        for (int hf = 0; hf < (TC + 63) / 64; ++hf)
          if (TC >= 64 || lane < TC)
            __builtin_amdgcn_global_load_lds(xsrc + ((xabs * 2 + pl) * XVEC + v) * xvec + hf * 1024,
                                             lx + (((pl * XVEC + v) * 4 + wave) * TC + hf * 64) * 16, 16, 0, 0);
    }
    // four groups on
    gi += 4;
    pxg += 4;
    while (pxg >= gpl) {
      pxg -= gpl;
      if (++py == a.Ho) { py = 0; ++pz; }
    }
  };
  auto mfma = [](f32x4_t c, u32x4_t x, u32x4_t y) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, y), c, 0, 0, 0);
  };
  const uint32_t aoff = (uint32_t)((lq * TN + wn * FNW * 16 + lr) * 16);
  const uint32_t boff = (uint32_t)((lq * TC + wc * FCW * 16 + lr) * 16);
  if (nchunks > 0) stage(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's group of chunk ch has landed
    __syncthreads();                                  // ... everybody's has, and chunk ch - 1 has been multiplied
    if (ch + 1 < nchunks) stage(buf ^ 1);
    const char* gs = smem + buf * BUF;
    const char* xs = gs + GBYTES;
    u32x4_t ah[FNW], al[FNW];
#pragma unroll
    for (int i = 0; i < FNW; ++i) {
      ah[i] = *(const u32x4_t*)(gs + aoff + i * 256);
      al[i] = *(const u32x4_t*)(gs + 4 * TN * 16 + aoff + i * 256);
    }
#pragma unroll
    for (int j = 0; j < FCW; ++j) {
      if (!cuse[j]) continue;
      uint32_t d[2][5];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const char* xb = xs + pl * XVEC * 4 * TC * 16 + boff + j * 256;
        const u32x4_t q = *(const u32x4_t*)xb;
        d[pl][0] = q.x; d[pl][1] = q.y; d[pl][2] = q.z; d[pl][3] = q.w;
        d[pl][4] = XVEC > 1 ? *(const uint32_t*)(xb + 4 * TC * 16) : 0u;
      }
      u32x4_t bh[KX], bl[KX];
#pragma unroll
      for (int t = 0; t < KX; ++t) {
        u32x4_t qh, ql;
        if (t == 0) {
          qh = u32x4_t{d[0][0], d[0][1], d[0][2], d[0][3]};
          ql = u32x4_t{d[1][0], d[1][1], d[1][2], d[1][3]};
        } else if (t == 1) {
          qh = u32x4_t{__builtin_amdgcn_alignbyte(d[0][1], d[0][0], 2), __builtin_amdgcn_alignbyte(d[0][2], d[0][1], 2),
                       __builtin_amdgcn_alignbyte(d[0][3], d[0][2], 2), __builtin_amdgcn_alignbyte(d[0][4], d[0][3], 2)};
          ql = u32x4_t{__builtin_amdgcn_alignbyte(d[1][1], d[1][0], 2), __builtin_amdgcn_alignbyte(d[1][2], d[1][1], 2),
                       __builtin_amdgcn_alignbyte(d[1][3], d[1][2], 2), __builtin_amdgcn_alignbyte(d[1][4], d[1][3], 2)};
        } else {
          qh = u32x4_t{d[0][1], d[0][2], d[0][3], d[0][4]};
          ql = u32x4_t{d[1][1], d[1][2], d[1][3], d[1][4]};
        }
        bh[t] = qh;
        bl[t] = ql;
      }
      // three products per accumulator, the accumulators of a product back to back (independent instructions)
      // (a wave whose n fragments are all real -- every wave but those of a layer's last tile -- runs them without a branch)
      if (nall) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int t = 0; t < KX; ++t)
#pragma unroll
            for (int i = 0; i < FNW; ++i) acc[t][i][j] = mfma(acc[t][i][j], pr == 1 ? al[i] : ah[i], pr == 2 ? bl[t] : bh[t]);
      } else {
#pragma unroll
        for (int i = 0; i < FNW; ++i) {
          if (!nuse[i]) continue;
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int t = 0; t < KX; ++t) acc[t][i][j] = mfma(acc[t][i][j], pr == 1 ? al[i] : ah[i], pr == 2 ? bl[t] : bh[t]);
        }
      }
    }
  }
  // acc[r]: row (n) = 4 (lane >> 4) + r, column (c) = lane & 15
#pragma unroll
  for (int t = 0; t < KX; ++t) {
    const int tap = trow * KX + t;
#pragma unroll
    for (int i = 0; i < FNW; ++i)
#pragma unroll
      for (int j = 0; j < FCW; ++j) {
        const int c = cbase + (wc * FCW + j) * 16 + lr;
        if (!nuse[i] || !cuse[j] || c >= a.C) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nn = nbase + (wn * FNW + i) * 16 + 4 * lq + r;
          // tap-major: the 16 lanes of a row are 64 contiguous bytes (in the OIDHW gradient they are 4 ntap bytes apart,
          // one cache line per lane)
          if (nn < a.N && acc[t][i][j][r] != 0.f) atomicAdd(&dwt_z[((size_t)tap * a.N + nn) * a.cin_total + a.cbase + c], acc[t][i][j][r]);
        }
      }
  }
}

// dw[n][c][tap] += dwt[tap][n][c]; dwt = 0 (ready for the next step).  A block takes 256 consecutive (n, c): the tap planes
// are read coalesced over (n, c), transposed through LDS and added into the OIDHW gradient as one contiguous run.
// nz, zstride (deterministic mode): the workspace is nz copies, one per line range of the launches; they are added in z order.
__global__ __launch_bounds__(256) void wgrad_finish_kernel(float* __restrict__ dwt, float* __restrict__ dw, size_t nc, int ntap, int nz,
                                                           size_t zstride) {
  __shared__ float tile[256 * 28];
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int n = (int)min((size_t)256, nc - i0);
  const int tid = threadIdx.x;
  auto take = [&](int t) -> float {  // the sum over the copies, each left zero
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) {
      float* q = dwt + (size_t)z * zstride + (size_t)t * nc + i0 + tid;
      const float v = *q;
      if (v != 0.f) { acc += v; *q = 0.f; }
    }
    return acc;
  };
  if (ntap > 27) {  // (no such kernel in the model family; plain form)
    if (tid < n)
      for (int t = 0; t < ntap; ++t) {
        const float v = take(t);
        if (v != 0.f) dw[(i0 + tid) * ntap + t] += v;
      }
    return;
  }
  if (tid < n)
    for (int t = 0; t < ntap; ++t) tile[tid * 28 + t] = take(t);
  __syncthreads();
  const int total = n * ntap;
  float* d = dw + i0 * ntap;
  for (int k = tid; k < total; k += 256) {
    const int i = k / ntap, t = k - i * ntap;
    const float v = tile[i * 28 + t];
    if (v != 0.f) d[k] += v;
  }
}

// The same for the deterministic mode's nz copies: 1024 threads = 256 (n, c) x 4 tap groups (a narrow layer is ONE workgroup here
// and its launches were cut into many line ranges: 27 taps x nz dependent loads per thread took milliseconds); four copies are in
// flight at a time and are added in z order.
constexpr int kDetMaxRanges = 32;  // line ranges per launch in the deterministic mode (launch_wgrad_x3_t)
__global__ __launch_bounds__(1024) void wgrad_finish_det_kernel(float* __restrict__ dwt, float* __restrict__ dw, size_t nc, int ntap, int nz,
                                                                size_t zstride) {
  __shared__ float tile[256 * 28];
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int n = (int)min((size_t)256, nc - i0);
  const int tid = threadIdx.x & 255, tg = threadIdx.x >> 8;
  if (tid < n)
    for (int t = tg; t < ntap; t += 4) {
      float* q = dwt + (size_t)t * nc + i0 + tid;
      float acc = 0.f;
      int z = 0;
      for (; z + 3 < nz; z += 4) {
        float* q0 = q + (size_t)z * zstride;
        const float v0 = q0[0], v1 = q0[zstride], v2 = q0[2 * zstride], v3 = q0[3 * zstride];
        acc += v0; acc += v1; acc += v2; acc += v3;
        if (v0 != 0.f) q0[0] = 0.f;
        if (v1 != 0.f) q0[zstride] = 0.f;
        if (v2 != 0.f) q0[2 * zstride] = 0.f;
        if (v3 != 0.f) q0[3 * zstride] = 0.f;
      }
      for (; z < nz; ++z) {
        float* q0 = q + (size_t)z * zstride;
        const float v = *q0;
        acc += v;
        if (v != 0.f) *q0 = 0.f;
      }
      tile[tid * 28 + t] = acc;
    }
  __syncthreads();
  const int total = n * ntap;
  float* d = dw + i0 * ntap;
  for (int k = threadIdx.x; k < total; k += 1024) {
    const int i = k / ntap, t = k - i * ntap;
    const float v = tile[i * 28 + t];
    if (v != 0.f) d[k] += v;
  }
}

// Which convolutions of the step run as split-bf16 launches: all of them under bsmi_unet_train_set_arithmetic(h, 1)
// (the default); BSMI_WGRAD_X3 / BSMI_DGRAD_X3 / BSMI_FWD_X3 = 0 take single ones back to f32 (dev knobs, read at begin).
static bool env_on(const char* name) {
  const char* e = getenv(name);
  return !e || e[0] != '0';
}
static bool wgrad_x3_enabled(const bsmi_unet* h) { return h->train_split && env_on("BSMI_WGRAD_X3"); }
static bool dgrad_x3_enabled(const bsmi_unet* h) { return h->train_split && env_on("BSMI_DGRAD_X3") && two_waves_per_simd(); }
static bool fwd_x3_enabled(const bsmi_unet* h) { return h->train_split && env_on("BSMI_FWD_X3") && two_waves_per_simd(); }

// tile widths the launcher picks (= channel padding of the packed operands): 32 / 64 / 128 output channels x 32 / 64
// input channels
static int wgrad_tile_n(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : 128); }
static int wgrad_tile_c(int c) { return c <= 32 ? 32 : 64; }
static int wgrad_pad(int channels, int tile) { return (channels + tile - 1) / tile * tile; }

template <int KX, int FNW, int FCW>
static int launch_wgrad_x3_t(WgradPk a, hipStream_t s) {
  constexpr int TN = 32 * FNW, TC = 32 * FCW, XVEC = KX > 1 ? 2 : 1;
  constexpr int smem = 2 * (2 * 4 * TN * 16 + 2 * XVEC * 4 * TC * 16);
  static DeviceOnce once;
  const int rc_once = once.run([&]() -> int {
    BSMI_HIP(hipFuncSetAttribute((const void*)wgrad_x3_kernel<KX, FNW, FCW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    return BSMI_OK;
  });
  if (rc_once) return rc_once;
  const int nlines = a.Do * a.Ho, trows = a.kz * a.ky;
  const int blocks_nc = ((a.N + TN - 1) / TN) * ((a.C + TC - 1) / TC);
  int zsplit = std::max(1, std::min(nlines, 4096 / std::max(1, blocks_nc * trows)));
  if (a.zstride) zsplit = std::min(zsplit, kDetMaxRanges);
  const int zcap = a.zsplit;  // deterministic mode: the copies of the workspace the caller has room for
  a.lines_per_block = (nlines + zsplit - 1) / zsplit;
  a.zsplit = (nlines + a.lines_per_block - 1) / a.lines_per_block;
  if (a.zstride && a.zsplit > zcap) BSMI_FAIL(BSMI_ERR_STATE, "weight-gradient launch cut into %d line ranges, workspace for %d", a.zsplit, zcap);
  const int units = blocks_nc * a.zsplit;
  hipLaunchKernelGGL((wgrad_x3_kernel<KX, FNW, FCW>), dim3((units + 7) / 8 * 8 * trows), dim3(256), smem, s, a);
  return BSMI_OK;
}

template <int KX>
static int launch_wgrad_x3_k(const WgradPk& a, hipStream_t s) {
  const int fn = wgrad_tile_n(a.N) / 32, fc = wgrad_tile_c(a.C) / 32;
  if (fn == 1 && fc == 1) return launch_wgrad_x3_t<KX, 1, 1>(a, s);
  if (fn == 1 && fc == 2) return launch_wgrad_x3_t<KX, 1, 2>(a, s);
  if (fn == 2 && fc == 1) return launch_wgrad_x3_t<KX, 2, 1>(a, s);
  if (fn == 2 && fc == 2) return launch_wgrad_x3_t<KX, 2, 2>(a, s);
  if (fn == 4 && fc == 1) return launch_wgrad_x3_t<KX, 4, 1>(a, s);
  return launch_wgrad_x3_t<KX, 4, 2>(a, s);
}

// dst[region at (oz, oy, ox)][cdst + c] += src[..][csrc + c] for c < C (gradient of crop + concat)
__global__ void scatter_add_kernel(const float* __restrict__ src, int D, int H, int W, int Cs, int csrc, float* __restrict__ dst, int Hd, int Wd,
                                   int Cd, int cdst, int oz, int oy, int ox, int C) {
  const size_t total = (size_t)D * H * W * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t v = i / C;
    const int x = (int)(v % W); v /= W;
    const int y = (int)(v % H);
    const int z = (int)(v / H);
    dst[(((size_t)(z + oz) * Hd + (y + oy)) * Wd + (x + ox)) * Cd + cdst + c] += src[(((size_t)z * H + y) * W + x) * Cs + csrc + c];
  }
}

// max-pool backward: the gradient of a window goes to its first maximum (torch: strict > while scanning z, y, x)
__global__ void maxpool_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ din, int H, int W, int C,
                                   int Do, int Ho, int Wo, int fz, int fy, int fx) {
  const size_t total = (size_t)Do * Ho * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t v = i / C;
    const int x = (int)(v % Wo); v /= Wo;
    const int y = (int)(v % Ho);
    const int z = (int)(v / Ho);
    float best = -INFINITY;
    size_t arg = 0;
    for (int dz = 0; dz < fz; ++dz)
      for (int dy = 0; dy < fy; ++dy)
        for (int dx = 0; dx < fx; ++dx) {
          const size_t s = ((size_t)((z * fz + dz) * H + (y * fy + dy)) * W + (x * fx + dx)) * C + c;
          const float val = in[s];
          if (val > best || (dz == 0 && dy == 0 && dx == 0)) { best = val; arg = s; }
        }
    din[arg] += dout[i];
  }
}

// trilinear upsample (align_corners = False, integer factors) + crop, backward: every output voxel adds its gradient to
// the up-to-8 input voxels it interpolated from (float atomics: windows of neighbouring outputs overlap)
__global__ void upsample_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int Di, int Hi, int Wi, int C, int Do, int Ho, int Wo,
                                    int fz, int fy, int fx, int oz, int oy, int ox) {
  const size_t total = (size_t)Do * Ho * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t v = i / C;
    const int x = (int)(v % Wo) + ox; v /= Wo;
    const int y = (int)(v % Ho) + oy;
    const int z = (int)(v / Ho) + oz;
    const float g = dout[i];
    if (g == 0.f) continue;
    int i0[3], i1[3];
    float w1[3];
    const int pos[3] = {z, y, x}, f[3] = {fz, fy, fx}, n[3] = {Di, Hi, Wi};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float s = ((float)pos[d] + 0.5f) / (float)f[d] - 0.5f;  // torch area_pixel_compute_source_index, align_corners = False
      s = s < 0.f ? 0.f : s;
      i0[d] = (int)s;
      i1[d] = i0[d] + (i0[d] < n[d] - 1 ? 1 : 0);
      w1[d] = s - (float)i0[d];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int iz = (k & 4) ? i1[0] : i0[0], iy = (k & 2) ? i1[1] : i0[1], ix = (k & 1) ? i1[2] : i0[2];
      const float wt = ((k & 4) ? w1[0] : 1.f - w1[0]) * ((k & 2) ? w1[1] : 1.f - w1[1]) * ((k & 1) ? w1[2] : 1.f - w1[2]);
      if (wt != 0.f) atomicAdd(&din[(((size_t)iz * Hi + iy) * Wi + ix) * C + c], g * wt);
    }
  }
}

// The same as a gather (deterministic mode): one thread per INPUT voxel and channel walks the outputs that interpolated from it, in
// z, y, x order, and adds their shares in that order -- no atomics.  Along one axis input j is the lower neighbour (i0) of the
// outputs whose source coordinate lies in [j, j + 1) and the upper one (i1) of those in [j - 1, j): 2 f candidates.
__global__ void upsample_bwd_gather_kernel(const float* __restrict__ dout, float* __restrict__ din, int Di, int Hi, int Wi, int C, int Do, int Ho,
                                           int Wo, int fz, int fy, int fx, int oz, int oy, int ox) {
  const size_t total = (size_t)Di * Hi * Wi * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t v = i / C;
    const int jx = (int)(v % Wi); v /= Wi;
    const int jy = (int)(v % Hi);
    const int jz = (int)(v / Hi);
    // share of output position p (upsampled coordinate) that input j gets along one axis: (1 - w1) if i0 == j, + w1 if i1 == j
    auto share = [](int p, int f, int n, int j, float& lo, float& hi) {
      float s = ((float)p + 0.5f) / (float)f - 0.5f;
      s = s < 0.f ? 0.f : s;
      const int i0 = (int)s, i1 = i0 + (i0 < n - 1 ? 1 : 0);
      const float w1 = s - (float)i0;
      lo = i0 == j ? 1.f - w1 : 0.f;
      hi = i1 == j ? w1 : 0.f;
    };
    float acc = 0.f;
    const int pz0 = max(oz, jz * fz - fz), pz1 = min(oz + Do, jz * fz + 2 * fz);
    const int py0 = max(oy, jy * fy - fy), py1 = min(oy + Ho, jy * fy + 2 * fy);
    const int px0 = max(ox, jx * fx - fx), px1 = min(ox + Wo, jx * fx + 2 * fx);
    for (int pz = pz0; pz < pz1; ++pz) {
      float zl, zh;
      share(pz, fz, Di, jz, zl, zh);
      if (zl == 0.f && zh == 0.f) continue;
      for (int py = py0; py < py1; ++py) {
        float yl, yh;
        share(py, fy, Hi, jy, yl, yh);
        if (yl == 0.f && yh == 0.f) continue;
        for (int px = px0; px < px1; ++px) {
          float xl, xh;
          share(px, fx, Wi, jx, xl, xh);
          if (xl == 0.f && xh == 0.f) continue;
          const float g = dout[(((size_t)(pz - oz) * Ho + (py - oy)) * Wo + (px - ox)) * C + c];
          if (g == 0.f) continue;
          // the eight products of the scatter form, in its order (k = 0 .. 7: z bit 4, y bit 2, x bit 1), those that land on j
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float wt = ((k & 4) ? zh : zl) * ((k & 2) ? yh : yl) * ((k & 1) ? xh : xl);
            if (wt != 0.f) acc += g * wt;
          }
        }
      }
    }
    if (acc != 0.f) din[i] += acc;
  }
}

// torch.optim.Adam (no weight decay, no amsgrad) on flat buffers; gscale folds the 1 / world_size of a summed all-reduce
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                            float beta1, float beta2, float eps, float bc1, float bc2_sqrt, float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = w[i] - (lr / bc1) * (mi / denom);
  }
}

// ------------------------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------------------------
struct ParamRef {
  std::string key;
  size_t off = 0, count = 0;
  std::vector<int64_t> shape;
};

struct PackJob {  // one packed weight image that must follow the parameters
  PackUnit* units = nullptr;  // device
  int nunits = 0, Npad = 0, nreal = 0;
  float* dst = nullptr;
  uint32_t *dst_hi = nullptr, *dst_lo = nullptr;  // fused split-bf16 image instead (units of 16 channels)
  int window = 2;  // units of one 32-channel chunk (taps x 2): the thread order of pack_weights_x3_kernel
  bool shadowed = false;  // f32 image of a forward launch that trains in its split-bf16 form (make_forward_x3)
  bool backward = false;  // image of an input-gradient launch: first read in the backward pass (packed on the side stream)
  bool late = false;      // forward image of a wide stage, first read a millisecond into the forward pass (side stream as well)
  // bias image (forward launches only)
  long long b0 = -1, b1 = -1;
  float* bias_dst = nullptr;
};

struct ConvBwd {  // backward data of one CONV plan step
  const PlanStep* st = nullptr;
  int P[3] = {0, 0, 0};  // border of the padded gradient
  TDesc gp;              // padded gradient [D + 2P][H + 2P][W + 2P][Cpad]
  void* gps = nullptr;   // the same in the split-bf16 activation layout (input gradients as split-bf16 launches), or null
  void* dsplit = nullptr;  // result of the split-bf16 input-gradient launch before split_to_f32_kernel (null: the launch writes f32 sums itself)
  bool dx3 = false;        // the input gradient is a split-bf16 launch
  bool need_dgrad = false;
  ConvArgs dgrad{};      // implicit-GEMM launch of the input gradient
  TileCfg dtile = TILE_256x32;
  TDesc dcat;            // its output when the pass input is a crop / concat (stage 0), else the previous stage's gradient
  bool scatter = false;
  hipEvent_t ev_g = nullptr;  // the padded gradient is written: the weight-gradient stream may start on this stage
};

// (PlanStep::tx3's type, unet_internal.h)
struct TrainFwdX3 {  // a forward CONV step as a fused split-bf16 launch
  ConvArgs a{};
  TileCfg tile = TILE_256x32;
  int nconv_src = 0;
  const void* src_f32[kMaxConvTensors] = {nullptr, nullptr, nullptr};  // sources to split before the launch (null: a split copy exists)
  void* src_split[kMaxConvTensors] = {nullptr, nullptr, nullptr};
  size_t src_g8[kMaxConvTensors] = {0, 0, 0};
  void* out_split = nullptr;  // the launch's result, turned into the step's f32 output tensor afterwards
  size_t out_g8 = 0;
  bool late = false;          // its weight image is packed on the side stream after an optimizer step (TrainState::ev_fwd_packed)
};

struct TrainState {
  std::vector<std::unique_ptr<TrainFwdX3>> fwd_x3;
  std::map<const void*, void*> split_of;  // f32 activation -> its split copy written by an earlier launch of the forward pass
  std::map<const void*, size_t> fwd_job_of;  // bias image of a forward launch -> index of its pack job
  bool f32_images_stale = false;
  int64_t in_shape[3] = {0, 0, 0};
  Plan* plan = nullptr;
  std::vector<ParamRef> params;
  std::map<std::string, size_t> index;
  size_t nparams = 0;
  float *w = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;
  float* gt = nullptr;  // tap-major workspace of the split-bf16 weight gradients (same offsets as g; zero between steps)
  char *pk_g = nullptr, *pk_x = nullptr;  // packed operands of the split-bf16 weight gradient (grown on first use)
  size_t pk_g_bytes = 0, pk_x_bytes = 0;
  // deterministic mode (bsmi_unet_train_set_deterministic): per-line-range copies of one weight tensor's tap-major workspace
  // (zero between launches, like gt), per-workgroup partial sums of the bias / head / loss reductions
  char *gt_det = nullptr, *det_part = nullptr;
  size_t gt_det_bytes = 0, det_part_bytes = 0;
  double* loss_part = nullptr;       // [512][4]
  hipStream_t wstream = nullptr;     // the weight gradients' own stream (null: BSMI_TRAIN_WSTREAM=0, everything on the caller's)
  bool own_wstream = false;
  hipEvent_t ev_join = nullptr;      // its last launch of a backward pass
  hipEvent_t ev_adam = nullptr, ev_packed = nullptr;  // optimizer step done / input-gradient images repacked on the side stream
  hipEvent_t ev_fwd_packed = nullptr;                  // the wide stages' forward images repacked there
  bool packed_pending = false, fwd_packed_pending = false;
  bool wstream_wanted = false;                         // BSMI_TRAIN_WSTREAM at begin (known before the stream itself is made)
  int adam_t = 0;
  // Gradient groups: the parameters of one ConvPass / head are one contiguous range of the flat buffers (their keys share
  // a prefix and the buffers follow the sorted keys); a group's gradients are final once the backward pass has left its
  // first stage.  group_order: groups in the order the backward pass finishes them; an event per group is recorded on
  // the backward stream so that a data-parallel caller can start reducing a group while the pass goes on.
  struct GradGroup { std::string prefix; size_t off = 0, count = 0; hipEvent_t ev = nullptr; };
  std::vector<GradGroup> groups;       // completion order
  std::map<std::string, int> group_of; // prefix -> index into groups
  std::map<void*, TDesc> grad_of;
  std::vector<std::pair<void*, size_t>> zero_list;  // gradient tensors cleared at the start of every backward pass
  std::vector<void*> allocs;
  std::vector<PackJob> jobs;
  std::vector<ConvBwd> convs;        // indexed like plan->steps (empty entries for other step types)
  float* zero_bias = nullptr;        // [2048] zeros: bias operand of the dgrad launches
  double* loss_sums = nullptr;       // [4] per head, reused
  float* loss_dev = nullptr;         // [1]
  std::vector<float*> head_out;      // per head: sigmoid outputs [C][D][H][W] of the last forward
  std::vector<float*> head_dp;       // per head: dL/dp
  size_t out_vox = 0;
};

static int talloc(TrainState* ts, void** p, size_t bytes, bool zero) {
  BSMI_HIP(hipMalloc(p, bytes + 256));
  ts->allocs.push_back(*p);
  if (zero) BSMI_HIP(hipMemset(*p, 0, bytes + 256));
  return BSMI_OK;
}

void free_train_state(bsmi_unet* h) {
  if (!h->train) return;
  if (h->train->plan)
    for (PlanStep& st : h->train->plan->steps) st.tx3 = nullptr;
  for (auto& g : h->train->groups)
    if (g.ev) (void)hipEventDestroy(g.ev);
  for (auto& cb : h->train->convs)
    if (cb.ev_g) (void)hipEventDestroy(cb.ev_g);
  if (h->train->ev_join) (void)hipEventDestroy(h->train->ev_join);
  if (h->train->ev_adam) (void)hipEventDestroy(h->train->ev_adam);
  if (h->train->ev_packed) (void)hipEventDestroy(h->train->ev_packed);
  if (h->train->ev_fwd_packed) (void)hipEventDestroy(h->train->ev_fwd_packed);
  // (wstream is the device's side stream, shared by every training state of the process: not destroyed here)
  if (h->train->own_wstream && h->train->wstream) (void)hipStreamDestroy(h->train->wstream);
  for (void* p : h->train->allocs) (void)hipFree(p);
  if (h->train->pk_g) (void)hipFree(h->train->pk_g);
  if (h->train->pk_x) (void)hipFree(h->train->pk_x);
  if (h->train->gt_det) (void)hipFree(h->train->gt_det);
  if (h->train->det_part) (void)hipFree(h->train->det_part);
  delete h->train;
  h->train = nullptr;
}

static size_t param_off(TrainState* ts, const std::string& key) {
  auto it = ts->index.find(key);
  return it == ts->index.end() ? (size_t)-1 : ts->params[it->second].off;
}

// gradient tensor of an activation (same geometry, f32), created on first use
static int grad_tensor(TrainState* ts, const TDesc& act, TDesc* out) {
  auto it = ts->grad_of.find(act.ptr);
  if (it != ts->grad_of.end()) {
    *out = it->second;
    return BSMI_OK;
  }
  TDesc g = act;
  const size_t bytes = (size_t)act.D * act.H * act.W * act.Cpad * sizeof(float);
  const size_t slack = (size_t)8 * act.W * act.Cpad * sizeof(float) + 4096;
  int rc = talloc(ts, &g.ptr, bytes + slack, true);
  if (rc) return rc;
  ts->grad_of[act.ptr] = g;
  ts->zero_list.push_back({g.ptr, bytes});
  *out = g;
  return BSMI_OK;
}

static int upload_units(TrainState* ts, const std::vector<PackUnit>& u, PackUnit** dev) {
  int rc = talloc(ts, (void**)dev, u.size() * sizeof(PackUnit), false);
  if (rc) return rc;
  BSMI_HIP(hipMemcpy(*dev, u.data(), u.size() * sizeof(PackUnit), hipMemcpyHostToDevice));
  return BSMI_OK;
}

// pack job of a forward launch: the unit list is the one the planner packed from (build_entries)
static int make_forward_job(bsmi_unet* h, TrainState* ts, PassSite& p, int ci) {
  PackedConv& pc = p.packed[BSMI_PREC_F32][ci];
  const bool last = ci == p.nconv - 1;
  const std::string base = p.prefix + ".conv_pass." + std::to_string(2 * ci);
  const size_t wm = param_off(ts, base + ".weight"), bm = param_off(ts, base + ".bias");
  const size_t wr = param_off(ts, p.prefix + ".residual.0.weight"), br = param_off(ts, p.prefix + ".residual.0.bias");
  const HostWeight& hm = h->weights[base + ".weight"];
  const HostWeight& hr = h->weights[p.prefix + ".residual.0.weight"];
  const int64_t cin_m = hm.shape[1], ntap = hm.shape[2] * hm.shape[3] * hm.shape[4], cin_r = hr.shape[1];
  std::vector<PackUnit> units(pc.entries.size());
  for (size_t u = 0; u < pc.entries.size(); ++u) {
    const PackEntry& e = pc.entries[u];
    PackUnit pu{};
    if (e.dummy) {
      pu.wbase = -1;
    } else if (e.wsrc == 0) {
      pu.wbase = (long long)wm + (long long)e.cin_base * ntap;
      pu.sn = (int)(cin_m * ntap); pu.sc = (int)ntap; pu.tap = e.tap;
    } else {
      pu.wbase = (long long)wr + e.cin_base;
      pu.sn = (int)cin_r; pu.sc = 1; pu.tap = 0;
    }
    pu.c0 = e.c0;
    pu.creal = e.creal;
    units[u] = pu;
  }
  PackJob job;
  int rc = upload_units(ts, units, &job.units);
  if (rc) return rc;
  job.nunits = (int)units.size();
  job.Npad = pc.Npad;
  job.nreal = p.cout;
  job.dst = (float*)pc.w;
  job.b0 = (long long)bm;
  job.b1 = last ? (long long)br : -1;
  job.bias_dst = pc.bias;
  ts->fwd_job_of[pc.bias] = ts->jobs.size();
  ts->jobs.push_back(job);
  return BSMI_OK;
}

// `lazy_f32`: leave out the f32 weight image of a forward launch that runs in its split-bf16 form during training
// (PackJob::shadowed); the images are then stale until train_refresh_f32_images, which an f32 inference call on the same
// handle triggers (unet_api.hip) -- the bias images and everything the step itself reads are always current.
// which: 0 every image, 1 those the forward pass reads first, 2 those only the backward pass reads, 3 the wide stages' forward images
static int run_pack_jobs(TrainState* ts, hipStream_t s, bool lazy_f32 = false, bool only_shadowed = false, int which = 0) {
  for (const PackJob& j : ts->jobs) {
    if (only_shadowed && !j.shadowed) continue;
    if (which == 1 && (j.backward || j.late)) continue;   // early forward images
    if (which == 2 && !j.backward) continue;              // input-gradient images
    if (which == 3 && !j.late) continue;                  // late forward images
    const size_t total = (size_t)j.nunits * j.Npad;
    if (j.dst_hi) {
      const int ugw = std::max(2, j.window);
      static const bool pack_t = env_on("BSMI_PACK_T");
      if (ugw >= 16 && ugw % 2 == 0 && j.Npad % PK_NB == 0 && pack_t) {
        const size_t lds = (size_t)PK_NB * ugw * 16 * sizeof(float) + (size_t)ugw * sizeof(PackUnit);
        const dim3 grid((unsigned)((j.nunits + ugw - 1) / ugw), (unsigned)(j.Npad / PK_NB));
        if (ugw == 54)
          hipLaunchKernelGGL(pack_weights_x3_t_kernel<27>, grid, dim3(256), lds, s, (const float*)ts->w, (const PackUnit*)j.units, j.nunits, j.Npad,
                             j.nreal, ugw, j.dst_hi, j.dst_lo);
        else
          hipLaunchKernelGGL(pack_weights_x3_t_kernel<0>, grid, dim3(256), lds, s, (const float*)ts->w, (const PackUnit*)j.units, j.nunits, j.Npad,
                             j.nreal, ugw, j.dst_hi, j.dst_lo);
      } else {
        const size_t padded = (size_t)((j.nunits + ugw - 1) / ugw) * ugw * j.Npad;
        hipLaunchKernelGGL(pack_weights_x3_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, s, (const float*)ts->w,
                           (const PackUnit*)j.units, j.nunits, j.Npad, j.nreal, ugw, j.dst_hi, j.dst_lo);
      }
    } else if (!(lazy_f32 && j.shadowed)) {
      hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)ts->w, (const PackUnit*)j.units,
                         j.nunits, j.Npad, j.nreal, j.dst);
    }
    if (j.bias_dst && !only_shadowed)
      hipLaunchKernelGGL(pack_bias_kernel, dim3((j.Npad + 255) / 256), dim3(256), 0, s, (const float*)ts->w, j.b0, j.b1, j.nreal, j.Npad, j.bias_dst);
  }
  if (lazy_f32) ts->f32_images_stale = true;
  if (only_shadowed) ts->f32_images_stale = false;
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// input-gradient launch of conv stage `ci` of pass p (see the file header).  For ci >= 1 the output is the gradient
// of the previous stage's activation; for ci == 0 it is `dcat`, the gradient of the (cropped, concatenated) pass input.
// The forward launch of a gather-form CONV step once more as a fused split-bf16 launch (conv_igemm.hip conv_x3_body): the
// same unit list at 16 channels per unit, K-steps over split copies of the sources (4 bytes per channel like the f32
// tensors: the strides and offsets are the f32 plan's), hi / lo weight images repacked from the parameters every step,
// the bias image of the f32 launch.  The result lands in a split tensor -- the next convolution's source as it is -- and
// is converted to the step's f32 output, which everything else (pooling, upsampling, the backward pass) reads.
static int make_forward_x3(bsmi_unet* h, TrainState* ts, PlanStep& st) {
  if (st.use_box) return BSMI_OK;  // (forms of the other precisions; an f32 raster-halo step has its gather form in st.conv too)
  PassSite& p = *st.site;
  const int ci = st.ci;
  const PackedConv& pf = p.packed[BSMI_PREC_F32][ci];
  std::vector<PackEntry> ents;
  build_entries(p, ci, BSMI_PREC_BF16X3, ents);
  const bool last = ci == p.nconv - 1;
  (void)last;
  const std::string base = p.prefix + ".conv_pass." + std::to_string(2 * ci);
  const size_t wm = param_off(ts, base + ".weight"), wr = param_off(ts, p.prefix + ".residual.0.weight");
  const HostWeight& hm = h->weights[base + ".weight"];
  const HostWeight& hr = h->weights[p.prefix + ".residual.0.weight"];
  const int64_t cin_m = hm.shape[1], ntap = hm.shape[2] * hm.shape[3] * hm.shape[4], cin_r = hr.shape[1];
  std::vector<PackUnit> units(ents.size());
  for (size_t u = 0; u < ents.size(); ++u) {
    const PackEntry& e = ents[u];
    PackUnit pu{};
    if (e.dummy) {
      pu.wbase = -1;
    } else if (e.wsrc == 0) {
      pu.wbase = (long long)wm + (long long)e.cin_base * ntap;
      pu.sn = (int)(cin_m * ntap); pu.sc = (int)ntap; pu.tap = e.tap;
    } else {
      pu.wbase = (long long)wr + e.cin_base;
      pu.sn = (int)cin_r; pu.sc = 1; pu.tap = 0;
    }
    pu.c0 = e.c0;
    pu.creal = e.creal;
    units[u] = pu;
  }
  const size_t nsteps = ents.size() / kUnitsPerStep;
  std::vector<KStep> ks(nsteps);
  const int64_t es = 4;
  for (size_t s = 0; s < nsteps; ++s) {
    const int slot = ents[kUnitsPerStep * s].slot;
    const TDesc& t = st.slots[slot];
    KStep k;
    memset(&k, 0, sizeof k);
    k.tensor = slot;
    for (int j = 0; j < kUnitsPerStep; ++j) {
      const PackEntry& e = ents[kUnitsPerStep * s + j];
      if (e.dummy) continue;
      const int64_t off = ((((int64_t)(e.dz + st.so[slot][0]) * t.H) + (e.dy + st.so[slot][1])) * t.W + (e.dx + st.so[slot][2])) * t.Cpad + e.c0;
      k.delta[j] = (int32_t)(off * es);
    }
    ks[s] = k;
  }
  std::unique_ptr<TrainFwdX3> fx(new TrainFwdX3());
  fx->tile = pf.tile;
  int rc;
  const size_t wimg = (nsteps * (size_t)pf.Npad + kWeightRowSlack) * kStepRowBytes;
  char* wdev = nullptr;
  if ((rc = talloc(ts, (void**)&wdev, 2 * wimg, true))) return rc;
  KStep* dks = nullptr;
  if ((rc = talloc(ts, (void**)&dks, ks.size() * sizeof(KStep), false))) return rc;
  BSMI_HIP(hipMemcpy(dks, ks.data(), ks.size() * sizeof(KStep), hipMemcpyHostToDevice));
  PackJob job;
  if ((rc = upload_units(ts, units, &job.units))) return rc;
  job.nunits = (int)units.size();
  job.Npad = pf.Npad;
  job.nreal = p.cout;
  job.dst = (float*)wdev;
  job.dst_hi = (uint32_t*)wdev;
  job.dst_lo = (uint32_t*)(wdev + wimg);
  job.window = kUnitsPerStep * (int)ntap;
  // the wide stages' images (2.4 M weights and more: 0.5 of the 0.6 ms of forward packing) are not read before the forward pass
  // has done its first, narrow stages: packed on the side stream, the forward pass waits for them where it first needs one
  job.late = ts->wstream_wanted && (size_t)job.nunits * job.Npad * 16 >= ((size_t)2 << 20);
  fx->late = job.late;
  ts->jobs.push_back(job);
  ConvArgs& a = fx->a;
  memset(&a, 0, sizeof a);
  for (int sl = 0; sl < kMaxConvTensors; ++sl) {
    const int q = sl < st.nsl ? sl : 0;
    const TDesc& t = st.slots[q];
    void* sp = nullptr;
    if (sl < st.nsl) {
      auto it = ts->split_of.find(t.ptr);
      const size_t bytes = (size_t)t.D * t.H * t.W * t.Cpad * sizeof(float);
      if (it != ts->split_of.end()) {
        sp = it->second;
      } else {
        const size_t slack = (size_t)8 * t.W * t.Cpad * sizeof(float) + 4096;
        if ((rc = talloc(ts, &sp, bytes + slack, true))) return rc;
        fx->src_f32[sl] = t.ptr;
        fx->src_g8[sl] = bytes / 32;
        ts->split_of[t.ptr] = sp;  // a later launch of the same source (the residual's) finds it split already
      }
      fx->src_split[sl] = sp;
    } else {
      sp = fx->src_split[0];
    }
    a.t[sl].base = (uint64_t)(uintptr_t)sp;
    a.t[sl].sz = (int32_t)((int64_t)t.H * t.W * t.Cpad * es);
    a.t[sl].sy = (int32_t)((int64_t)t.W * t.Cpad * es);
    a.t[sl].sx = (int32_t)((int64_t)t.Cpad * es);
  }
  fx->nconv_src = st.nsl;
  const size_t obytes = (size_t)st.out.D * st.out.H * st.out.W * st.out.Cpad * sizeof(float);
  const size_t oslack = (size_t)8 * st.out.W * st.out.Cpad * sizeof(float) + 4096;
  if ((rc = talloc(ts, &fx->out_split, obytes + oslack, true))) return rc;
  fx->out_g8 = obytes / 32;
  ts->split_of[st.out.ptr] = fx->out_split;
  a.steps = dks;
  a.nsteps = (int)ks.size();
  a.w = wdev;
  a.w_lo = wdev + wimg;
  a.bias = pf.bias;
  a.out = fx->out_split;
  a.Do = st.out.D; a.Ho = st.out.H; a.Wo = st.out.W; a.Co = st.out.Cpad;
  a.M = st.out.D * st.out.H * st.out.W;
  a.Npad = pf.Npad;
  a.relu = 1;
  st.tx3 = fx.get();
  ts->fwd_x3.push_back(std::move(fx));
  auto fj = ts->fwd_job_of.find(pf.bias);
  if (fj != ts->fwd_job_of.end()) ts->jobs[fj->second].shadowed = true;
  return BSMI_OK;
}

int train_refresh_f32_images(bsmi_unet* h, hipStream_t s) {
  if (!h->train || !h->train->f32_images_stale) return BSMI_OK;
  return run_pack_jobs(h->train, s, false, true);
}

int train_forward_conv_x3(bsmi_unet* h, const PlanStep& st, hipStream_t s) {
  const TrainFwdX3& fx = *st.tx3;
  if (fx.late && h->train && h->train->fwd_packed_pending) {  // the first wide stage after an optimizer step: its image comes from the side stream
    BSMI_HIP(hipStreamWaitEvent(s, h->train->ev_fwd_packed, 0));
    h->train->fwd_packed_pending = false;
  }
  for (int sl = 0; sl < fx.nconv_src; ++sl)
    if (fx.src_f32[sl])
      hipLaunchKernelGGL(f32_to_split_kernel, dim3((unsigned)std::min<size_t>((fx.src_g8[sl] + 255) / 256, 16384)), dim3(256), 0, s,
                         (const float4*)fx.src_f32[sl], (uint4*)fx.src_split[sl], fx.src_g8[sl]);
  const int rc = launch_conv_igemm(fx.a, BSMI_PREC_BF16X3, fx.tile, s, h->sk_ws, h->sk_grid);
  if (rc) return rc;
  hipLaunchKernelGGL(split_to_f32_kernel, dim3((unsigned)std::min<size_t>((fx.out_g8 + 255) / 256, 16384)), dim3(256), 0, s,
                     (const uint4*)fx.out_split, (float4*)st.out.ptr, fx.out_g8);
  return BSMI_OK;
}

static int make_dgrad(bsmi_unet* h, TrainState* ts, ConvBwd& cb, const ConvBwd* last_cb) {
  const PlanStep& st = *cb.st;
  PassSite& p = *st.site;
  const int ci = st.ci, n = p.nconv;
  const int* k = p.k[ci];
  const int ntap = k[0] * k[1] * k[2];
  const int cin_total = ci == 0 ? p.cin[0] + (p.nslots > 1 ? p.cin[1] : 0) : p.cout;
  const bool x3 = cb.gps != nullptr;  // split-bf16 launch: units of 16 channels, K-steps of 32
  const int SUB = x3 ? 16 : 8;
  const std::string base = p.prefix + ".conv_pass." + std::to_string(2 * ci);
  const size_t wm = param_off(ts, base + ".weight"), wr = param_off(ts, p.prefix + ".residual.0.weight");
  int crop[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) crop[d] += p.k[i][d] - 1;
  const bool with_res = ci == 0 && n > 1;  // the residual 1x1x1 reads the pass input, whose gradient this launch produces
  const int cpad_g = cb.gp.Cpad;

  std::vector<PackUnit> units;
  std::vector<KStep> steps;
  auto close_step = [&]() {
    while (units.size() % kUnitsPerStep) {
      PackUnit pu{};
      pu.wbase = -1;
      units.push_back(pu);
    }
  };
  // source 0: this stage's padded gradient, all taps; the weight tap is the mirrored one
  const int64_t es = 4;
  const TDesc& g0 = cb.gp;
  for (int c16 = 0; c16 < cpad_g; c16 += kUnitsPerStep * SUB)
    for (int z = 0; z < k[0]; ++z)
      for (int y = 0; y < k[1]; ++y)
        for (int x = 0; x < k[2]; ++x) {
          KStep ks{};
          ks.tensor = 0;
          int j = 0;
          for (int c0 = c16; c0 < std::min(cpad_g, c16 + kUnitsPerStep * SUB); c0 += SUB, ++j) {
            PackUnit pu{};
            pu.wbase = (long long)wm;
            pu.sn = ntap;                       // n of the launch = input channel of the weight
            pu.sc = cin_total * ntap;           // K channel = output channel of the weight
            pu.tap = ((k[0] - 1 - z) * k[1] + (k[1] - 1 - y)) * k[2] + (k[2] - 1 - x);
            pu.c0 = c0;
            pu.creal = p.cout;
            units.push_back(pu);
            const int oz = cb.P[0] - (k[0] - 1) + z, oy = cb.P[1] - (k[1] - 1) + y, ox = cb.P[2] - (k[2] - 1) + x;
            ks.delta[j] = (int32_t)(((((int64_t)oz * g0.H + oy) * g0.W + ox) * g0.Cpad + c0) * es);
          }
          close_step();
          steps.push_back(ks);
        }
  if (with_res) {
    const TDesc& gl = last_cb->gp;
    for (int c16 = 0; c16 < gl.Cpad; c16 += kUnitsPerStep * SUB) {
      KStep ks{};
      ks.tensor = 1;
      int j = 0;
      for (int c0 = c16; c0 < std::min(gl.Cpad, c16 + kUnitsPerStep * SUB); c0 += SUB, ++j) {
        PackUnit pu{};
        pu.wbase = (long long)wr;
        pu.sn = 1;
        pu.sc = cin_total;
        pu.tap = 0;
        pu.c0 = c0;
        pu.creal = p.cout;
        units.push_back(pu);
        const int oz = last_cb->P[0] - crop[0] / 2, oy = last_cb->P[1] - crop[1] / 2, ox = last_cb->P[2] - crop[2] / 2;
        ks.delta[j] = (int32_t)(((((int64_t)oz * gl.H + oy) * gl.W + ox) * gl.Cpad + c0) * es);
      }
      close_step();
      steps.push_back(ks);
    }
  }
  if (steps.size() % 2) {  // even number of K-steps (conv_igemm.hip)
    for (int j = 0; j < kUnitsPerStep; ++j) {
      PackUnit pu{};
      pu.wbase = -1;
      units.push_back(pu);
    }
    steps.push_back(KStep{});
  }
  cb.dtile = choose_tile(cin_total);
  const int Npad = round_up(cin_total, tile_bn(cb.dtile));
  // output tensor
  TDesc out;
  out.C = cin_total;
  out.Cpad = round_up(cin_total, kChanPad);
  out.D = st.out.D + k[0] - 1;
  out.H = st.out.H + k[1] - 1;
  out.W = st.out.W + k[2] - 1;
  int rc;
  if (ci == 0) {
    const size_t bytes = (size_t)out.D * out.H * out.W * out.Cpad * sizeof(float);
    rc = talloc(ts, &out.ptr, bytes, true);
    if (rc) return rc;
    cb.scatter = true;
  } else {
    // previous stage's activation is slot 0 of this launch
    rc = grad_tensor(ts, st.slots[0], &out);
    if (rc) return rc;
  }
  cb.dcat = out;
  // packed weights + K-steps on the device
  float* wdev = nullptr;  // f32: rows of 16 floats; split-bf16: rows of 32 bf16, hi image then lo image (the same 64 bytes per row)
  const size_t wimg = (steps.size() * (size_t)Npad + kWeightRowSlack) * 16 * sizeof(float);
  rc = talloc(ts, (void**)&wdev, wimg * (x3 ? 2 : 1), true);
  if (rc) return rc;
  KStep* dks = nullptr;
  rc = talloc(ts, (void**)&dks, steps.size() * sizeof(KStep), false);
  if (rc) return rc;
  BSMI_HIP(hipMemcpy(dks, steps.data(), steps.size() * sizeof(KStep), hipMemcpyHostToDevice));
  PackJob job;
  rc = upload_units(ts, units, &job.units);
  if (rc) return rc;
  job.nunits = (int)units.size();
  job.Npad = Npad;
  job.nreal = cin_total;
  job.dst = wdev;
  if (x3) {
    job.dst_hi = (uint32_t*)wdev;
    job.dst_lo = (uint32_t*)((char*)wdev + wimg);
    job.window = kUnitsPerStep * ntap;
  }
  job.backward = true;
  ts->jobs.push_back(job);
  if (Npad > 2048) BSMI_FAIL(BSMI_ERR_INVALID, "dgrad launch wider than the zero-bias buffer");
  if (x3 && with_res && !last_cb->gps) BSMI_FAIL(BSMI_ERR_STATE, "training plan: the residual source has no split copy");
  ConvArgs& a = cb.dgrad;
  memset(&a, 0, sizeof a);
  const TDesc* srcs[kMaxConvTensors] = {&g0, with_res ? &last_cb->gp : &g0, &g0};
  const void* sptr[kMaxConvTensors] = {cb.gps, with_res ? last_cb->gps : cb.gps, cb.gps};
  for (int sl = 0; sl < kMaxConvTensors; ++sl) {
    const TDesc& t = *srcs[sl];
    a.t[sl].base = (uint64_t)(uintptr_t)(x3 ? sptr[sl] : t.ptr);  // (the split layout keeps 4 bytes per channel: same strides)
    a.t[sl].sz = (int32_t)((int64_t)t.H * t.W * t.Cpad * es);
    a.t[sl].sy = (int32_t)((int64_t)t.W * t.Cpad * es);
    a.t[sl].sx = (int32_t)((int64_t)t.Cpad * es);
  }
  a.steps = dks;
  a.nsteps = (int)steps.size();
  a.w = wdev;
  a.bias = ts->zero_bias;
  a.out = out.ptr;
  if (x3) {
    a.w_lo = (const char*)wdev + wimg;
    cb.dx3 = true;
    // no bias, no ReLU, and the reader wants f32: the launch stores its raw sums (ConvArgs::raw, the epilogue of the
    // Winograd GEMMs) straight into the gradient tensor instead of (hi, lo) pairs that split_to_f32_kernel took apart again
    static const bool raw_out = env_on("BSMI_DGRAD_RAW");
    if (raw_out && out.Cpad % 4 == 0) {
      a.raw = 1;
    } else {
      const size_t obytes = (size_t)out.D * out.H * out.W * out.Cpad * sizeof(float);
      rc = talloc(ts, &cb.dsplit, obytes, true);
      if (rc) return rc;
      a.out = cb.dsplit;
    }
  }
  a.Do = out.D; a.Ho = out.H; a.Wo = out.W; a.Co = out.Cpad;
  a.M = out.D * out.H * out.W;
  a.Npad = Npad;
  a.relu = 0;
  cb.need_dgrad = true;
  (void)h;
  return BSMI_OK;
}

}  // namespace bsmi

using namespace bsmi;

namespace bsmi {

// ---- affinity training targets (GrowBoundary -> AddAffinities -> BalanceLabels) --------------------------
constexpr int kMaxNeighborhood = 16;
struct Neighborhood {
  int n;
  int off[kMaxNeighborhood][3];
};

// out[p] = labels[p] if every voxel within L1 distance `steps` of p (same section if only_xy) has p's label,
// is unknown (unl == 0) or lies outside the block; else 0.  `steps` erosions with the 6- (4-) neighbour cross =
// one erosion with that L1 ball.  An unknown voxel belongs to every label's mask (custom_grow_boundary.py:96-100):
// it survives if the known voxels of its ball carry at most one label.
__global__ void grow_boundary_kernel(const int64_t* __restrict__ labels, const uint8_t* __restrict__ unl, int64_t* __restrict__ out,
                                     int D, int H, int W, int steps, int only_xy) {
  const size_t nvox = (size_t)D * H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < nvox; p += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / ((size_t)W * H));
    const int64_t mine = labels[p];
    const bool known = !unl || unl[p];
    int64_t want = known ? mine : -1;  // -1: any one label
    bool keep = !(known && mine == 0);
    const int rz = only_xy ? 0 : steps;
    for (int dz = -rz; dz <= rz && keep; ++dz) {
      const int zz = z + dz;
      if (zz < 0 || zz >= D) continue;
      const int ry = steps - abs(dz);
      for (int dy = -ry; dy <= ry && keep; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const int rx = ry - abs(dy);
        for (int dx = -rx; dx <= rx; ++dx) {
          const int xx = x + dx;
          if (xx < 0 || xx >= W) continue;
          const size_t q = ((size_t)zz * H + yy) * W + xx;
          if (unl && !unl[q]) continue;
          const int64_t l = labels[q];
          if (want == -1) want = l;
          if (l != want || l == 0) { keep = false; break; }
        }
      }
    }
    out[p] = keep ? mine : 0;
  }
}

__global__ void affinity_targets_kernel(const int64_t* __restrict__ labels, const uint8_t* __restrict__ unl, Neighborhood nb, int D, int H,
                                        int W, float* __restrict__ affs, float* __restrict__ mask, unsigned long long* __restrict__ counts) {
  const size_t nvox = (size_t)D * H * W;
  unsigned long long n_mask = 0, n_pos = 0;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < nvox; p += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / ((size_t)W * H));
    const int64_t a = labels[p];
    const bool known = !unl || unl[p];
    for (int e = 0; e < nb.n; ++e) {
      const int zz = z + nb.off[e][0], yy = y + nb.off[e][1], xx = x + nb.off[e][2];
      const bool inside = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
      float aff = 0.f, m = 0.f;
      if (inside) {
        const int64_t b = labels[((size_t)zz * H + yy) * W + xx];
        aff = (a == b && a > 0) ? 1.f : 0.f;
        m = known ? 1.f : 0.f;
      }
      affs[(size_t)e * nvox + p] = aff;
      mask[(size_t)e * nvox + p] = m;
      n_mask += m > 0.f;
      n_pos += (m > 0.f && aff > 0.f);
    }
  }
  // wave reduction, then one atomic pair per wave
  for (int o = 32; o > 0; o >>= 1) {
    n_mask += __shfl_down(n_mask, o);
    n_pos += __shfl_down(n_pos, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], n_mask);
    atomicAdd(&counts[1], n_pos);
  }
}

__global__ void balance_kernel(const float* __restrict__ affs, float* __restrict__ weights, size_t total,
                               const unsigned long long* __restrict__ counts, float clip_min, float clip_max) {
  const float n_mask = fmaxf((float)counts[0], 1.f);
  float frac = (float)counts[1] / n_mask;
  frac = fminf(fmaxf(frac, clip_min), clip_max);
  const float w_pos = 1.f / (2.f * frac), w_neg = 1.f / (2.f * (1.f - frac));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    weights[i] = weights[i] * (affs[i] > 0.f ? w_pos : w_neg);
}

}  // namespace bsmi

extern "C" {

int bsmi_unet_train_set_arithmetic(bsmi_unet* h, int split_bf16) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  if (h->train) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_set_arithmetic: call before bsmi_unet_train_begin");
  h->train_split = split_bf16 ? 1 : 0;
  return BSMI_OK;
}

int bsmi_unet_train_set_deterministic(bsmi_unet* h, int on) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  h->train_det = on ? 1 : 0;
  return BSMI_OK;
}

// a scratch buffer of the training state that only grows (first steps); zero_new: a grown buffer starts as zeros
static int grow_buf(hipStream_t s, char** buf, size_t* have, size_t need, bool zero_new) {
  if (need <= *have) return BSMI_OK;
  BSMI_HIP(hipStreamSynchronize(s));
  if (*buf) BSMI_HIP(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  BSMI_HIP(hipMalloc((void**)buf, need + 4096));
  // on the stream that uses the buffer: hipMemset runs on the null stream, which a non-blocking stream (the weight gradients' own)
  // does not wait for -- a launch could add into the buffer before the fill had passed (seen: two "deterministic" runs 2 974
  // gradient values apart, once in four test-suite runs)
  if (zero_new) BSMI_HIP(hipMemsetAsync(*buf, 0, need + 4096, s));
  *have = need;
  return BSMI_OK;
}

int bsmi_unet_train_begin(bsmi_unet* h, const int64_t in_shape[3]) {
  if (!h || !in_shape) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (!h->finalized[BSMI_PREC_F32]) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_finalize(BSMI_PREC_F32) first: training runs in fp32");
  BSMI_HIP(hipSetDevice(h->device));
  free_train_state(h);
  std::unique_ptr<TrainState> ts(new TrainState);
  ts->wstream_wanted = env_on("BSMI_TRAIN_WSTREAM");
  for (int d = 0; d < 3; ++d) ts->in_shape[d] = in_shape[d];
  int rc = get_plan(h, BSMI_PREC_F32, in_shape, &ts->plan);
  if (rc) return rc;
  // flat parameter buffer in the order of the weight table (= sorted state_dict keys)
  for (auto& kv : h->weights) {
    ParamRef pr;
    pr.key = kv.first;
    pr.shape = kv.second.shape;
    pr.count = kv.second.data.size();
    pr.off = ts->nparams;
    ts->nparams += (pr.count + 3) / 4 * 4;
    ts->index[pr.key] = ts->params.size();
    ts->params.push_back(pr);
  }
  const size_t pb = ts->nparams * sizeof(float);
  if ((rc = talloc(ts.get(), (void**)&ts->w, pb, true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->g, pb, true))) return rc;
  if (wgrad_x3_enabled(h) && (rc = talloc(ts.get(), (void**)&ts->gt, pb, true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->m, pb, true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->v, pb, true))) return rc;
  for (const ParamRef& pr : ts->params)
    BSMI_HIP(hipMemcpy(ts->w + pr.off, h->weights[pr.key].data.data(), pr.count * sizeof(float), hipMemcpyHostToDevice));
  if ((rc = talloc(ts.get(), (void**)&ts->zero_bias, 2048 * sizeof(float), true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->loss_sums, 4 * sizeof(double), true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->loss_part, 512 * 4 * sizeof(double), true))) return rc;
  if ((rc = talloc(ts.get(), (void**)&ts->loss_dev, sizeof(float), true))) return rc;

  Plan& plan = *ts->plan;
  ts->out_vox = (size_t)plan.out_shape[0] * plan.out_shape[1] * plan.out_shape[2];
  for (const HeadSite& hd : h->heads) {
    float *o = nullptr, *dp = nullptr;
    if ((rc = talloc(ts.get(), (void**)&o, ts->out_vox * hd.cout * sizeof(float), true))) return rc;
    if ((rc = talloc(ts.get(), (void**)&dp, ts->out_vox * hd.cout * sizeof(float), true))) return rc;
    ts->head_out.push_back(o);
    ts->head_dp.push_back(dp);
  }
  // forward pack jobs (conv stages, then heads are repacked by pack_head_kernel in the step)
  for (auto* sites : {&h->l_conv, &h->r_conv})
    for (PassSite& p : *sites)
      for (int ci = 0; ci < p.nconv; ++ci)
        if ((rc = make_forward_job(h, ts.get(), p, ci))) return rc;

  if (fwd_x3_enabled(h))
    for (size_t i = 0; i < plan.steps.size(); ++i) {
      if (plan.steps[i].type != PlanStep::CONV || (plan.fused_first && i < 3)) continue;
      if ((rc = make_forward_x3(h, ts.get(), plan.steps[i]))) return rc;
    }

  // backward data of the conv steps, in plan order; the first CONV step of the plan is the net's first conv
  ts->convs.resize(plan.steps.size());
  bool first_conv = true;
  for (size_t i = 0; i < plan.steps.size(); ++i) {
    const PlanStep& st = plan.steps[i];
    if (st.type != PlanStep::CONV) continue;
    ConvBwd& cb = ts->convs[i];
    cb.st = &st;
    PassSite& p = *st.site;
    const int n = p.nconv;
    int crop[3] = {0, 0, 0};
    for (int q = 0; q < n; ++q)
      for (int d = 0; d < 3; ++d) crop[d] += p.k[q][d] - 1;
    for (int d = 0; d < 3; ++d) cb.P[d] = st.ci == n - 1 ? std::max(p.k[st.ci][d] - 1, crop[d] / 2) : p.k[st.ci][d] - 1;
    if (p.k[st.ci][2] > 3) BSMI_FAIL(BSMI_ERR_INVALID, "training: kernels wider than 3 along x are not supported");
    cb.gp = st.out;
    cb.gp.D += 2 * cb.P[0]; cb.gp.H += 2 * cb.P[1]; cb.gp.W += 2 * cb.P[2];
    const size_t bytes = (size_t)cb.gp.D * cb.gp.H * cb.gp.W * cb.gp.Cpad * sizeof(float);
    const size_t slack = (size_t)8 * cb.gp.W * cb.gp.Cpad * sizeof(float) + 4096;
    if ((rc = talloc(ts.get(), &cb.gp.ptr, bytes + slack, true))) return rc;
    if (dgrad_x3_enabled(h) && (rc = talloc(ts.get(), &cb.gps, bytes + slack, true))) return rc;
    TDesc gy;
    if ((rc = grad_tensor(ts.get(), st.out, &gy))) return rc;
    cb.need_dgrad = !(first_conv && st.ci == 0);
    first_conv = false;
  }
  // dgrad launches need the padded gradient of the pass's LAST stage (residual source): second sweep
  for (size_t i = 0; i < plan.steps.size(); ++i) {
    ConvBwd& cb = ts->convs[i];
    if (!cb.st || !cb.need_dgrad) continue;
    const ConvBwd* last_cb = nullptr;
    for (size_t j = i; j < plan.steps.size(); ++j)
      if (ts->convs[j].st && ts->convs[j].st->site == cb.st->site && ts->convs[j].st->ci == cb.st->site->nconv - 1) {
        last_cb = &ts->convs[j];
        break;
      }
    if (!last_cb) BSMI_FAIL(BSMI_ERR_STATE, "training plan: last stage of %s not found", cb.st->site->prefix.c_str());
    cb.need_dgrad = false;
    if ((rc = make_dgrad(h, ts.get(), cb, last_cb))) return rc;
  }
  // gradient tensors of the remaining activations (pool / upsample / head inputs and outputs)
  for (const PlanStep& st : plan.steps) {
    TDesc t;
    if (st.type == PlanStep::POOL || st.type == PlanStep::UP) {
      if ((rc = grad_tensor(ts.get(), st.in, &t))) return rc;
      if ((rc = grad_tensor(ts.get(), st.out, &t))) return rc;
    } else if (st.type == PlanStep::HEAD) {
      if ((rc = grad_tensor(ts.get(), st.in, &t))) return rc;
    }
  }
  // gradient groups in completion order = the order in which the backward pass (plan steps in reverse) leaves them
  auto add_group = [&](const std::string& prefix) -> int {
    if (ts->group_of.count(prefix)) return BSMI_OK;
    TrainState::GradGroup g;
    g.prefix = prefix;
    size_t lo = (size_t)-1, hi = 0;
    for (const ParamRef& pr : ts->params)
      if (pr.key.compare(0, prefix.size() + 1, prefix + ".") == 0) {
        lo = std::min(lo, pr.off);
        hi = std::max(hi, pr.off + (pr.count + 3) / 4 * 4);
      }
    if (lo == (size_t)-1) BSMI_FAIL(BSMI_ERR_STATE, "training plan: no parameters under %s", prefix.c_str());
    g.off = lo;
    g.count = hi - lo;
    BSMI_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
    ts->group_of[prefix] = (int)ts->groups.size();
    ts->groups.push_back(g);
    return BSMI_OK;
  };
  for (size_t i = plan.steps.size(); i-- > 0;) {
    const PlanStep& st = plan.steps[i];
    if (st.type == PlanStep::HEAD) { if ((rc = add_group(h->heads[st.head].prefix))) return rc; }
    else if (st.type == PlanStep::CONV && st.ci == 0) { if ((rc = add_group(st.site->prefix))) return rc; }
  }
  {
    size_t covered = 0;
    for (auto& g : ts->groups) covered += g.count;
    if (covered != ts->nparams) BSMI_FAIL(BSMI_ERR_STATE, "training plan: gradient groups cover %zu of %zu parameters", covered, ts->nparams);
  }
  if (env_on("BSMI_TRAIN_WSTREAM")) {
    // ONE side stream per device and process, kept: the runtime has few hardware queues (GPU_MAX_HW_QUEUES), a process that
    // already runs 20 segmentation lanes and two predict lanes shares queues from the next stream on, and a stream per Trainer
    // (bench.py builds four) left later work in the process measurably slower
    static std::mutex mu;
    static hipStream_t side[16] = {nullptr};
    std::lock_guard<std::mutex> lk(mu);
    if (h->device >= 0 && h->device < 16) {
      if (!side[h->device]) BSMI_HIP(hipStreamCreateWithFlags(&side[h->device], hipStreamNonBlocking));
      ts->wstream = side[h->device];
    } else {  // (no such node; a stream of the state's own, destroyed with it)
      BSMI_HIP(hipStreamCreateWithFlags(&ts->wstream, hipStreamNonBlocking));
      ts->own_wstream = true;
    }
    BSMI_HIP(hipEventCreateWithFlags(&ts->ev_join, hipEventDisableTiming));
    BSMI_HIP(hipEventCreateWithFlags(&ts->ev_adam, hipEventDisableTiming));
    BSMI_HIP(hipEventCreateWithFlags(&ts->ev_packed, hipEventDisableTiming));
    BSMI_HIP(hipEventCreateWithFlags(&ts->ev_fwd_packed, hipEventDisableTiming));
    for (size_t i = 0; i < ts->convs.size(); ++i)
      if (plan.steps[i].type == PlanStep::CONV) BSMI_HIP(hipEventCreateWithFlags(&ts->convs[i].ev_g, hipEventDisableTiming));
  }
  h->train = ts.release();
  return run_pack_jobs(h->train, nullptr) || hipDeviceSynchronize() != hipSuccess ? BSMI_ERR_HIP : BSMI_OK;
}

int bsmi_unet_train_num_params(bsmi_unet* h, uint64_t* count) {
  if (!h || !h->train || !count) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  *count = h->train->nparams;
  return BSMI_OK;
}

int bsmi_unet_train_buffers(bsmi_unet* h, float** params_dev, float** grads_dev) {
  if (!h || !h->train) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  if (params_dev) *params_dev = h->train->w;
  if (grads_dev) *grads_dev = h->train->g;
  return BSMI_OK;
}

int bsmi_unet_train_param_info(bsmi_unet* h, const char* key, uint64_t* offset, uint64_t* count) {
  if (!h || !h->train || !key) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  auto it = h->train->index.find(key);
  if (it == h->train->index.end()) BSMI_FAIL(BSMI_ERR_MISSING, "no parameter \"%s\"", key);
  if (offset) *offset = h->train->params[it->second].off;
  if (count) *count = h->train->params[it->second].count;
  return BSMI_OK;
}

int bsmi_unet_train_forward_backward(bsmi_unet* h, const float* raw_dev, const float* const* targets_dev, const float* const* weights_dev,
                                     float* loss_host, void* stream) {
  if (!h || !h->train || !raw_dev || !targets_dev || !weights_dev) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  TrainState* ts = h->train;
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  Plan& plan = *ts->plan;
  const int nheads = (int)h->heads.size();
  const bool det = h->train_det != 0;
  // forward (the inference engine), sigmoid outputs kept for the loss
  h->train_forward = true;  // CONV steps with a split-bf16 form run it (PlanStep::tx3)
  int rc = bsmi_unet_forward(h, BSMI_PREC_F32, raw_dev, BSMI_RAW_F32, ts->in_shape, ts->head_out.data(), nullptr, stream);
  h->train_forward = false;
  if (rc) return rc;
  // clear gradients
  BSMI_HIP(hipMemsetAsync(ts->g, 0, ts->nparams * sizeof(float), s));
  for (auto& z : ts->zero_list) BSMI_HIP(hipMemsetAsync(z.first, 0, z.second, s));
  BSMI_HIP(hipMemsetAsync(ts->loss_dev, 0, sizeof(float), s));
  // loss and dL/dp per head
  for (int hd = 0; hd < nheads; ++hd) {
    const size_t n = ts->out_vox * h->heads[hd].cout;
    BSMI_HIP(hipMemsetAsync(ts->loss_sums, 0, 4 * sizeof(double), s));
    hipLaunchKernelGGL(loss_sums_kernel, dim3(512), dim3(256), 0, s, (const float*)ts->head_out[hd], targets_dev[hd], weights_dev[hd], n, ts->loss_sums,
                       det ? ts->loss_part : (double*)nullptr);
    if (det) hipLaunchKernelGGL(fold_kernel<double>, dim3(1), dim3(1024), 0, s, (const double*)ts->loss_part, 512, 4, 4, ts->loss_sums, (double*)nullptr, 1);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(512), dim3(256), 0, s, (const float*)ts->head_out[hd], targets_dev[hd], weights_dev[hd], n,
                       (const double*)ts->loss_sums, ts->head_dp[hd], ts->loss_dev);
  }
  // backward through the plan
  if (ts->packed_pending) {  // the input-gradient images of the last optimizer step (packed on the side stream)
    BSMI_HIP(hipStreamWaitEvent(s, ts->ev_packed, 0));
    ts->packed_pending = false;
  }
  for (size_t i = plan.steps.size(); i-- > 0;) {
    const PlanStep& st = plan.steps[i];
    switch (st.type) {
      case PlanStep::HEAD: {
        const HeadSite& hd = h->heads[st.head];
        TDesc dz = ts->grad_of[st.in.ptr];
        const std::string pre = hd.prefix;
        float* gwc = ts->g + param_off(ts, pre + ".conv_pass.0.weight");
        float* gwr = ts->g + param_off(ts, pre + ".residual.0.weight");
        float* gbc = ts->g + param_off(ts, pre + ".conv_pass.0.bias");
        float* gbr = ts->g + param_off(ts, pre + ".residual.0.bias");
        if (hd.cin > 32 || hd.cout > 16) BSMI_FAIL(BSMI_ERR_INVALID, "head backward: at most 32 input and 16 output channels");
        const size_t nv = ts->out_vox;
        const int nred = hd.cout * hd.cin + hd.cout;
        const unsigned hblocks = (unsigned)((nv + 255) / 256);
        if (det && (rc = grow_buf(s, &ts->det_part, &ts->det_part_bytes, (size_t)hblocks * nred * sizeof(float), false))) return rc;
        hipLaunchKernelGGL(head_bwd_kernel, dim3(hblocks), dim3(256), (size_t)(det ? 4 : 1) * nred * sizeof(float), s,
                           (const float*)st.in.ptr, st.in.Cpad, (const float*)ts->head_out[st.head], (const float*)ts->head_dp[st.head], nv, hd.cin,
                           hd.cout, (const float*)hd.hw, (float*)dz.ptr, gwc, gwr, gbc, gbr, det ? (float*)ts->det_part : (float*)nullptr);
        if (det) {  // the workgroups' rows in index order: weights, then biases
          const int nw = hd.cout * hd.cin;
          hipLaunchKernelGGL(fold_kernel<float>, dim3((nw + 31) / 32), dim3(1024), 0, s, (const float*)ts->det_part, (int)hblocks, nred, nw, gwc, gwr, 0);
          hipLaunchKernelGGL(fold_kernel<float>, dim3((hd.cout + 31) / 32), dim3(1024), 0, s, (const float*)ts->det_part + nw, (int)hblocks, nred, hd.cout, gbc, gbr, 0);
        }
        BSMI_HIP(hipEventRecord(ts->groups[ts->group_of[hd.prefix]].ev, s));
        break;
      }
      case PlanStep::UP: {
        TDesc din = ts->grad_of[st.in.ptr], dout = ts->grad_of[st.out.ptr];
        const size_t total = (size_t)st.out.D * st.out.H * st.out.W * st.out.Cpad;
        if (det) {
          const size_t total_in = (size_t)st.in.D * st.in.H * st.in.W * st.in.Cpad;
          hipLaunchKernelGGL(upsample_bwd_gather_kernel, dim3((unsigned)std::min<size_t>((total_in + 255) / 256, 65536)), dim3(256), 0, s,
                             (const float*)dout.ptr, (float*)din.ptr, st.in.D, st.in.H, st.in.W, st.in.Cpad, st.out.D, st.out.H, st.out.W, st.f[0],
                             st.f[1], st.f[2], st.o[0], st.o[1], st.o[2]);
          break;
        }
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, (const float*)dout.ptr,
                           (float*)din.ptr, st.in.D, st.in.H, st.in.W, st.in.Cpad, st.out.D, st.out.H, st.out.W, st.f[0], st.f[1], st.f[2], st.o[0],
                           st.o[1], st.o[2]);
        break;
      }
      case PlanStep::POOL: {
        TDesc din = ts->grad_of[st.in.ptr], dout = ts->grad_of[st.out.ptr];
        const size_t total = (size_t)st.out.D * st.out.H * st.out.W * st.out.Cpad;
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, (const float*)st.in.ptr,
                           (const float*)dout.ptr, (float*)din.ptr, st.in.H, st.in.W, st.in.Cpad, st.out.D, st.out.H, st.out.W, st.f[0], st.f[1], st.f[2]);
        break;
      }
      case PlanStep::CONV: {
        ConvBwd& cb = ts->convs[i];
        PassSite& p = *st.site;
        const int ci = st.ci, n = p.nconv;
        const bool last = ci == n - 1;
        const int* k = p.k[ci];
        TDesc gy = ts->grad_of[st.out.ptr];
        const size_t total4 = (size_t)st.out.D * st.out.H * st.out.W * (st.out.Cpad / 4);
        const std::string base = p.prefix + ".conv_pass." + std::to_string(2 * ci);
        float* gb = ts->g + param_off(ts, base + ".bias");
        float* gbr = last ? ts->g + param_off(ts, p.prefix + ".residual.0.bias") : nullptr;
        static const bool fuse_colsum = env_on("BSMI_TRAIN_FUSE_COLSUM");
        if (fuse_colsum && !det) {
          // the bias gradient (column sums of g) in the same pass: a grid whose stride is a multiple of the channel groups keeps a
          // thread on its four channels; few workgroups, each ends with one atomic per channel
          const int c4n = st.out.Cpad / 4;
          size_t blocks = std::min<size_t>((total4 + 255) / 256, 1024);
          if (c4n % 256 != 0) {  // stride = blocks * 256 = 0 mod c4n  <=  blocks = 0 mod (c4n / gcd(c4n, 256))
            int a = c4n, b = 256;
            while (b) { const int t = a % b; a = b; b = t; }
            const size_t q = (size_t)c4n / a;
            if (blocks >= q) blocks = blocks / q * q;
          }
          hipLaunchKernelGGL(relu_bwd_pad_kernel, dim3((unsigned)blocks), dim3(256), (size_t)st.out.Cpad * sizeof(float), s, (const float*)gy.ptr,
                             (const float*)st.out.ptr, st.out.D, st.out.H, st.out.W, st.out.Cpad, cb.P[0], cb.P[1], cb.P[2], (float*)cb.gp.ptr,
                             (uint16_t*)cb.gps, p.cout, gb, gbr);
        } else {
          hipLaunchKernelGGL(relu_bwd_pad_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 16384)), dim3(256), 0, s, (const float*)gy.ptr,
                             (const float*)st.out.ptr, st.out.D, st.out.H, st.out.W, st.out.Cpad, cb.P[0], cb.P[1], cb.P[2], (float*)cb.gp.ptr,
                             (uint16_t*)cb.gps, 0, (float*)nullptr, (float*)nullptr);
          for (int c0 = 0; c0 < st.out.Cpad; c0 += 512) {
            const int Cc = std::min(512, st.out.Cpad - c0);
            const int threads = std::max(Cc, 256 / Cc * Cc);
            const int lanes = threads / Cc;
            if (det && (rc = grow_buf(s, &ts->det_part, &ts->det_part_bytes, (size_t)256 * st.out.Cpad * sizeof(float), false))) return rc;
            if (threads != lanes * Cc) BSMI_FAIL(BSMI_ERR_STATE, "column sums: block of %d threads for %d channels", threads, Cc);
            hipLaunchKernelGGL(colsum_kernel, dim3(256), dim3(threads), det ? (size_t)threads * sizeof(float) : 0, s, (const float*)cb.gp.ptr, st.out.D, st.out.H, st.out.W, st.out.Cpad, c0, Cc,
                               cb.P[0], cb.P[1], cb.P[2], p.cout, gb, gbr, det ? (float*)ts->det_part : (float*)nullptr);
            if (det) {  // the real channels of this chunk, rows in index order
              const int wd = std::min(Cc, p.cout - c0);
              if (wd > 0)
                hipLaunchKernelGGL(fold_kernel<float>, dim3((wd + 31) / 32), dim3(1024), 0, s, (const float*)ts->det_part + c0, 256, st.out.Cpad, wd,
                                   gb + c0, gbr ? gbr + c0 : (float*)nullptr, 0);
            }
          }
        }
        // weight gradients: on their own stream (ts->wstream), beside the input-gradient launch of this stage and whatever the main
        // stream does next -- they only read g (ready: the event below) and forward activations, and write dW.  Many launches of
        // the step have too few tiles for the card; two chains of them side by side fill it better.  BSMI_TRAIN_WSTREAM=0: one stream.
        hipStream_t sw = s;
        if (ts->wstream) {
          sw = ts->wstream;
          BSMI_HIP(hipEventRecord(cb.ev_g, s));
          BSMI_HIP(hipStreamWaitEvent(sw, cb.ev_g, 0));
        }
        const int64_t gsx = cb.gp.Cpad, gsy = (int64_t)cb.gp.W * gsx, gsz = (int64_t)cb.gp.H * gsy;
        const float* ginterior = (const float*)cb.gp.ptr + cb.P[0] * gsz + cb.P[1] * gsy + cb.P[2] * gsx;
        const int cin_total = ci == 0 ? p.cin[0] + (p.nslots > 1 ? p.cin[1] : 0) : p.cout;
        bool used_x3 = false, g_packed = false;
        int x3_rc = BSMI_OK;
        // deterministic mode: the launches of one weight tensor add into per-line-range copies of its workspace (det_nz of them,
        // det_numel floats apart, in ts->gt_det), which the finish kernel adds in order
        int det_nz = 1;
        size_t det_numel = 0;
        auto finish = [&](float* dw, int ct, const int* kk) {  // after the launches of one weight tensor
          if (!used_x3) return;
          const size_t nc = (size_t)p.cout * ct;
          if (det && kk[0] * kk[1] * kk[2] > 27) { x3_rc = BSMI_ERR_INVALID; bsmi::set_error("deterministic weight gradients: kernels of at most 27 taps"); return; }
          if (det)
            hipLaunchKernelGGL(wgrad_finish_det_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(1024), 0, sw, (float*)ts->gt_det, dw, nc,
                               kk[0] * kk[1] * kk[2], det_nz, det_numel);
          else
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, sw, ts->gt + (dw - ts->g), dw, nc,
                               kk[0] * kk[1] * kk[2], 1, (size_t)0);
          used_x3 = false;
        };
        auto grow = [&](char** buf, size_t* have, size_t need) -> int { return grow_buf(sw, buf, have, need, false); };  // first steps only
        // the line ranges launch_wgrad_x3_t will cut a slot's launch into (its arithmetic)
        auto x3_zsplit = [&](int N, int C, int nlines, int trows) {
          const int TN = wgrad_tile_n(N), TC = wgrad_tile_c(C);
          const int blocks_nc = ((N + TN - 1) / TN) * ((C + TC - 1) / TC);
          const int zs = std::min(kDetMaxRanges, std::max(1, std::min(nlines, 4096 / std::max(1, blocks_nc * trows))));
          const int lpb = (nlines + zs - 1) / zs;
          return (nlines + lpb - 1) / lpb;
        };
        auto det_prepare = [&](int ct, const int* kk, const int* slot_c, int nsl) -> int {  // before the launches of one weight tensor
          if (!det || !ts->gt) return BSMI_OK;
          det_numel = (size_t)kk[0] * kk[1] * kk[2] * p.cout * ct;
          det_nz = 1;
          for (int sl = 0; sl < nsl; ++sl) det_nz = std::max(det_nz, x3_zsplit(p.cout, slot_c[sl], st.out.D * st.out.H, kk[0] * kk[1]));
          return grow_buf(sw, &ts->gt_det, &ts->gt_det_bytes, (size_t)det_nz * det_numel * sizeof(float), true);
        };
        // split-bf16 form (wgrad_x3_kernel): pack g once per conv stage, x per launch
        auto wgrad_x3 = [&](const WgradArgs& a) -> int {
          const int gpl = (a.Wo + 7) / 8, nlines = a.Do * a.Ho;
          const int Np = wgrad_pad(a.N, wgrad_tile_n(a.N)), Cp = wgrad_pad(a.C, wgrad_tile_c(a.C)), xvec = a.kx > 1 ? 2 : 1;
          const int Dil = a.Do + a.kz - 1, Hil = a.Ho + a.ky - 1;
          int rc2;
          if (!g_packed) {
            const size_t need = ((size_t)nlines * gpl + 1) * 2 * Np * 16;
            if ((rc2 = grow(&ts->pk_g, &ts->pk_g_bytes, need))) return rc2;
            const size_t items = ((size_t)nlines * gpl + 1) * Np;
            if (items >= ((size_t)1 << 31)) return BSMI_ERR_INVALID;
            hipLaunchKernelGGL(wgrad_pack_kernel, dim3((unsigned)std::min<size_t>((items + 255) / 256, 65536)), dim3(256), 0, sw, a.g, a.gsz, a.gsy,
                               a.gsx, a.Do, a.Ho, a.Wo, a.N, Np, gpl, 1, 1, (u32x4_t*)ts->pk_g);
            g_packed = true;
          }
          const size_t needx = (size_t)Dil * Hil * gpl * 2 * xvec * Cp * 16;
          if ((rc2 = grow(&ts->pk_x, &ts->pk_x_bytes, needx))) return rc2;
          const size_t itemsx = (size_t)Dil * Hil * gpl * Cp;
          if (itemsx >= ((size_t)1 << 31)) return BSMI_ERR_INVALID;
          hipLaunchKernelGGL(wgrad_pack_kernel, dim3((unsigned)std::min<size_t>((itemsx + 255) / 256, 65536)), dim3(256), 0, sw, a.x, a.xsz, a.xsy, a.xsx,
                             Dil, Hil, a.Wo + a.kx - 1, a.C, Cp, gpl, xvec, 0, (u32x4_t*)ts->pk_x);
          WgradPk k;
          k.gp = ts->pk_g; k.xp = ts->pk_x; k.Np = Np; k.Cp = Cp; k.gpl = gpl;
          k.Do = a.Do; k.Ho = a.Ho; k.Hil = Hil; k.N = a.N; k.C = a.C; k.kz = a.kz; k.ky = a.ky;
          k.dwt = det ? (float*)ts->gt_det : a.dwt; k.cin_total = a.cin_total; k.cbase = a.cbase; k.ntap = a.ntap; k.lines_per_block = 0; k.zsplit = 1;
          k.zstride = det ? det_numel : 0;
          if (det) k.zsplit = det_nz;
          return a.kx == 1 ? launch_wgrad_x3_k<1>(k, sw) : launch_wgrad_x3_k<3>(k, sw);
        };
        auto wgrad = [&](const TDesc& x, const int* org, int C, int cbase, float* dw, int ct, const int* kk) {
          WgradArgs a;
          a.g = ginterior; a.gsz = gsz; a.gsy = gsy; a.gsx = gsx;
          a.xsx = x.Cpad; a.xsy = (int64_t)x.W * a.xsx; a.xsz = (int64_t)x.H * a.xsy;
          a.x = (const float*)x.ptr + org[0] * a.xsz + org[1] * a.xsy + org[2] * a.xsx;
          a.Do = st.out.D; a.Ho = st.out.H; a.Wo = st.out.W;
          a.N = p.cout; a.C = C;
          a.kz = kk[0]; a.ky = kk[1]; a.kx = kk[2];
          a.dw = dw; a.cin_total = ct; a.cbase = cbase; a.ntap = kk[0] * kk[1] * kk[2];
          a.dwt = ts->gt ? ts->gt + (dw - ts->g) : nullptr;
          const int nlines = a.Do * a.Ho;
          const int trows = a.kz * a.ky;
          if (a.dwt && (a.kx == 1 || a.kx == 3)) {
            const int rc2 = wgrad_x3(a);
            if (rc2) x3_rc = rc2;
            used_x3 = true;
            return;
          }
          const bool tiled = a.N > 32 && a.C > 32;  // narrow layers: the per-wave form wastes fewer MFMAs on padding
          const int blocks_nc = tiled ? ((a.N + 127) / 128) * ((a.C + 127) / 128) : ((a.N + 31) / 32) * ((a.C + 63) / 64);
          int zsplit = std::max(1, std::min(nlines, (tiled ? 2048 : 8192) / std::max(1, blocks_nc * trows)));
          if (det) zsplit = 1;  // the f32 forms add straight into dw: one workgroup per element = one (exact) addition to a zero
          a.lines_per_block = (nlines + zsplit - 1) / zsplit;
          zsplit = (nlines + a.lines_per_block - 1) / a.lines_per_block;
          const dim3 grid(blocks_nc, trows, zsplit);
          if (tiled) {
            switch (a.kx) {
              case 1: hipLaunchKernelGGL(wgrad_tiled_kernel<1>, grid, dim3(256), 0, sw, a); break;
              case 2: hipLaunchKernelGGL(wgrad_tiled_kernel<2>, grid, dim3(256), 0, sw, a); break;
              case 3: hipLaunchKernelGGL(wgrad_tiled_kernel<3>, grid, dim3(256), 0, sw, a); break;
              default: return;
            }
            return;
          }
          switch (a.kx) {
            case 1: hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(64), 0, sw, a); break;
            case 2: hipLaunchKernelGGL(wgrad_kernel<2>, grid, dim3(64), 0, sw, a); break;
            case 3: hipLaunchKernelGGL(wgrad_kernel<3>, grid, dim3(64), 0, sw, a); break;
            default: return;  // checked in bsmi_unet_train_begin
          }
        };
        float* dwm = ts->g + param_off(ts, base + ".weight");
        {
          const int one_slot[1] = {p.cout};
          if ((rc = det_prepare(cin_total, k, ci == 0 ? p.cin : one_slot, ci == 0 ? p.nslots : 1))) return rc;
        }
        if (ci == 0) {
          int cbase = 0;
          for (int sl = 0; sl < p.nslots; ++sl) {
            wgrad(st.slots[sl], st.so[sl], p.cin[sl], cbase, dwm, cin_total, k);
            cbase += p.cin[sl];
          }
        } else {
          wgrad(st.slots[0], st.so[0], p.cout, 0, dwm, cin_total, k);
        }
        finish(dwm, cin_total, k);
        if (x3_rc) return x3_rc;
        if (last) {
          int crop[3] = {0, 0, 0};
          for (int q = 0; q < n; ++q)
            for (int d = 0; d < 3; ++d) crop[d] += p.k[q][d] - 1;
          float* dwr = ts->g + param_off(ts, p.prefix + ".residual.0.weight");
          const int first_slot = ci == 0 ? 0 : 1;
          const int ones[3] = {1, 1, 1};
          const int rin = p.cin[0] + (p.nslots > 1 ? p.cin[1] : 0);
          int cbase = 0;
          if ((rc = det_prepare(rin, ones, p.cin, p.nslots))) return rc;
          for (int sl = 0; sl < p.nslots; ++sl) {
            int org[3];
            for (int d = 0; d < 3; ++d) org[d] = st.so[first_slot + sl][d] + crop[d] / 2;
            wgrad(st.slots[first_slot + sl], org, p.cin[sl], cbase, dwr, rin, ones);
            cbase += p.cin[sl];
          }
          finish(dwr, rin, ones);
          if (x3_rc) return x3_rc;
        }
        // input gradient
        if (cb.need_dgrad) {
          rc = launch_conv_igemm(cb.dgrad, cb.dx3 ? BSMI_PREC_BF16X3 : BSMI_PREC_F32, cb.dtile, s, h->sk_ws, h->sk_grid);
          if (rc) return rc;
          if (cb.dsplit) {
            const size_t g8 = (size_t)cb.dcat.D * cb.dcat.H * cb.dcat.W * cb.dcat.Cpad / 8;
            hipLaunchKernelGGL(split_to_f32_kernel, dim3((unsigned)std::min<size_t>((g8 + 255) / 256, 16384)), dim3(256), 0, s,
                               (const uint4*)cb.dsplit, (float4*)cb.dcat.ptr, g8);
          }
          if (cb.scatter) {
            int cbase = 0;
            for (int sl = 0; sl < p.nslots; ++sl) {
              TDesc gt = ts->grad_of[st.slots[sl].ptr];
              if (!gt.ptr) BSMI_FAIL(BSMI_ERR_STATE, "training plan: no gradient tensor for an input of %s", p.prefix.c_str());
              const size_t total = (size_t)cb.dcat.D * cb.dcat.H * cb.dcat.W * p.cin[sl];
              hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s,
                                 (const float*)cb.dcat.ptr, cb.dcat.D, cb.dcat.H, cb.dcat.W, cb.dcat.Cpad, cbase, (float*)gt.ptr, gt.H, gt.W, gt.Cpad, 0,
                                 st.so[sl][0], st.so[sl][1], st.so[sl][2], p.cin[sl]);
              cbase += p.cin[sl];
            }
          }
        }
        // the pass's gradients are final when its last weight gradient is (its bias gradients were written before ev_g)
        if (ci == 0) BSMI_HIP(hipEventRecord(ts->groups[ts->group_of[p.prefix]].ev, ts->wstream ? ts->wstream : s));
        break;
      }
      default: break;
    }
  }
  BSMI_HIP(hipGetLastError());
  if (ts->wstream) {  // the caller's stream ends the pass after the last weight gradient
    BSMI_HIP(hipEventRecord(ts->ev_join, ts->wstream));
    BSMI_HIP(hipStreamWaitEvent(s, ts->ev_join, 0));
  }
  if (loss_host) {
    BSMI_HIP(hipMemcpyAsync(loss_host, ts->loss_dev, sizeof(float), hipMemcpyDeviceToHost, s));
    BSMI_HIP(hipStreamSynchronize(s));
  }
  return BSMI_OK;
}

int bsmi_unet_train_last_loss(bsmi_unet* h, float* loss_host, void* stream) {
  if (!h || !h->train || !loss_host) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipMemcpyAsync(loss_host, h->train->loss_dev, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  BSMI_HIP(hipStreamSynchronize((hipStream_t)stream));
  return BSMI_OK;
}

int bsmi_unet_train_prediction(bsmi_unet* h, int head, float** out_dev, uint64_t* count) {
  if (!h || !h->train || !out_dev) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  if (head < 0 || head >= (int)h->train->head_out.size()) BSMI_FAIL(BSMI_ERR_INVALID, "head %d out of range", head);
  *out_dev = h->train->head_out[head];
  if (count) *count = (uint64_t)h->train->out_vox * h->heads[head].cout;
  return BSMI_OK;
}

int bsmi_unet_train_grad_groups(bsmi_unet* h, int max_n, int* n, uint64_t* offsets, uint64_t* counts) {
  if (!h || !h->train || !n) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  *n = (int)h->train->groups.size();
  for (int i = 0; i < *n && i < max_n; ++i) {
    if (offsets) offsets[i] = h->train->groups[i].off;
    if (counts) counts[i] = h->train->groups[i].count;
  }
  return BSMI_OK;
}

int bsmi_unet_train_wait_grad_group(bsmi_unet* h, int group, void* stream) {
  if (!h || !h->train) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  if (group < 0 || group >= (int)h->train->groups.size()) BSMI_FAIL(BSMI_ERR_INVALID, "gradient group %d out of range", group);
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipStreamWaitEvent((hipStream_t)stream, h->train->groups[group].ev, 0));
  return BSMI_OK;
}

int bsmi_unet_train_write_param(bsmi_unet* h, const char* key, int what, const float* host_in) {
  if (!h || !h->train || !key || !host_in) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  if (what != 2 && what != 3) BSMI_FAIL(BSMI_ERR_INVALID, "only the Adam moments (2, 3) can be written; parameters go through bsmi_unet_load_weight");
  auto it = h->train->index.find(key);
  if (it == h->train->index.end()) BSMI_FAIL(BSMI_ERR_MISSING, "no parameter \"%s\"", key);
  const ParamRef& pr = h->train->params[it->second];
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipDeviceSynchronize());
  BSMI_HIP(hipMemcpy((what == 2 ? h->train->m : h->train->v) + pr.off, host_in, pr.count * sizeof(float), hipMemcpyHostToDevice));
  return BSMI_OK;
}

int bsmi_unet_train_step_count(bsmi_unet* h, int set_to, int* value) {
  if (!h || !h->train) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  if (set_to >= 0) h->train->adam_t = set_to;
  if (value) *value = h->train->adam_t;
  return BSMI_OK;
}

int bsmi_unet_train_adam_step(bsmi_unet* h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (!h || !h->train) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called");
  TrainState* ts = h->train;
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  ts->adam_t += 1;
  const float bc1 = 1.f - powf(beta1, (float)ts->adam_t);
  const float bc2 = 1.f - powf(beta2, (float)ts->adam_t);
  hipLaunchKernelGGL(adam_kernel, dim3(1024), dim3(256), 0, s, ts->w, (const float*)ts->g, ts->m, ts->v, ts->nparams, lr, beta1, beta2, eps, bc1, sqrtf(bc2),
                     grad_scale);
  int rc;
  if (ts->wstream) {
    // the images of the input-gradient launches are first read in the NEXT backward pass: they are packed on the side stream,
    // beside the forward images here and the forward pass that follows (bsmi_unet_train_forward_backward waits for ev_packed
    // before its backward pass)
    BSMI_HIP(hipEventRecord(ts->ev_adam, s));
    BSMI_HIP(hipStreamWaitEvent(ts->wstream, ts->ev_adam, 0));
    if ((rc = run_pack_jobs(ts, ts->wstream, true, false, 3))) return rc;
    BSMI_HIP(hipEventRecord(ts->ev_fwd_packed, ts->wstream));
    ts->fwd_packed_pending = true;
    if ((rc = run_pack_jobs(ts, ts->wstream, true, false, 2))) return rc;
    BSMI_HIP(hipEventRecord(ts->ev_packed, ts->wstream));
    ts->packed_pending = true;
    rc = run_pack_jobs(ts, s, true, false, 1);
  } else {
    rc = run_pack_jobs(ts, s, /*lazy_f32=*/true);
  }
  if (rc) return rc;
  for (HeadSite& hd : h->heads) {
    const std::string pre = hd.prefix;
    hipLaunchKernelGGL(pack_head_kernel, dim3((hd.cout * hd.cin + 255) / 256), dim3(256), 0, s, (const float*)ts->w,
                       (long long)param_off(ts, pre + ".conv_pass.0.weight"), (long long)param_off(ts, pre + ".residual.0.weight"),
                       (long long)param_off(ts, pre + ".conv_pass.0.bias"), (long long)param_off(ts, pre + ".residual.0.bias"), hd.cout, hd.cin, hd.hw, hd.hb);
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int bsmi_unet_train_read_param(bsmi_unet* h, const char* key, int what, float* host_out) {
  if (!h || !h->train || !key || !host_out) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_train_begin has not been called / null argument");
  auto it = h->train->index.find(key);
  if (it == h->train->index.end()) BSMI_FAIL(BSMI_ERR_MISSING, "no parameter \"%s\"", key);
  const ParamRef& pr = h->train->params[it->second];
  const float* src = what == 0 ? h->train->w : (what == 1 ? h->train->g : (what == 2 ? h->train->m : h->train->v));
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipDeviceSynchronize());
  BSMI_HIP(hipMemcpy(host_out, src + pr.off, pr.count * sizeof(float), hipMemcpyDeviceToHost));
  return BSMI_OK;
}

int bsmi_train_affinity_targets(int device, int64_t* labels_dev, const uint8_t* unlabelled_dev, const int64_t shape[3],
                                const int32_t* neighborhood, int n, int grow_steps, int only_xy, float clip_min, float clip_max,
                                float* affs_dev, float* weights_dev, void* stream) {
  if (!labels_dev || !shape || !neighborhood || !affs_dev || !weights_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n < 1 || n > kMaxNeighborhood) BSMI_FAIL(BSMI_ERR_INVALID, "neighborhood of %d offsets (1..%d supported)", n, kMaxNeighborhood);
  if (grow_steps < 0 || grow_steps > 16) BSMI_FAIL(BSMI_ERR_INVALID, "grow_steps %d outside 0..16", grow_steps);
  for (int d = 0; d < 3; ++d)
    if (shape[d] < 1 || shape[d] > 4096) BSMI_FAIL(BSMI_ERR_INVALID, "bad shape");
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
  const size_t nvox = (size_t)D * H * W;
  Neighborhood nb;
  nb.n = n;
  for (int e = 0; e < n; ++e)
    for (int d = 0; d < 3; ++d) nb.off[e][d] = neighborhood[3 * e + d];
  // scratch on the stream: the grown labels (the erosion reads its neighbours' old values) and two counters
  int64_t* grown = nullptr;
  unsigned long long* counts = nullptr;
  BSMI_HIP(hipMallocAsync((void**)&grown, nvox * sizeof(int64_t) + 2 * sizeof(unsigned long long), s));
  counts = (unsigned long long*)(grown + nvox);
  BSMI_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), s));
  const int bs = 256;
  const unsigned grid = (unsigned)std::min<size_t>((nvox + bs - 1) / bs, 65535);
  hipLaunchKernelGGL(grow_boundary_kernel, dim3(grid), dim3(bs), 0, s, labels_dev, unlabelled_dev, grown, D, H, W, grow_steps, only_xy);
  BSMI_HIP(hipMemcpyAsync(labels_dev, grown, nvox * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(affinity_targets_kernel, dim3(grid), dim3(bs), 0, s, grown, unlabelled_dev, nb, D, H, W, affs_dev, weights_dev, counts);
  const size_t total = nvox * (size_t)n;
  hipLaunchKernelGGL(balance_kernel, dim3((unsigned)std::min<size_t>((total + bs - 1) / bs, 65535)), dim3(bs), 0, s, affs_dev, weights_dev, total,
                     counts, clip_min, clip_max);
  BSMI_HIP(hipGetLastError());
  BSMI_HIP(hipFreeAsync(grown, s));
  return BSMI_OK;
}

// ---- local shape descriptors (3-D, 10 channels) -------------------------------------------------------------
// lsd.train.LsdExtractor.get_descriptors as AddLocalShapeDescriptor calls it (reference models/3d_mtlsd/train.py:134-141;
// the lsd package is not in /root/reference: restated from its published algorithm, see oracle/lsd_ref.py).  For a voxel p
// of object l the statistics are those of l inside a Gaussian window around p's cell of the `df`-times sub-sampled grid:
//   count = sum_t w(t - s) [label(t) == l],  mean = sum w c(t) / count,  cov = sum w c c^T / count - mean mean^T
// with s = p / df (integer), t over the sub-sampled grid, c = world coordinates of the sub-grid points and w the product
// of normalised 1-D Gaussians truncated at 3 sigma (scipy.ndimage.gaussian_filter(mode="constant", truncate=3.0)).
// Channels: mean - c(s) (z, y, x) / sigma * 0.5 + 0.5 | variances / sigma^2 | Pearson zy, zx, yx * 0.5 + 0.5 | count;
// clipped to [0, 1]; background voxels are all zero.  Coordinates are taken relative to s (the differences are what
// enters; the library's absolute float32 coordinates only add rounding).
struct LsdArgs {
  const int64_t* labels;  // [D][H][W] with the context the window needs
  int D, H, W;
  int oz, oy, ox, d, h, w;  // output ROI inside the label array
  int df;                   // sub-sampling factor
  int rz, ry, rx;           // window radii on the sub-sampled grid
  float step[3];            // world distance between sub-grid points
  float sigma[3];           // world units
  const float* wz; const float* wy; const float* wx;  // normalised 1-D weights [2r + 1]
};

__global__ void lsd_targets_kernel(LsdArgs a, const uint8_t* __restrict__ unl, float* __restrict__ lsds, float* __restrict__ weights) {
  const size_t nout = (size_t)a.d * a.h * a.w;
  const int SD = a.D / a.df, SH = a.H / a.df, SW = a.W / a.df;  // sub-sampled extent (labels[::df])
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < nout; p += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % a.w), y = (int)((p / a.w) % a.h), z = (int)(p / ((size_t)a.w * a.h));
    const int Z = z + a.oz, Y = y + a.oy, X = x + a.ox;
    const size_t q0 = ((size_t)Z * a.H + Y) * a.W + X;
    const int64_t l = a.labels[q0];
    float out[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (l != 0) {
      const int sz = Z / a.df, sy = Y / a.df, sx = X / a.df;
      double n = 0, m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
      for (int dz = -a.rz; dz <= a.rz; ++dz) {
        const int tz = sz + dz;
        if (tz < 0 || tz >= SD) continue;
        const float gz = a.wz[dz + a.rz];
        for (int dy = -a.ry; dy <= a.ry; ++dy) {
          const int ty = sy + dy;
          if (ty < 0 || ty >= SH) continue;
          const float gzy = gz * a.wy[dy + a.ry];
          const int64_t* row = a.labels + ((size_t)(tz * a.df) * a.H + (size_t)ty * a.df) * a.W;
          for (int dx = -a.rx; dx <= a.rx; ++dx) {
            const int tx = sx + dx;
            if (tx < 0 || tx >= SW) continue;
            if (row[(size_t)tx * a.df] != l) continue;
            const double wgt = (double)(gzy * a.wx[dx + a.rx]);
            const double cz = dz * (double)a.step[0], cy = dy * (double)a.step[1], cx = dx * (double)a.step[2];
            n += wgt;
            m[0] += wgt * cz; m[1] += wgt * cy; m[2] += wgt * cx;
            c[0] += wgt * cz * cz; c[1] += wgt * cy * cy; c[2] += wgt * cx * cx;
            c[3] += wgt * cz * cy; c[4] += wgt * cz * cx; c[5] += wgt * cy * cx;
          }
        }
      }
      const double cnt = n == 0 ? 1.0 : n;
      double mean[3], var[3], pe[3];
      for (int i = 0; i < 3; ++i) mean[i] = m[i] / cnt;
      for (int i = 0; i < 3; ++i) var[i] = c[i] / cnt - mean[i] * mean[i];
      pe[0] = c[3] / cnt - mean[0] * mean[1];
      pe[1] = c[4] / cnt - mean[0] * mean[2];
      pe[2] = c[5] / cnt - mean[1] * mean[2];
      for (int i = 0; i < 3; ++i) var[i] = var[i] < 1e-3 ? 1e-3 : var[i];
      pe[0] /= sqrt(var[0] * var[1]);
      pe[1] /= sqrt(var[0] * var[2]);
      pe[2] /= sqrt(var[1] * var[2]);
      for (int i = 0; i < 3; ++i) {
        out[i] = (float)(mean[i] / a.sigma[i] * 0.5 + 0.5);
        out[3 + i] = (float)(var[i] / ((double)a.sigma[i] * a.sigma[i]));
        out[6 + i] = (float)(pe[i] * 0.5 + 0.5);
      }
      out[9] = (float)n;
      for (int i = 0; i < 10; ++i) out[i] = out[i] < 0.f ? 0.f : (out[i] > 1.f ? 1.f : out[i]);
    }
    // lsds_mask: labelled voxels, times the known-voxel mask (AddLocalShapeDescriptor.process)
    const float wv = (l != 0 && (!unl || unl[q0])) ? 1.f : 0.f;
    for (int i = 0; i < 10; ++i) {
      lsds[(size_t)i * nout + p] = out[i];
      if (weights) weights[(size_t)i * nout + p] = wv;
    }
  }
}

extern "C" int bsmi_train_lsd_targets(int device, const int64_t* labels_dev, const uint8_t* unlabelled_dev, const int64_t shape[3],
                                      const int64_t roi_offset[3], const int64_t roi_shape[3], const float sigma[3],
                                      const float voxel_size[3], int downsample, float* lsds_dev, float* weights_dev, void* stream) {
  if (!labels_dev || !shape || !roi_offset || !roi_shape || !sigma || !voxel_size || !lsds_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (downsample < 1 || downsample > 8) BSMI_FAIL(BSMI_ERR_INVALID, "downsample %d outside 1..8", downsample);
  LsdArgs a;
  a.labels = labels_dev;
  a.D = (int)shape[0]; a.H = (int)shape[1]; a.W = (int)shape[2];
  a.oz = (int)roi_offset[0]; a.oy = (int)roi_offset[1]; a.ox = (int)roi_offset[2];
  a.d = (int)roi_shape[0]; a.h = (int)roi_shape[1]; a.w = (int)roi_shape[2];
  a.df = downsample;
  for (int i = 0; i < 3; ++i) {
    if (shape[i] < 1 || shape[i] > 4096 || roi_shape[i] < 1 || roi_offset[i] < 0 || roi_offset[i] + roi_shape[i] > shape[i])
      BSMI_FAIL(BSMI_ERR_INVALID, "bad shape / ROI");
    if (shape[i] % downsample || roi_offset[i] % downsample || roi_shape[i] % downsample)
      BSMI_FAIL(BSMI_ERR_INVALID, "shape and ROI must be multiples of the downsample factor %d (as the lsd package requires)", downsample);
    if (!(sigma[i] > 0.f) || !(voxel_size[i] > 0.f)) BSMI_FAIL(BSMI_ERR_INVALID, "sigma and voxel_size must be positive");
    a.sigma[i] = sigma[i];
    a.step[i] = voxel_size[i] * downsample;
  }
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  // normalised 1-D weights as scipy's gaussian_filter1d builds them (sigma in sub-grid voxels, truncate = 3.0)
  int r[3];
  std::vector<float> w[3];
  for (int i = 0; i < 3; ++i) {
    const double sv = (double)sigma[i] / ((double)voxel_size[i] * downsample);
    r[i] = (int)(3.0 * sv + 0.5);
    if (r[i] > 512) BSMI_FAIL(BSMI_ERR_INVALID, "LSD window radius %d too large", r[i]);
    std::vector<double> g(2 * r[i] + 1);
    double sum = 0;
    for (int k = -r[i]; k <= r[i]; ++k) sum += g[k + r[i]] = exp(-0.5 * (double)k * k / (sv * sv));
    w[i].resize(g.size());
    for (size_t k = 0; k < g.size(); ++k) w[i][k] = (float)(g[k] / sum);
  }
  a.rz = r[0]; a.ry = r[1]; a.rx = r[2];
  float* wdev = nullptr;
  const size_t nw = w[0].size() + w[1].size() + w[2].size();
  BSMI_HIP(hipMallocAsync((void**)&wdev, nw * sizeof(float), s));
  std::vector<float> all;
  for (int i = 0; i < 3; ++i) all.insert(all.end(), w[i].begin(), w[i].end());
  // the host vector must outlive the asynchronous copy: copy synchronously (a few hundred bytes)
  BSMI_HIP(hipStreamSynchronize(s));
  BSMI_HIP(hipMemcpy(wdev, all.data(), nw * sizeof(float), hipMemcpyHostToDevice));
  a.wz = wdev; a.wy = wdev + w[0].size(); a.wx = wdev + w[0].size() + w[1].size();
  const size_t nout = (size_t)a.d * a.h * a.w;
  hipLaunchKernelGGL(lsd_targets_kernel, dim3((unsigned)std::min<size_t>((nout + 127) / 128, 65535)), dim3(128), 0, s, a, unlabelled_dev, lsds_dev,
                     weights_dev);
  BSMI_HIP(hipGetLastError());
  BSMI_HIP(hipFreeAsync(wdev, s));
  return BSMI_OK;
}

int bsmi_unet_train_end(bsmi_unet* h) {
  if (h && h->train) {
    // the trained parameters become the handle's weights: host copies refreshed, the bf16 images re-packed on demand
    BSMI_HIP(hipSetDevice(h->device));
    BSMI_HIP(hipDeviceSynchronize());
    const int rrc = train_refresh_f32_images(h, nullptr);  // the f32 images stay the handle's: not left stale
    if (rrc) return rrc;
    BSMI_HIP(hipDeviceSynchronize());
    for (const ParamRef& pr : h->train->params)
      BSMI_HIP(hipMemcpy(h->weights[pr.key].data.data(), h->train->w + pr.off, pr.count * sizeof(float), hipMemcpyDeviceToHost));
    for (auto* sites : {&h->l_conv, &h->r_conv})
      for (PassSite& p : *sites)
        for (int ci = 0; ci < p.nconv; ++ci)
          for (int prec : {BSMI_PREC_BF16, BSMI_PREC_BF16X3}) {
            PackedConv& pc = p.packed[prec][ci];
            if (pc.w) (void)hipFree(pc.w);
            if (pc.bias) (void)hipFree(pc.bias);
            pc = PackedConv();
          }
    h->finalized[BSMI_PREC_BF16] = false;
    h->finalized[BSMI_PREC_BF16X3] = false;
  }
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipDeviceSynchronize());
  free_train_state(h);
  return BSMI_OK;
}

}  // extern "C"
