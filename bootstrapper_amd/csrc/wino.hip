// Winograd F(2x2, 3x3) over the in-plane axes of a 3x3x3 "valid" convolution, split-bf16 mode (BSMI_PREC_BF16X3).
//
// out[z][y][x][n] = sum_{kz,ky,kx,c} in[z+kz][y+ky][x+kx][c] w[n][c][kz][ky][kx]   (reference models/3d_affs/unet.py:26-34, a
// torch.nn.Conv3d) is computed per 2 x 2 in-plane output tile as  Y = A^T [ sum_{kz,c} (G g G^T) . (B^T d B) ] A  with the
// 4 x 4 input tile d and the kernel plane g = w[n][c][kz]: 16 multiplies per 4 outputs and (kz, c) instead of 36 -- 12 per
// output instead of 27.  The 16 element-wise products over (kz, c) are 16 independent GEMMs
//     M[b][m][n] = sum_{kz,c} V[b][z(m)+kz][ty(m)][tx(m)][c] U[b][kz][c][n],      b = 4 xi + nu,
// i.e. a (3,1,1) convolution of the transformed tensor V[b]: they run as ONE batched launch of the fused split-bf16
// implicit-GEMM kernel (conv_igemm.hip, ConvArgs::nbatch) with raw f32 sums as its output.  This file holds the two
// memory-bound transforms around it and the host-side weight transform.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// Arithmetic: d = hi + lo is exact in f32; B^T d B is sums of four such values (f32 rounding, 2^-24); V and U are stored as
// (hi, lo) bf16 pairs like every operand of this mode (2^-17 per product); M and A^T M A are f32.  Measured against the direct
// form on the full-size block: DESIGN.md section 4.
#include "wino.h"

#include <cstring>

namespace bsmi {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ uint16_t to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ void unpack8(u32x4_t v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)to_bf16(f[2 * i]) | ((uint32_t)to_bf16(f[2 * i + 1]) << 16);
  u32x4_t v = {w[0], w[1], w[2], w[3]};
  return v;
}
// 8 channels at element index e (a multiple of 8) of the plain [voxel][Cpad] order of a split tensor: the 16-byte vector of
// hi values is followed by that of the lo values (conv_dev.h act_index)
__device__ __forceinline__ void load_split8(const uint16_t* base, size_t e, float* f) {
  const uint16_t* p = base + 2 * e;
  float g[8];
  unpack8(*(const u32x4_t*)p, f);
  unpack8(*(const u32x4_t*)(p + 8), g);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] += g[k];
}
template <bool NT>
__device__ __forceinline__ void store_split8(uint16_t* base, size_t e, const float* f) {
  uint16_t* p = base + 2 * e;
  const u32x4_t hv = pack8(f);
  float h[8], r[8];
  unpack8(hv, h);
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = f[k] - h[k];
  const u32x4_t lv = pack8(r);
  if constexpr (NT) {
    __builtin_nontemporal_store(hv, (u32x4_t*)p);
    __builtin_nontemporal_store(lv, (u32x4_t*)(p + 8));
  } else {
    *(u32x4_t*)p = hv;
    *(u32x4_t*)(p + 8) = lv;
  }
}

// source index pair and weights of one output index of torch's linear upsampling (align_corners = False); the arithmetic of
// unet_ops.hip lin_src, operation for operation
__device__ __forceinline__ void lin_src(int dst, int f, int n, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)dst + 0.5f) * (1.0f / (float)f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < n - 1 ? i0 : n - 1;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = src - (float)i0;
  w1 = w1 < 0.f ? 0.f : (w1 > 1.f ? 1.f : w1);
  w0 = 1.f - w1;
}

// One thread: 8 channels of source `src` along one row of tiles (z, ty), tx = 0 .. Tx - 1.  Neighbouring tiles share two of
// their four input columns: the thread keeps the row-transformed columns (B^T d, a per-column operation) of the previous tile
// and loads two new columns per tile -- half the loads and half the row arithmetic of a tile-per-thread form.  Blocks are
// numbered XCD-contiguously (consecutive block ids go to different XCDs): the tile rows ty and ty + 1, which share two of
// their four input rows, then run on the same XCD and meet in its L2 instead of fetching those rows twice over the fabric.
__global__ __launch_bounds__(256) void wino_in_kernel(const WinoInArgs a, int src, size_t total) {
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  const size_t i = (size_t)blk * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncv = a.Cpad[src] >> 3;
  const int cv = (int)(i % ncv);
  const size_t t = i / ncv;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const uint16_t* sp = (const uint16_t*)a.src[src];
  const int H = a.H[src], W = a.W[src], C = a.Cpad[src];
  // element index of input voxel (row 0 of the tile row, column 0) of this thread's channels
  const size_t e_in = (((size_t)(z + a.oz[src]) * H + (2 * ty + a.oy[src])) * W + a.ox[src]) * C + 8 * cv;
  const size_t rstride = (size_t)W * C;
  uint16_t* V = (uint16_t*)a.V;
  const size_t plane = (size_t)a.Dv * a.Ty * a.Tx * a.Cv;  // elements of one batch of V
  const size_t e_out = (((size_t)z * a.Ty + ty) * a.Tx) * a.Cv + a.cv0[src] + 8 * cv;
  float col[4][4][8];  // col[c][xi]: (B^T d)[xi] of input column 2 tx + c
  auto load_col = [&](int c, int x) __attribute__((always_inline)) {
    float d[4][8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) load_split8(sp, e_in + rr * rstride + (size_t)x * C, d[rr]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      col[c][0][k] = d[0][k] - d[2][k];
      col[c][1][k] = d[1][k] + d[2][k];
      col[c][2][k] = d[2][k] - d[1][k];
      col[c][3][k] = d[1][k] - d[3][k];
    }
  };
  load_col(0, 0);
  load_col(1, 1);
  for (int tx = 0; tx < a.Tx; ++tx) {
    load_col(2, 2 * tx + 2);
    load_col(3, 2 * tx + 3);
    const size_t e0 = e_out + (size_t)tx * a.Cv;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      float v[4][8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[0][k] = col[0][xi][k] - col[2][xi][k];
        v[1][k] = col[1][xi][k] + col[2][xi][k];
        v[2][k] = col[2][xi][k] - col[1][xi][k];
        v[3][k] = col[1][xi][k] - col[3][xi][k];
      }
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) store_split8<false>(V, (size_t)(4 * xi + nu) * plane + e0, v[nu]);
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        col[0][xi][k] = col[2][xi][k];
        col[1][xi][k] = col[3][xi][k];
      }
  }
}

// The input transform of a source that is read through a factor-2 in-plane upsampling (WinoInArgs::upf = 2).  A 4 x 4 tile of
// the upsampled map interpolates a 4 x 4 window L of the low-resolution tensor, d = Uy L Ux^T, so the transform is
// V = (B^T Uy) L (B^T Ux)^T with constant 4 x 4 matrices that depend only on the parity of the tile's first upsampled index:
//   even (rows 2a .. 2a+3 from low rows a-1 .. a+2):  Uy = [.25 .75 0 0; 0 .75 .25 0; 0 .25 .75 0; 0 0 .75 .25]
//   odd  (rows 2a+1 .. 2a+4 from low rows a .. a+3):   Uy = [.75 .25 0 0; .25 .75 0 0; 0 .75 .25 0; 0 .25 .75 0]
// (torch's weights for scale 2, align_corners = False; at the faces the window repeats the outermost row / column, which is
// what the clamped source index of the reference amounts to, to within the rounding of .25 x + .75 x against x).  All
// coefficients are exact binary fractions.  The thread walks a row of tiles: the window moves by ONE low-resolution column per
// tile, so a tile costs four loads -- half of what the plain transform needs -- and no upsampled map exists anywhere.
template <int PAR>
struct UpT {  // T = B^T U_PAR, row xi over the window's four entries
  static constexpr float v[4][4] = {
      {PAR ? 0.75f : 0.25f, PAR ? -0.5f : 0.5f, PAR ? -0.25f : -0.75f, 0.f},   // U0 - U2
      {PAR ? 0.25f : 0.f, PAR ? 1.5f : 1.f, PAR ? 0.25f : 1.f, 0.f},            // U1 + U2
      {PAR ? -0.25f : 0.f, PAR ? 0.f : -0.5f, PAR ? 0.25f : 0.5f, 0.f},         // U2 - U1
      {PAR ? 0.25f : 0.f, PAR ? 0.5f : 0.75f, PAR ? -0.75f : -0.5f, PAR ? 0.f : -0.25f}};  // U1 - U3
};
template <int PY, int PX>
__global__ __launch_bounds__(256) void wino_in_up_kernel(const WinoInArgs a, int src, size_t total) {
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  const size_t i = (size_t)blk * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncv = a.Cpad[src] >> 3;
  const int cv = (int)(i % ncv);
  const size_t t = i / ncv;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const uint16_t* sp = (const uint16_t*)a.src[src];
  const int H = a.H[src], W = a.W[src], C = a.Cpad[src];
  const size_t e_low = ((size_t)(z + a.oz[src]) * H) * W * C + 8 * cv;
  // low-resolution rows of the window (clamped) and the column of its first entry for tile 0
  const int ay = ((2 * ty + a.oy[src]) >> 1) - 1 + PY;
  int wy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = ay + k;
    wy[k] = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
  }
  const int bx = (a.ox[src] >> 1) - 1 + PX;
  uint16_t* V = (uint16_t*)a.V;
  const size_t plane = (size_t)a.Dv * a.Ty * a.Tx * a.Cv;
  const size_t e_out = (((size_t)z * a.Ty + ty) * a.Tx) * a.Cv + a.cv0[src] + 8 * cv;
  float ct[4][4][8];  // ct[c][xi] = sum_k Ty[xi][k] L[k][window column c]
  auto load_ct = [&](int c, int xlow) __attribute__((always_inline)) {
    const int x = xlow < 0 ? 0 : (xlow > W - 1 ? W - 1 : xlow);
    float L[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) load_split8(sp, e_low + ((size_t)wy[k] * W + x) * C, L[k]);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (UpT<PY>::v[xi][k] != 0.f) acc += UpT<PY>::v[xi][k] * L[k][ch];
        ct[c][xi][ch] = acc;
      }
  };
  load_ct(0, bx);
  load_ct(1, bx + 1);
  load_ct(2, bx + 2);
  for (int tx = 0; tx < a.Tx; ++tx) {
    load_ct(3, bx + tx + 3);
    const size_t e0 = e_out + (size_t)tx * a.Cv;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        float v[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (UpT<PX>::v[nu][c] != 0.f) acc += UpT<PX>::v[nu][c] * ct[c][xi][ch];
          v[ch] = acc;
        }
        store_split8<false>(V, (size_t)(4 * xi + nu) * plane + e0, v);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) ct[c][xi][ch] = ct[c + 1][xi][ch];
  }
}

// one thread: the 2 x 2 output tile of 8 channels at (z, ty, tx)
template <bool LOW>
__global__ __launch_bounds__(256) void wino_out_kernel(const WinoOutArgs a, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncv = a.Co >> 3;
  const int cv = (int)(i % ncv);
  size_t t = i / ncv;
  const int tx = (int)(t % a.Tx);
  t /= a.Tx;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const size_t Mrows = (size_t)a.Do * a.Ty * a.Tx;
  const size_t m = ((size_t)z * a.Ty + ty) * a.Tx + tx;
  float y[2][2][8];
  // A^T M A = sum_b c_p(xi) c_q(nu) M[b]:  c_0 = (1, 1, 1, 0), c_1 = (0, 1, -1, -1)
  float tp[2][4][8];  // A^T M: rows
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    float mm[4][8];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const float* p = a.M + ((size_t)(4 * xi + nu) * Mrows + m) * a.Co + 8 * cv;
      const f32x4_t v0 = __builtin_nontemporal_load((const f32x4_t*)p);
      const f32x4_t v1 = __builtin_nontemporal_load((const f32x4_t*)(p + 4));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm[xi][k] = v0[k];
        mm[xi][4 + k] = v1[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tp[0][nu][k] = mm[0][k] + mm[1][k] + mm[2][k];
      tp[1][nu][k] = mm[1][k] - mm[2][k] - mm[3][k];
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      y[p][0][k] = tp[p][0][k] + tp[p][1][k] + tp[p][2][k];
      y[p][1][k] = tp[p][1][k] - tp[p][2][k] - tp[p][3][k];
    }
  float bv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bv[k] = a.bias[8 * cv + k];
  const int Ho = 2 * a.Ty, Wo = 2 * a.Tx;
  uint16_t* out = (uint16_t*)a.out;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const size_t e = (((size_t)z * Ho + (2 * ty + p)) * Wo + (2 * tx + q)) * a.Co + 8 * cv;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = y[p][q][k] + bv[k];
      if (a.addend) {
        const f32x4_t r0 = __builtin_nontemporal_load((const f32x4_t*)(a.addend + e));
        const f32x4_t r1 = __builtin_nontemporal_load((const f32x4_t*)(a.addend + e + 4));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] += r0[k];
          v[4 + k] += r1[k];
        }
      }
      if constexpr (LOW) {  // the residual branch's sums over the upsampled map, interpolated from their low-resolution form
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        lin_src(2 * ty + p + a.loy, a.lf, a.lH, y0, y1, wy0, wy1);
        lin_src(2 * tx + q + a.lox, a.lf, a.lW, x0, x1, wx0, wx1);
        const float* base = a.low + ((size_t)(z + a.loz) * a.lH * a.lW) * a.Co + 8 * cv;
        float r[4][8];
        const size_t o[4] = {((size_t)y0 * a.lW + x0) * a.Co, ((size_t)y0 * a.lW + x1) * a.Co, ((size_t)y1 * a.lW + x0) * a.Co,
                             ((size_t)y1 * a.lW + x1) * a.Co};
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const f32x4_t r0 = *(const f32x4_t*)(base + o[t4]);
          const f32x4_t r1 = *(const f32x4_t*)(base + o[t4] + 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            r[t4][k] = r0[k];
            r[t4][4 + k] = r1[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float pa = wx0 * r[0][k] + wx1 * r[1][k];
          const float pb = wx0 * r[2][k] + wx1 * r[3][k];
          v[k] += wy0 * pa + wy1 * pb;
        }
      }
      if (a.relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      }
      store_split8<false>(out, e, v);
    }
}

}  // namespace

int launch_wino_in(const WinoInArgs& a, hipStream_t s) {
  if (a.nsrc < 1 || a.nsrc > kWinoMaxSrc || a.Dv <= 0 || a.Ty <= 0 || a.Tx <= 0 || a.Cv % 8) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: bad geometry");
  for (int q = 0; q < a.nsrc; ++q) {
    if (a.Cpad[q] % 8 || a.cv0[q] % 8 || a.cv0[q] + a.Cpad[q] > a.Cv) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: bad channel layout of source %d", q);
    const int upf = a.upf[q] > 0 ? a.upf[q] : 1;  // an upsampled source is upf times as large in the plane
    if (a.oy[q] + 2 * a.Ty + 2 > a.H[q] * upf || a.ox[q] + 2 * a.Tx + 2 > a.W[q] * upf) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: tiles leave source %d", q);
    const size_t total = (size_t)a.Dv * a.Ty * (a.Cpad[q] / 8);  // one thread per (z, tile row, 8 channels)
    const dim3 grid((unsigned)((total + 255) / 256));
    if (a.upf[q] > 0) {
      if (a.upf[q] != 2 || a.oy[q] < 0 || a.ox[q] < 0) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: only a factor-2 upsampling can be fused");
      const int par = 2 * (a.oy[q] & 1) + (a.ox[q] & 1);
      if (par == 0) hipLaunchKernelGGL((wino_in_up_kernel<0, 0>), grid, dim3(256), 0, s, a, q, total);
      else if (par == 1) hipLaunchKernelGGL((wino_in_up_kernel<0, 1>), grid, dim3(256), 0, s, a, q, total);
      else if (par == 2) hipLaunchKernelGGL((wino_in_up_kernel<1, 0>), grid, dim3(256), 0, s, a, q, total);
      else hipLaunchKernelGGL((wino_in_up_kernel<1, 1>), grid, dim3(256), 0, s, a, q, total);
    } else {
      hipLaunchKernelGGL(wino_in_kernel, grid, dim3(256), 0, s, a, q, total);
    }
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int launch_wino_out(const WinoOutArgs& a, hipStream_t s) {
  if (a.Do <= 0 || a.Ty <= 0 || a.Tx <= 0 || a.Co % 8) BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: bad geometry");
  if (a.low && (a.lf < 1 || a.loz < 0 || a.loz + a.Do > a.lD || a.loy < 0 || a.loy + 2 * a.Ty > a.lH * a.lf || a.lox < 0 || a.lox + 2 * a.Tx > a.lW * a.lf))
    BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: the low-resolution residual does not cover the output");
  const size_t total = (size_t)a.Do * a.Ty * a.Tx * (a.Co / 8);
  if (a.low) hipLaunchKernelGGL(wino_out_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, total);
  else hipLaunchKernelGGL(wino_out_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, total);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

void wino_units(int Cv, std::vector<WinoUnit>& out) {
  out.clear();
  for (int c32 = 0; c32 < Cv; c32 += 32)
    for (int kz = 0; kz < 3; ++kz)
      for (int j = 0; j < kUnitsPerStep; ++j) {
        const int c = c32 + 16 * j;
        if (c < Cv) out.push_back(WinoUnit{kz, c, false});
        else out.push_back(WinoUnit{0, 0, true});
      }
  if ((out.size() / kUnitsPerStep) % 2)  // the 16x16x32 kernel body walks K-steps in pairs
    for (int j = 0; j < kUnitsPerStep; ++j) out.push_back(WinoUnit{0, 0, true});
}

static inline uint16_t host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float host_bf16_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

void wino_pack_weights(const float* w, int cout, int cin, const std::vector<int>& cin_of_v, int Npad, const std::vector<WinoUnit>& units,
                       std::vector<uint16_t>& packed, size_t& image_elems, size_t& batch_elems) {
  const size_t nsteps = units.size() / kUnitsPerStep;
  batch_elems = nsteps * (size_t)Npad * 32;
  image_elems = kWinoBatch * batch_elems + (size_t)kWeightRowSlack * 32;
  packed.assign(2 * image_elems, 0);
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  host_parallel_for((size_t)cout, [&](size_t n_) {   // one output channel per task: its weights read once, its image rows its own lines
    const int n = (int)n_;
    for (size_t u = 0; u < units.size(); ++u) {
      const WinoUnit& un = units[u];
      if (un.dummy) continue;
      const size_t s = u / kUnitsPerStep, j = u % kUnitsPerStep;
      for (int kk = 0; kk < 16; ++kk) {
        const int c = cin_of_v[un.vc0 + kk];
        if (c < 0) continue;
        const float* g = w + (((size_t)n * cin + c) * 3 + un.kz) * 9;  // [ky][kx]
        double t[4][3];
        for (int xi = 0; xi < 4; ++xi)
          for (int kx = 0; kx < 3; ++kx) t[xi][kx] = G[xi][0] * g[kx] + G[xi][1] * g[3 + kx] + G[xi][2] * g[6 + kx];
        for (int xi = 0; xi < 4; ++xi)
          for (int nu = 0; nu < 4; ++nu) {
            const float v = (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
            const size_t idx = (size_t)(4 * xi + nu) * batch_elems + (s * Npad + n) * 32 + j * 16 + kk;
            const uint16_t hi = host_bf16(v);
            packed[idx] = hi;
            packed[image_elems + idx] = host_bf16(v - host_bf16_f32(hi));
          }
      }
    }
  });
}

}  // namespace bsmi
