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

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

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


// ---- F(4x4, 3x3) (round 4) ----------------------------------------------------------------------------------------------------
// 6 x 6 input tile -> 4 x 4 outputs: 36 products per 16 outputs and (kz, c), 2.25 per output where F(2x2) spends 4 (and V holds
// 2.25 values per input voxel instead of 4, M 2.25 sums per output instead of 4).  Interpolation points 0, +-1/sqrt2, +-sqrt2, inf
// instead of the textbook 0, +-1, +-2, inf: in this mode the operands V and U are (hi, lo) bf16 pairs, 2^-17 each, and the
// transforms amplify exactly that rounding -- measured on one 300- and one 1500-channel layer (CPU emulation of the arithmetic,
// DESIGN.md section 4): largest error relative to the layer's largest output 6e-6 for F(2x2), 8-10e-5 for the textbook points,
// 2.4e-5 for these (a b = 1 keeps the powers of the points that make up G, B^T and A^T within [1/4, 4]).
//   B^T = [1 0 -5/2 0 1 0;  0 r 2 -r/2 -1 0;  0 -r 2 r/2 -1 0;  0 -r -1 2r 2 0;  0 r -1 -2r 2 0;  0 1 0 -5/2 0 1]          r = sqrt2
//   G   = [1 0 0;  2/3 (1, a, 1/2);  2/3 (1, -a, 1/2);  1/12 (1, b, 2);  1/12 (1, -b, 2);  0 0 1]                      a = 1/r, b = r
//   A^T = [1 1 1 1 1 0;  0 a -a b -b 0;  0 1/2 1/2 2 2 0;  0 a^3 -a^3 b^3 -b^3 1]
// A thread takes FOUR channels (half of a 16-byte vector: 8 bytes of hi values, 8 of lo values): the 36 intermediate values of
// a tile are 144 registers at four channels.
constexpr float kR2 = 1.41421356237309504880f;
__host__ __device__ constexpr float bt4(int i, int j) {
  constexpr float B[6][6] = {{1.f, 0.f, -2.5f, 0.f, 1.f, 0.f},
                             {0.f, kR2, 2.f, -0.5f * kR2, -1.f, 0.f},
                             {0.f, -kR2, 2.f, 0.5f * kR2, -1.f, 0.f},
                             {0.f, -kR2, -1.f, 2.f * kR2, 2.f, 0.f},
                             {0.f, kR2, -1.f, -2.f * kR2, 2.f, 0.f},
                             {0.f, 1.f, 0.f, -2.5f, 0.f, 1.f}};
  return B[i][j];
}
// r = B^T d for one column of six values, FOUR channels each
__device__ __forceinline__ void bt4_apply(const float (*d)[4], float (*r)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float e = 2.f * d[2][k] - d[4][k], o = kR2 * (d[1][k] - 0.5f * d[3][k]);
    const float e2 = 2.f * d[4][k] - d[2][k], o2 = kR2 * (2.f * d[3][k] - d[1][k]);
    r[0][k] = d[0][k] - 2.5f * d[2][k] + d[4][k];
    r[1][k] = e + o;
    r[2][k] = e - o;
    r[3][k] = e2 + o2;
    r[4][k] = e2 - o2;
    r[5][k] = d[1][k] - 2.5f * d[3][k] + d[5][k];
  }
}
// y = A^T m for six values, four channels each
__device__ __forceinline__ void at4_apply(const float (*m)[4], float (*y)[4]) {
  constexpr float a = 0.70710678118654752440f, b = kR2, a3 = 0.35355339059327376220f, b3 = 2.82842712474619009760f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float s12 = m[1][k] + m[2][k], d12 = m[1][k] - m[2][k], s34 = m[3][k] + m[4][k], d34 = m[3][k] - m[4][k];
    y[0][k] = m[0][k] + s12 + s34;
    y[1][k] = a * d12 + b * d34;
    y[2][k] = 0.5f * s12 + 2.f * s34;
    y[3][k] = a3 * d12 + b3 * d34 + m[5][k];
  }
}
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// 4 channels (half `hf` of the 8-channel group at element index e) of a split tensor
__device__ __forceinline__ void load_split4(const uint16_t* base, size_t e, int hf, float* f) {
  const uint16_t* p = base + 2 * e + 4 * hf;
  const u32x2_t h = *(const u32x2_t*)p, l = *(const u32x2_t*)(p + 8);
  f[0] = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
  f[1] = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
  f[2] = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
  f[3] = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
}
__device__ __forceinline__ void store_split4(uint16_t* base, size_t e, int hf, const float* f) {
  uint16_t* p = base + 2 * e + 4 * hf;
  uint16_t hb[4], lb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hb[k] = to_bf16(f[k]);
    lb[k] = to_bf16(f[k] - __uint_as_float((uint32_t)hb[k] << 16));
  }
  const u32x2_t hv = {(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)};
  const u32x2_t lv = {(uint32_t)lb[0] | ((uint32_t)lb[1] << 16), (uint32_t)lb[2] | ((uint32_t)lb[3] << 16)};
  *(u32x2_t*)p = hv;
  *(u32x2_t*)(p + 8) = lv;
}

// One thread: 4 channels along one row of tiles (z, ty); neighbouring tiles share two of their six input columns, whose
// row transforms are kept.  Rows / columns past the source (overhanging last tiles) are clamped: whatever they hold only
// reaches outputs that are never stored.
__global__ __launch_bounds__(256) void wino4_in_kernel(const WinoInArgs a, int src, size_t total) {
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  const size_t i = (size_t)blk * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncg = a.Cpad[src] >> 2;
  const int cg = (int)(i % ncg), cv = cg >> 1, hf = cg & 1;
  const size_t t = i / ncg;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const uint16_t* sp = (const uint16_t*)a.src[src];
  const int H = a.H[src], W = a.W[src], C = a.Cpad[src];
  size_t erow[6];
#pragma unroll
  for (int rr = 0; rr < 6; ++rr) {
    int y = 4 * ty + a.oy[src] + rr;
    y = y < H - 1 ? y : H - 1;
    erow[rr] = (((size_t)(z + a.oz[src]) * H + y) * W) * C + 8 * cv;
  }
  uint16_t* V = (uint16_t*)a.V;
  const size_t plane = (size_t)a.Dv * a.Ty * a.Tx * a.Cv;
  const size_t e_out = (((size_t)z * a.Ty + ty) * a.Tx) * a.Cv + a.cv0[src] + 8 * cv;
  float col[6][6][4];  // col[c][xi]: (B^T d)[xi] of input column 4 tx + c
  auto load_col = [&](int c, int x) __attribute__((always_inline)) {
    x += a.ox[src];
    x = x < W - 1 ? x : W - 1;
    float d[6][4];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) load_split4(sp, erow[rr] + (size_t)x * C, hf, d[rr]);
    bt4_apply(d, col[c]);
  };
  load_col(0, 0);
  load_col(1, 1);
  for (int tx = 0; tx < a.Tx; ++tx) {
#pragma unroll
    for (int c = 2; c < 6; ++c) load_col(c, 4 * tx + c);
    const size_t e0 = e_out + (size_t)tx * a.Cv;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
      float in[6][4], v[6][4];
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) in[c][k] = col[c][xi][k];
      bt4_apply(in, v);
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) store_split4(V, (size_t)(6 * xi + nu) * plane + e0, hf, v[nu]);
    }
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        col[0][xi][k] = col[4][xi][k];
        col[1][xi][k] = col[5][xi][k];
      }
  }
}

// The same through a factor-2 in-plane upsampling (see wino_in_up_kernel): the six upsampled rows of a tile interpolate FIVE
// low-resolution rows (four when the tile starts at an odd row), d = Uy L Ux^T, V = (B^T Uy) L (B^T Ux)^T; the window moves by
// two low-resolution columns per tile.
//   even start 2a   (low rows a-1 .. a+3):  Uy = [.25 .75 0 0 0; 0 .75 .25 0 0; 0 .25 .75 0 0; 0 0 .75 .25 0; 0 0 .25 .75 0; 0 0 0 .75 .25]
//   odd  start 2a+1 (low rows a .. a+4):    Uy = [.75 .25 0 0 0; .25 .75 0 0 0; 0 .75 .25 0 0; 0 .25 .75 0 0; 0 0 .75 .25 0; 0 0 .25 .75 0]
__host__ __device__ constexpr float up4(int par, int row, int k) {
  // entry (row, k) of U_par: upsampled row `row` of the tile blends window rows f and f + 1
  //   even start: (0,1) (1,2) (1,2) (2,3) (2,3) (3,4), the first with weight .25 on even rows, .75 on odd ones
  //   odd start:  (0,1) (0,1) (1,2) (1,2) (2,3) (2,3), the first with weight .75 on even rows, .25 on odd ones
  const int f = par ? row / 2 : (row + 1) / 2;
  const bool first_heavy = par ? (row & 1) == 0 : (row & 1) == 1;
  return k == f ? (first_heavy ? 0.75f : 0.25f) : (k == f + 1 ? (first_heavy ? 0.25f : 0.75f) : 0.f);
}
__host__ __device__ constexpr float tu4(int par, int xi, int k) {  // (B^T U_par)[xi][k]
  float s = 0.f;
  for (int r = 0; r < 6; ++r) s += bt4(xi, r) * up4(par, r, k);
  return s;
}
template <int PY, int PX>
__global__ __launch_bounds__(256) void wino4_in_up_kernel(const WinoInArgs a, int src, size_t total) {
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  const size_t i = (size_t)blk * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncg = a.Cpad[src] >> 2;
  const int cg = (int)(i % ncg), cv = cg >> 1, hf = cg & 1;
  const size_t t = i / ncg;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const uint16_t* sp = (const uint16_t*)a.src[src];
  const int H = a.H[src], W = a.W[src], C = a.Cpad[src];
  const size_t e_low = ((size_t)(z + a.oz[src]) * H) * W * C + 8 * cv;
  const int ay = ((4 * ty + a.oy[src]) >> 1) - 1 + PY;
  int wy[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int y = ay + k;
    wy[k] = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
  }
  const int bx = (a.ox[src] >> 1) - 1 + PX;
  uint16_t* V = (uint16_t*)a.V;
  const size_t plane = (size_t)a.Dv * a.Ty * a.Tx * a.Cv;
  const size_t e_out = (((size_t)z * a.Ty + ty) * a.Tx) * a.Cv + a.cv0[src] + 8 * cv;
  float ct[5][6][4];  // ct[c][xi] = sum_k (B^T Uy)[xi][k] L[k][window column c]
  auto load_ct = [&](int c, int xlow) __attribute__((always_inline)) {
    const int x = xlow < 0 ? 0 : (xlow > W - 1 ? W - 1 : xlow);
    float L[5][4];
#pragma unroll
    for (int k = 0; k < 5; ++k) load_split4(sp, e_low + ((size_t)wy[k] * W + x) * C, hf, L[k]);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (tu4(PY, xi, k) != 0.f) acc += tu4(PY, xi, k) * L[k][ch];
        ct[c][xi][ch] = acc;
      }
  };
  load_ct(0, bx);
  load_ct(1, bx + 1);
  load_ct(2, bx + 2);
  for (int tx = 0; tx < a.Tx; ++tx) {
    load_ct(3, bx + 2 * tx + 3);
    load_ct(4, bx + 2 * tx + 4);
    const size_t e0 = e_out + (size_t)tx * a.Cv;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        float v[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 5; ++c)
            if (tu4(PX, nu, c) != 0.f) acc += tu4(PX, nu, c) * ct[c][xi][ch];
          v[ch] = acc;
        }
        store_split4(V, (size_t)(6 * xi + nu) * plane + e0, hf, v);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int xi = 0; xi < 6; ++xi)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) ct[c][xi][ch] = ct[c + 2][xi][ch];
  }
}

// one thread: the 4 x 4 output tile of 4 channels at (z, ty, tx); outputs past (Ho, Wo) -- overhanging tiles -- are not stored
template <bool LOW>
__global__ __launch_bounds__(256) void wino4_out_kernel(const WinoOutArgs a, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ncg = a.Co >> 2;
  const int cg = (int)(i % ncg), cv = cg >> 1, hf = cg & 1;
  size_t t = i / ncg;
  const int tx = (int)(t % a.Tx);
  t /= a.Tx;
  const int ty = (int)(t % a.Ty);
  const int z = (int)(t / a.Ty);
  const size_t Mrows = (size_t)a.Do * a.Ty * a.Tx;
  const size_t m = ((size_t)z * a.Ty + ty) * a.Tx + tx;
  float tp[6][4][4];  // tp[nu][p]: (A^T M)[p][nu]
#pragma unroll
  for (int nu = 0; nu < 6; ++nu) {
    float mm[6][4];
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
      const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)(a.M + ((size_t)(6 * xi + nu) * Mrows + m) * a.Co + 4 * cg));
#pragma unroll
      for (int k = 0; k < 4; ++k) mm[xi][k] = v[k];
    }
    at4_apply(mm, tp[nu]);
  }
  float bv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bv[k] = a.bias[4 * cg + k];
  uint16_t* out = (uint16_t*)a.out;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float in[6][4], y[4][4];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
      for (int k = 0; k < 4; ++k) in[nu][k] = tp[nu][p][k];
    at4_apply(in, y);
    const int oy = 4 * ty + p;
    if (oy >= a.Ho) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = 4 * tx + q;
      if (ox >= a.Wo) continue;
      const size_t e8 = (((size_t)z * a.Ho + oy) * a.Wo + ox) * a.Co + 8 * cv;
      const size_t e = e8 + 4 * hf;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = y[q][k] + bv[k];
      if (a.addend) {
        const f32x4_t r0 = __builtin_nontemporal_load((const f32x4_t*)(a.addend + e));
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += r0[k];
      }
      if constexpr (LOW) {
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        lin_src(oy + a.loy, a.lf, a.lH, y0, y1, wy0, wy1);
        lin_src(ox + a.lox, a.lf, a.lW, x0, x1, wx0, wx1);
        const float* base = a.low + ((size_t)(z + a.loz) * a.lH * a.lW) * a.Co + 4 * cg;
        const f32x4_t r00 = *(const f32x4_t*)(base + ((size_t)y0 * a.lW + x0) * a.Co), r01 = *(const f32x4_t*)(base + ((size_t)y0 * a.lW + x1) * a.Co);
        const f32x4_t r10 = *(const f32x4_t*)(base + ((size_t)y1 * a.lW + x0) * a.Co), r11 = *(const f32x4_t*)(base + ((size_t)y1 * a.lW + x1) * a.Co);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float pa = wx0 * r00[k] + wx1 * r01[k];
          const float pb = wx0 * r10[k] + wx1 * r11[k];
          v[k] += wy0 * pa + wy1 * pb;
        }
      }
      if (a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      }
      store_split4(out, e8, hf, v);
    }
  }
}

}  // namespace

int launch_wino_in(const WinoInArgs& a, hipStream_t s) {
  const int m = a.m == 4 ? 4 : 2;
  if (a.m != 0 && a.m != 2 && a.m != 4) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: tile edge %d (2 or 4)", a.m);
  if (a.nsrc < 1 || a.nsrc > kWinoMaxSrc || a.Dv <= 0 || a.Ty <= 0 || a.Tx <= 0 || a.Cv % 8) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: bad geometry");
  for (int q = 0; q < a.nsrc; ++q) {
    if (a.Cpad[q] % 8 || a.cv0[q] % 8 || a.cv0[q] + a.Cpad[q] > a.Cv) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: bad channel layout of source %d", q);
    const int upf = a.upf[q] > 0 ? a.upf[q] : 1;  // an upsampled source is upf times as large in the plane
    if (m == 2 && (a.oy[q] + 2 * a.Ty + 2 > a.H[q] * upf || a.ox[q] + 2 * a.Tx + 2 > a.W[q] * upf)) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: tiles leave source %d", q);
    // F(4x4): the last tile row / column may overhang by up to three voxels (clamped reads); more than that is a planning error
    if (m == 4 && (a.oy[q] < 0 || a.ox[q] < 0 || a.oy[q] + 4 * a.Ty + 2 > a.H[q] * upf + 3 || a.ox[q] + 4 * a.Tx + 2 > a.W[q] * upf + 3))
      BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: F(4x4) tiles leave source %d by more than a tile's overhang", q);
    const size_t total = (size_t)a.Dv * a.Ty * (a.Cpad[q] / (m == 4 ? 4 : 8));  // one thread per (z, tile row, 8 resp. 4 channels)
    const dim3 grid((unsigned)((total + 255) / 256));
    if (a.upf[q] > 0) {
      if (a.upf[q] != 2 || a.oy[q] < 0 || a.ox[q] < 0) BSMI_FAIL(BSMI_ERR_INVALID, "winograd input transform: only a factor-2 upsampling can be fused");
      const int par = 2 * (a.oy[q] & 1) + (a.ox[q] & 1);
      if (m == 4) {
        if (par == 0) hipLaunchKernelGGL((wino4_in_up_kernel<0, 0>), grid, dim3(256), 0, s, a, q, total);
        else if (par == 1) hipLaunchKernelGGL((wino4_in_up_kernel<0, 1>), grid, dim3(256), 0, s, a, q, total);
        else if (par == 2) hipLaunchKernelGGL((wino4_in_up_kernel<1, 0>), grid, dim3(256), 0, s, a, q, total);
        else hipLaunchKernelGGL((wino4_in_up_kernel<1, 1>), grid, dim3(256), 0, s, a, q, total);
      } else if (par == 0) hipLaunchKernelGGL((wino_in_up_kernel<0, 0>), grid, dim3(256), 0, s, a, q, total);
      else if (par == 1) hipLaunchKernelGGL((wino_in_up_kernel<0, 1>), grid, dim3(256), 0, s, a, q, total);
      else if (par == 2) hipLaunchKernelGGL((wino_in_up_kernel<1, 0>), grid, dim3(256), 0, s, a, q, total);
      else hipLaunchKernelGGL((wino_in_up_kernel<1, 1>), grid, dim3(256), 0, s, a, q, total);
    } else if (m == 4) {
      hipLaunchKernelGGL(wino4_in_kernel, grid, dim3(256), 0, s, a, q, total);
    } else {
      hipLaunchKernelGGL(wino_in_kernel, grid, dim3(256), 0, s, a, q, total);
    }
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int launch_wino_out(const WinoOutArgs& a, hipStream_t s) {
  const int m = a.m == 4 ? 4 : 2;
  if (a.m != 0 && a.m != 2 && a.m != 4) BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: tile edge %d (2 or 4)", a.m);
  if (a.Do <= 0 || a.Ty <= 0 || a.Tx <= 0 || a.Co % 8) BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: bad geometry");
  const int Ho = m == 4 ? a.Ho : 2 * a.Ty, Wo = m == 4 ? a.Wo : 2 * a.Tx;
  if (m == 4 && (Ho < 1 || Wo < 1 || Ho > 4 * a.Ty || Wo > 4 * a.Tx || Ho + 3 < 4 * a.Ty || Wo + 3 < 4 * a.Tx)) BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: output extent and F(4x4) tile counts disagree");
  if (a.low && (a.lf < 1 || a.loz < 0 || a.loz + a.Do > a.lD || a.loy < 0 || a.loy + Ho > a.lH * a.lf || a.lox < 0 || a.lox + Wo > a.lW * a.lf))
    BSMI_FAIL(BSMI_ERR_INVALID, "winograd output transform: the low-resolution residual does not cover the output");
  const size_t total = (size_t)a.Do * a.Ty * a.Tx * (a.Co / (m == 4 ? 4 : 8));
  const dim3 grid((unsigned)((total + 255) / 256));
  if (m == 4) {
    if (a.low) hipLaunchKernelGGL(wino4_out_kernel<true>, grid, dim3(256), 0, s, a, total);
    else hipLaunchKernelGGL(wino4_out_kernel<false>, grid, dim3(256), 0, s, a, total);
  } else if (a.low) hipLaunchKernelGGL(wino_out_kernel<true>, grid, dim3(256), 0, s, a, total);
  else hipLaunchKernelGGL(wino_out_kernel<false>, grid, dim3(256), 0, s, a, total);
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
                       std::vector<uint16_t>& packed, size_t& image_elems, size_t& batch_elems, int m) {
  const size_t nsteps = units.size() / kUnitsPerStep;
  const int T = m + 2, nbatch = T * T;
  batch_elems = nsteps * (size_t)Npad * 32;
  image_elems = nbatch * batch_elems + (size_t)kWeightRowSlack * 32;
  packed.assign(2 * image_elems, 0);
  static const double G2[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  // F(4x4), points 0, +-1/sqrt2, +-sqrt2, inf; the rows carry the scales taken out of B^T (see the kernels above)
  const double a = 0.70710678118654752440, b = 1.41421356237309504880;
  const double G4[6][3] = {{1, 0, 0}, {2. / 3, 2. / 3 * a, 1. / 3}, {2. / 3, -2. / 3 * a, 1. / 3}, {1. / 12, b / 12, 1. / 6}, {1. / 12, -b / 12, 1. / 6}, {0, 0, 1}};
  double G[6][3];
  for (int i = 0; i < T; ++i)
    for (int k = 0; k < 3; ++k) G[i][k] = m == 4 ? G4[i][k] : G2[i][k];
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
        double t[6][3];
        for (int xi = 0; xi < T; ++xi)
          for (int kx = 0; kx < 3; ++kx) t[xi][kx] = G[xi][0] * g[kx] + G[xi][1] * g[3 + kx] + G[xi][2] * g[6 + kx];
        for (int xi = 0; xi < T; ++xi)
          for (int nu = 0; nu < T; ++nu) {
            const float v = (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
            const size_t idx = (size_t)(T * xi + nu) * batch_elems + (s * Npad + n) * 32 + j * 16 + kk;
            const uint16_t hi = host_bf16(v);
            packed[idx] = hi;
            packed[image_elems + idx] = host_bf16(v - host_bf16_f32(hi));
          }
      }
    }
  });
}


// The same on the device (round 4): the transformed weights of the 1500 -> 1500 stage are 36 x 3 x 1500 x 1536 (hi, lo) pairs,
// 1 GB that the host computed on 16 threads and then uploaded -- 0.8 s of `bs predict`'s start.  Here the stage's raw weights go
// up (243 MB) and a kernel writes the images: one thread per (unit, output channel, channel of the unit), G g G^T in double
// without contraction (the host's operation order: the images are the same bit for bit), rounded to f32, split into (hi, lo).
struct WinoPackDev {
  const float* w;        // OIDHW f32
  const int* cin_of_v;   // V channel -> input channel, -1: pad
  const int* unit_kz;    // per unit: z tap, or -1 for a dummy unit
  const int* unit_vc0;
  uint16_t* image;       // hi image, then lo image
  size_t image_elems, batch_elems;
  int cout, cin, Npad, nunits, T;
};
// T (4 or 6) is a compile-time constant: t[][] and the rows of G are indexed by unrolled loops only and stay in registers.
// (With T read from the arguments they lived in a 160-byte scratch segment; no kernel of the network engine has one now.)
template <int T>
__global__ __launch_bounds__(256) void wino_pack_kernel(const WinoPackDev a) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)(a.nunits / kUnitsPerStep) * a.cout * 32;
  if (i >= total) return;
  const int kk = (int)(i & 15), j = (int)((i >> 4) & 1);
  const size_t r = i >> 5;
  const int n = (int)(r % a.cout);
  const size_t s = r / a.cout;
  const int u = (int)(s * kUnitsPerStep + j);
  const int kz = a.unit_kz[u];
  if (kz < 0) return;
  const int c = a.cin_of_v[a.unit_vc0[u] + kk];
  if (c < 0) return;
  const float* g = a.w + (((size_t)n * a.cin + c) * 3 + kz) * 9;
  double gd[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) gd[k] = (double)g[k];
  constexpr double aa = 0.70710678118654752440, bb = 1.41421356237309504880;
  constexpr double G4[6][3] = {{1, 0, 0}, {2. / 3, 2. / 3 * aa, 1. / 3}, {2. / 3, -2. / 3 * aa, 1. / 3}, {1. / 12, bb / 12, 1. / 6}, {1. / 12, -bb / 12, 1. / 6}, {0, 0, 1}};
  constexpr double G2[6][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}, {0, 0, 0}, {0, 0, 0}};
  double t[T][3];
#pragma unroll
  for (int xi = 0; xi < T; ++xi)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const double g0 = T == 6 ? G4[xi][0] : G2[xi][0], g1 = T == 6 ? G4[xi][1] : G2[xi][1], g2 = T == 6 ? G4[xi][2] : G2[xi][2];
      t[xi][kx] = g0 * gd[kx] + g1 * gd[3 + kx] + g2 * gd[6 + kx];
    }
#pragma unroll
  for (int xi = 0; xi < T; ++xi)
#pragma unroll
    for (int nu = 0; nu < T; ++nu) {
      const double g0 = T == 6 ? G4[nu][0] : G2[nu][0], g1 = T == 6 ? G4[nu][1] : G2[nu][1], g2 = T == 6 ? G4[nu][2] : G2[nu][2];
      const float v = (float)(t[xi][0] * g0 + t[xi][1] * g1 + t[xi][2] * g2);
      const size_t idx = (size_t)(T * xi + nu) * a.batch_elems + (s * a.Npad + n) * 32 + j * 16 + kk;
      const uint16_t hi = to_bf16(v);
      a.image[idx] = hi;
      a.image[a.image_elems + idx] = to_bf16(v - __uint_as_float((uint32_t)hi << 16));
    }
}

int wino_pack_weights_dev(const float* w, int cout, int cin, const std::vector<int>& cin_of_v, int Npad, const std::vector<WinoUnit>& units, int m,
                          void** image_dev, size_t& image_elems, size_t& batch_elems) {
  const size_t nsteps = units.size() / kUnitsPerStep;
  const int T = m + 2, nbatch = T * T;
  batch_elems = nsteps * (size_t)Npad * 32;
  image_elems = nbatch * batch_elems + (size_t)kWeightRowSlack * 32;
  uint16_t* image = nullptr;
  BSMI_HIP(hipMalloc((void**)&image, 2 * image_elems * sizeof(uint16_t)));
  hipError_t err = hipMemsetAsync(image, 0, 2 * image_elems * sizeof(uint16_t), nullptr);
  float* wd = nullptr;
  int* meta = nullptr;
  std::vector<int> hm(cin_of_v);
  const size_t o_kz = hm.size();
  for (const WinoUnit& u : units) hm.push_back(u.dummy ? -1 : u.kz);
  const size_t o_vc = hm.size();
  for (const WinoUnit& u : units) hm.push_back(u.dummy ? 0 : u.vc0);
  const size_t wbytes = (size_t)cout * cin * 27 * sizeof(float);
  if (err == hipSuccess) err = hipMalloc((void**)&wd, wbytes);
  if (err == hipSuccess) err = hipMalloc((void**)&meta, hm.size() * sizeof(int));
  if (err == hipSuccess) err = hipMemcpy(wd, w, wbytes, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(meta, hm.data(), hm.size() * sizeof(int), hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    WinoPackDev a;
    a.w = wd; a.cin_of_v = meta; a.unit_kz = meta + o_kz; a.unit_vc0 = meta + o_vc; a.image = image;
    a.image_elems = image_elems; a.batch_elems = batch_elems; a.cout = cout; a.cin = cin; a.Npad = Npad; a.nunits = (int)units.size(); a.T = T;
    const size_t total = nsteps * (size_t)cout * 32;
    if (T == 6) hipLaunchKernelGGL(wino_pack_kernel<6>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, a);
    else hipLaunchKernelGGL(wino_pack_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, a);
    err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
  }
  if (wd) (void)hipFree(wd);
  if (meta) (void)hipFree(meta);
  if (err != hipSuccess) {
    (void)hipFree(image);
    BSMI_HIP(err);
  }
  *image_dev = image;
  return BSMI_OK;
}

}  // namespace bsmi
