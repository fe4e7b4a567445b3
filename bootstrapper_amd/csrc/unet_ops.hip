// Memory-bound U-Net operators around the implicit-GEMM conv (gfx950):
// input normalisation, max-pool, trilinear upsample + crop, sigmoid heads, and the
// reflect-padded block extraction of the predict worker.  All activation tensors are
// channels-last [D][H][W][Cpad]; every thread moves one 16-byte channel vector.
#include "unet_ops.h"

namespace bsmi {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

template <typename T>
struct Vec;  // 16-byte vector of T, unpacked to floats
template <>
struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(u32x4_t v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t v = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    return v;
  }
};
template <>
struct Vec<uint16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(u32x4_t v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4_t pack(const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(f[2 * i]) | ((uint32_t)f32_to_bf16(f[2 * i + 1]) << 16);
    u32x4_t v = {w[0], w[1], w[2], w[3]};
    return v;
  }
};

// One 16-byte channel vector as floats, at element index `e` (a multiple of the vector length) of the plain
// [voxel][Cpad] order.  SP (BSMI_PREC_BF16X3): every 16-byte vector of hi values is followed by the 16 bytes of their
// lo values (conv_dev.h act_index); a value is hi + lo (exact in f32) and is stored as hi = bf16(v), lo = bf16(v - hi).
template <typename T, bool SP>
__device__ __forceinline__ void load_vec(const T* base, size_t e, float* f) {
  const T* p = base + (SP ? 2 * e : e);
  Vec<T>::unpack(*(const u32x4_t*)p, f);
  if constexpr (SP) {
    float g[Vec<T>::N];
    Vec<T>::unpack(*(const u32x4_t*)(p + Vec<T>::N), g);
#pragma unroll
    for (int k = 0; k < Vec<T>::N; ++k) f[k] += g[k];
  }
}
template <typename T, bool SP, bool NT = false>
__device__ __forceinline__ void store_vec(T* base, size_t e, const float* f) {
  T* p = base + (SP ? 2 * e : e);
  const u32x4_t hv = Vec<T>::pack(f);
  if constexpr (NT) __builtin_nontemporal_store(hv, (u32x4_t*)p);
  else *(u32x4_t*)p = hv;
  if constexpr (SP) {
    float h[Vec<T>::N], r[Vec<T>::N];
    Vec<T>::unpack(hv, h);
#pragma unroll
    for (int k = 0; k < Vec<T>::N; ++k) r[k] = f[k] - h[k];
    const u32x4_t lv = Vec<T>::pack(r);
    if constexpr (NT) __builtin_nontemporal_store(lv, (u32x4_t*)(p + Vec<T>::N));
    else *(u32x4_t*)(p + Vec<T>::N) = lv;
  }
}

// ---- input: raw [Cin][D][H][W] (u8 or f32) -> [D][H][W][Cpad] ---------------------
// u8 path restates gp.Normalize + IntensityScaleShift(raw, 2, -1)
// (reference models/3d_affs/predict.py:147-149): x = u8 * (1/255) * 2 - 1 in f32;
// `unit`: gp.Normalize only (x = u8 * (1/255)), the inputs of the second-stage nets.
template <typename T, typename RAW, bool SP = false>
__global__ void input_prep_kernel(const RAW* raw, T* out, int cin, int cpad, size_t nvox, int unit) {
  // one 16-byte channel vector per thread
  constexpr int N = Vec<T>::N;
  const int cv = cpad / N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox * cv) return;
  const int c0 = (int)(i % cv) * N;
  const size_t v = i / cv;
  float f[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float x = 0.f;
    if (c0 + k < cin) {
      if constexpr (sizeof(RAW) == 1) {
        x = (float)raw[(size_t)(c0 + k) * nvox + v] * (1.0f / 255.0f);
        if (!unit) x = x * 2.0f + -1.0f;
      } else {
        x = (float)raw[(size_t)(c0 + k) * nvox + v];
      }
    }
    f[k] = x;
  }
  store_vec<T, SP>(out, i * N, f);
}

int launch_input_prep(int precision, const void* raw, int raw_dtype, void* out, int cin, int cpad,
                      size_t nvox, hipStream_t s) {
  const int bs = 256;
  const int unit = raw_dtype == BSMI_RAW_U8_UNIT;
  if (precision == BSMI_PREC_F32) {
    const unsigned grid = (unsigned)ceil_div64((int64_t)(nvox * (cpad / 4)), bs);
    if (raw_dtype != BSMI_RAW_F32)
      hipLaunchKernelGGL((input_prep_kernel<float, uint8_t>), dim3(grid), dim3(bs), 0, s, (const uint8_t*)raw, (float*)out, cin, cpad, nvox, unit);
    else
      hipLaunchKernelGGL((input_prep_kernel<float, float>), dim3(grid), dim3(bs), 0, s, (const float*)raw, (float*)out, cin, cpad, nvox, 0);
  } else if (precision == BSMI_PREC_BF16X3) {
    const unsigned grid = (unsigned)ceil_div64((int64_t)(nvox * (cpad / 8)), bs);
    if (raw_dtype != BSMI_RAW_F32)
      hipLaunchKernelGGL((input_prep_kernel<uint16_t, uint8_t, true>), dim3(grid), dim3(bs), 0, s, (const uint8_t*)raw, (uint16_t*)out, cin, cpad, nvox, unit);
    else
      hipLaunchKernelGGL((input_prep_kernel<uint16_t, float, true>), dim3(grid), dim3(bs), 0, s, (const float*)raw, (uint16_t*)out, cin, cpad, nvox, 0);
  } else {
    const unsigned grid = (unsigned)ceil_div64((int64_t)(nvox * (cpad / 8)), bs);
    if (raw_dtype != BSMI_RAW_F32)
      hipLaunchKernelGGL((input_prep_kernel<uint16_t, uint8_t>), dim3(grid), dim3(bs), 0, s, (const uint8_t*)raw, (uint16_t*)out, cin, cpad, nvox, unit);
    else
      hipLaunchKernelGGL((input_prep_kernel<uint16_t, float>), dim3(grid), dim3(bs), 0, s, (const float*)raw, (uint16_t*)out, cin, cpad, nvox, 0);
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// ---- max-pool (reference unet.py:79-106 Downsample, MaxPool3d kernel=stride=factor) ---
template <typename T, bool SP = false>
__global__ void maxpool_kernel(const T* in, T* out, int H, int W, int C, int Do, int Ho, int Wo,
                               int fz, int fy, int fx) {
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Do * Ho * Wo * cv;
  if (i >= total) return;
  const int c = (int)(i % cv);
  size_t v = i / cv;
  const int x = (int)(v % Wo); v /= Wo;
  const int y = (int)(v % Ho);
  const int z = (int)(v / Ho);
  float m[N];
#pragma unroll
  for (int k = 0; k < N; ++k) m[k] = -INFINITY;
  for (int dz = 0; dz < fz; ++dz)
    for (int dy = 0; dy < fy; ++dy)
      for (int dx = 0; dx < fx; ++dx) {
        const size_t src = ((size_t)((z * fz + dz) * H + (y * fy + dy)) * W + (x * fx + dx)) * C + c * N;
        float f[N];
        load_vec<T, SP>(in, src, f);
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], f[k]);
      }
  store_vec<T, SP>(out, i * N, m);
}

int launch_maxpool(int precision, const void* in, void* out, int D, int H, int W, int C, int fz,
                   int fy, int fx, hipStream_t s) {
  const int Do = D / fz, Ho = H / fy, Wo = W / fx;
  const int bs = 256;
  if (precision == BSMI_PREC_F32) {
    const size_t total = (size_t)Do * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)ceil_div64(total, bs)), dim3(bs), 0, s,
                       (const float*)in, (float*)out, H, W, C, Do, Ho, Wo, fz, fy, fx);
  } else if (precision == BSMI_PREC_BF16X3) {
    const size_t total = (size_t)Do * Ho * Wo * (C / 8);
    hipLaunchKernelGGL((maxpool_kernel<uint16_t, true>), dim3((unsigned)ceil_div64(total, bs)), dim3(bs), 0, s,
                       (const uint16_t*)in, (uint16_t*)out, H, W, C, Do, Ho, Wo, fz, fy, fx);
  } else {
    const size_t total = (size_t)Do * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool_kernel<uint16_t>, dim3((unsigned)ceil_div64(total, bs)), dim3(bs), 0, s,
                       (const uint16_t*)in, (uint16_t*)out, H, W, C, Do, Ho, Wo, fz, fy, fx);
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// ---- trilinear upsample (align_corners=False) fused with the centre crop -------------
// reference unet.py:143 torch.nn.Upsample(scale_factor, mode="trilinear") followed by
// crop_to_factor (unet.py:147-201).  out voxel (z,y,x) = upsampled voxel (z+oz, y+oy, x+ox).
// Source index rule of torch (area_pixel_compute_source_index, align_corners=False):
//   src = (dst + 0.5) / f - 0.5, clamped at 0; i0 = floor(src), i1 = min(i0+1, n-1),
//   w1 = src - i0, w0 = 1 - w1.
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

// One output voxel (z, y, x), one channel vector, the general way.
template <typename T, bool SP>
__device__ __forceinline__ void upsample_one(const T* in, T* out, int D, int H, int W, int C, int Ho, int Wo,
                                             int fz, int fy, int fx, int oz, int oy, int ox, int z, int y, int x, int c) {
  constexpr int N = Vec<T>::N;
  int z0, z1, y0, y1, x0, x1;
  float wz0, wz1, wy0, wy1, wx0, wx1;
  lin_src(z + oz, fz, D, z0, z1, wz0, wz1);
  lin_src(y + oy, fy, H, y0, y1, wy0, wy1);
  lin_src(x + ox, fx, W, x0, x1, wx0, wx1);
  auto plane = [&](int zz, float* o) {
    float a[N], b[N], p[N], q[N];
    load_vec<T, SP>(in, (((size_t)zz * H + y0) * W + x0) * C + c, a);
    load_vec<T, SP>(in, (((size_t)zz * H + y0) * W + x1) * C + c, b);
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = wx0 * a[k] + wx1 * b[k];
    load_vec<T, SP>(in, (((size_t)zz * H + y1) * W + x0) * C + c, a);
    load_vec<T, SP>(in, (((size_t)zz * H + y1) * W + x1) * C + c, b);
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = wx0 * a[k] + wx1 * b[k];
#pragma unroll
    for (int k = 0; k < N; ++k) o[k] = wy0 * p[k] + wy1 * q[k];
  };
  float r[N], acc[N];
  plane(z0, r);
  if (z1 != z0) {
    float r1[N];
    plane(z1, r1);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = wz0 * r[k] + wz1 * r1[k];
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = wz0 * r[k] + wz1 * r[k];
  }
  store_vec<T, SP, true>(out, (((size_t)z * Ho + y) * Wo + x) * C + c, acc);
}

// General factors: one workgroup per output row (z, y), a thread walks the row's (x, channel vector) pairs.
template <typename T, bool SP = false>
__global__ void upsample_crop_kernel(const T* in, T* out, int D, int H, int W, int C, int Do, int Ho,
                                     int Wo, int fz, int fy, int fx, int oz, int oy, int ox) {
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const int y = blockIdx.x % Ho, z = blockIdx.x / Ho;
  for (int i = threadIdx.x; i < Wo * cv; i += blockDim.x) {
    const int x = i / cv, c = (i - x * cv) * N;
    upsample_one<T, SP>(in, out, D, H, W, C, Ho, Wo, fz, fy, fx, oz, oy, ox, z, y, x, c);
  }
}

// Factor (1, 2, 2), the factor of every shipped setup: the outputs at global (pre-crop) coordinates (2p - 1, 2p)
// along y and along x all interpolate the same two source lines / columns (p - 1, p) with weights (3/4, 1/4) and
// (1/4, 3/4), so a thread that owns such a 2 x 2 patch loads 4 source vectors for 4 outputs instead of 16.  The
// arithmetic is that of upsample_one, operation for operation; patches that touch the border, where the source
// index is clamped, take the general path.
template <typename T, bool SP = false>
__global__ void upsample2x_kernel(const T* in, T* out, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                  int oz, int oy, int ox, int py0, int npy, int px0, int npx) {
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const int py = py0 + blockIdx.x % npy, z = blockIdx.x / npy;
  const int ya = 2 * py - 1 - oy;  // output rows ya, ya + 1
  const bool row_in = ya >= 0 && ya + 1 < Ho && py - 1 >= 0 && py <= H - 1;
  const int n = npx * cv;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int pxi = i / cv, c = (i - pxi * cv) * N;
    const int px = px0 + pxi;
    const int xa = 2 * px - 1 - ox;
    if (row_in && xa >= 0 && xa + 1 < Wo && px - 1 >= 0 && px <= W - 1) {
      float s00[N], s01[N], s10[N], s11[N];
      const size_t base = (((size_t)(z + oz) * H + (py - 1)) * W + (px - 1)) * C + c;
      load_vec<T, SP>(in, base, s00);
      load_vec<T, SP>(in, base + C, s01);
      load_vec<T, SP>(in, base + (size_t)W * C, s10);
      load_vec<T, SP>(in, base + (size_t)W * C + C, s11);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float wy1 = r ? 0.75f : 0.25f, wy0 = 1.f - wy1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float wx1 = q ? 0.75f : 0.25f, wx0 = 1.f - wx1;
          float acc[N];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const float p = wx0 * s00[k] + wx1 * s01[k];
            const float qq = wx0 * s10[k] + wx1 * s11[k];
            const float v = wy0 * p + wy1 * qq;
            acc[k] = 1.f * v + 0.f * v;  // the z weights (1, 0) of upsample_one
          }
          store_vec<T, SP, true>(out, (((size_t)z * Ho + ya + r) * Wo + xa + q) * C + c, acc);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int y = ya + r, x = xa + q;
          if (y >= 0 && y < Ho && x >= 0 && x < Wo)
            upsample_one<T, SP>(in, out, D, H, W, C, Ho, Wo, 1, 2, 2, oz, oy, ox, z, y, x, c);
        }
    }
  }
}

int launch_upsample_crop(int precision, const void* in, void* out, int D, int H, int W, int C, int Do,
                         int Ho, int Wo, int fz, int fy, int fx, int oz, int oy, int ox, hipStream_t s) {
  const int bs = 256;
  const bool sp = precision == BSMI_PREC_BF16X3;
  if (fz == 1 && fy == 2 && fx == 2) {
    const int py0 = (oy + 1) / 2, npy = (oy + Ho) / 2 - py0 + 1;
    const int px0 = (ox + 1) / 2, npx = (ox + Wo) / 2 - px0 + 1;
    const unsigned grid = (unsigned)(Do * npy);
    if (precision == BSMI_PREC_F32)
      hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(grid), dim3(bs), 0, s, (const float*)in, (float*)out, D, H, W, C,
                         Do, Ho, Wo, oz, oy, ox, py0, npy, px0, npx);
    else if (sp)
      hipLaunchKernelGGL((upsample2x_kernel<uint16_t, true>), dim3(grid), dim3(bs), 0, s, (const uint16_t*)in, (uint16_t*)out,
                         D, H, W, C, Do, Ho, Wo, oz, oy, ox, py0, npy, px0, npx);
    else
      hipLaunchKernelGGL(upsample2x_kernel<uint16_t>, dim3(grid), dim3(bs), 0, s, (const uint16_t*)in, (uint16_t*)out, D,
                         H, W, C, Do, Ho, Wo, oz, oy, ox, py0, npy, px0, npx);
    BSMI_HIP(hipGetLastError());
    return BSMI_OK;
  }
  const unsigned grid = (unsigned)(Do * Ho);
  if (precision == BSMI_PREC_F32)
    hipLaunchKernelGGL(upsample_crop_kernel<float>, dim3(grid), dim3(bs), 0, s, (const float*)in, (float*)out, D, H, W, C,
                       Do, Ho, Wo, fz, fy, fx, oz, oy, ox);
  else if (sp)
    hipLaunchKernelGGL((upsample_crop_kernel<uint16_t, true>), dim3(grid), dim3(bs), 0, s, (const uint16_t*)in, (uint16_t*)out,
                       D, H, W, C, Do, Ho, Wo, fz, fy, fx, oz, oy, ox);
  else
    hipLaunchKernelGGL(upsample_crop_kernel<uint16_t>, dim3(grid), dim3(bs), 0, s, (const uint16_t*)in, (uint16_t*)out, D,
                       H, W, C, Do, Ho, Wo, fz, fy, fx, oz, oy, ox);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

// ---- sigmoid head: ConvPass(C, dims, [[1,1,1]], "Sigmoid") ----------------------------
// reference model.py:54-56 + unet.py:63-76: sigmoid((W1 z + b1) + (W2 z + b2)); f32 math.
// hw: [cout][2][cin] (conv_pass.0 then residual.0), hb: [cout][2].
// No private array with a run-time index: the thread's CPAD input channels stay in registers (the loops over channels are
// unrolled at compile time, the head's real channel count only masks them).  The earlier form kept them in scratch memory,
// and a kernel with a scratch segment is what two forward passes side by side on one card corrupted.
template <typename T, bool SP, int CPAD>
__global__ void __launch_bounds__(256) head_kernel(const T* z, int cin, int cout, const float* hw, const float* hb,
                                                   float* out_f32, uint8_t* out_u8, size_t nvox) {
  extern __shared__ float sw[];  // [cout][2][CPAD], zero beyond cin; then [cout][2] biases
  float* sb = sw + cout * 2 * CPAD;
  for (int i = threadIdx.x; i < cout * 2 * CPAD; i += blockDim.x) {
    const int c = i % CPAD;
    sw[i] = c < cin ? hw[(i / CPAD) * cin + c] : 0.f;
  }
  for (int i = threadIdx.x; i < cout * 2; i += blockDim.x) sb[i] = hb[i];
  __syncthreads();
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  constexpr int N = Vec<T>::N;
  static_assert(CPAD % N == 0, "channel padding is a multiple of the vector width");
  float f[CPAD];
#pragma unroll
  for (int c = 0; c < CPAD; c += N) load_vec<T, SP>(z, v * CPAD + c, f + c);
  for (int o = 0; o < cout; ++o) {
    const float* w1 = sw + (o * 2 + 0) * CPAD;
    const float* w2 = sw + (o * 2 + 1) * CPAD;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPAD; ++c)
      if (c < cin) {  // uniform; keeps the sums those of the cin real channels, in order
        s1 = fmaf(w1[c], f[c], s1);
        s2 = fmaf(w2[c], f[c], s2);
      }
    const float y = (s1 + sb[o * 2]) + (s2 + sb[o * 2 + 1]);
    const float sg = 1.0f / (1.0f + expf(-y));
    if (out_f32) out_f32[(size_t)o * nvox + v] = sg;
    // IntensityScaleShift(pred, 255, 0) then the uint8 dataset cast (truncation)
    if (out_u8) out_u8[(size_t)o * nvox + v] = (uint8_t)(sg * 255.0f);
  }
}

#ifdef BSMI_HEAD_SCRATCH
// Dev build only (make CXXFLAGS_EXTRA=-DBSMI_HEAD_SCRATCH OUT=../libbsmi_headscratch.so BUILD=build_hs): the head kernel as it was
// until round 4, its channels in a private array with a run-time index = a 272-byte scratch segment per lane.  This is the
// reproducer of the concurrent-forward defect (tools/debug_two_streams.py with BSMI_LIB pointing at that build).
template <typename T, bool SP>
__global__ void head_scratch_kernel(const T* z, int cpad, int cin, int cout, const float* hw, const float* hb, float* out_f32, uint8_t* out_u8,
                                    size_t nvox) {
  extern __shared__ float sw[];
  float* sb = sw + cout * 2 * cin;
  for (int i = threadIdx.x; i < cout * 2 * cin; i += blockDim.x) sw[i] = hw[i];
  for (int i = threadIdx.x; i < cout * 2; i += blockDim.x) sb[i] = hb[i];
  __syncthreads();
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
#ifndef BSMI_HEAD_SCRATCH_N
#define BSMI_HEAD_SCRATCH_N 64  // (16: a small scratch segment, for the question whether its size matters; cpad must be 16 then)
#endif
  float f[BSMI_HEAD_SCRATCH_N];
  constexpr int N = Vec<T>::N;
  for (int c = 0; c < cpad && c < BSMI_HEAD_SCRATCH_N; c += N) load_vec<T, SP>(z, v * cpad + c, f + c);
  for (int o = 0; o < cout; ++o) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < cin; ++c) {
      s1 = fmaf(sw[(o * 2 + 0) * cin + c], f[c], s1);
      s2 = fmaf(sw[(o * 2 + 1) * cin + c], f[c], s2);
    }
    const float y = (s1 + sb[o * 2]) + (s2 + sb[o * 2 + 1]);
    const float sg = 1.0f / (1.0f + expf(-y));
    if (out_f32) out_f32[(size_t)o * nvox + v] = sg;
    if (out_u8) out_u8[(size_t)o * nvox + v] = (uint8_t)(sg * 255.0f);
  }
}
#endif

template <typename T, bool SP>
static int launch_head_t(const T* z, int cpad, int cin, int cout, const float* hw, const float* hb, float* out_f32, uint8_t* out_u8,
                         size_t nvox, hipStream_t s) {
  const int bs = 256;
  const size_t smem = (size_t)(cout * 2 * cpad + cout * 2) * sizeof(float);
  const unsigned grid = (unsigned)ceil_div64((int64_t)nvox, bs);
#ifdef BSMI_HEAD_SCRATCH
  hipLaunchKernelGGL((head_scratch_kernel<T, SP>), dim3(grid), dim3(bs), (size_t)(cout * 2 * cin + cout * 2) * sizeof(float), s, z, cpad, cin, cout,
                     hw, hb, out_f32, out_u8, nvox);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
#endif
  switch (cpad) {
#define BSMI_HEAD_CASE(CP)                                                                                                       \
  case CP:                                                                                                                       \
    hipLaunchKernelGGL((head_kernel<T, SP, CP>), dim3(grid), dim3(bs), smem, s, z, cin, cout, hw, hb, out_f32, out_u8, nvox);     \
    break;
    BSMI_HEAD_CASE(16) BSMI_HEAD_CASE(32) BSMI_HEAD_CASE(48) BSMI_HEAD_CASE(64)  // multiples of kChanPad
#undef BSMI_HEAD_CASE
    default: BSMI_FAIL(BSMI_ERR_INVALID, "head kernel: %d padded input channels (a multiple of 16 up to 64 expected)", cpad);
  }
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

int launch_head(int precision, const void* z, int cpad, int cin, int cout, const float* hw,
                const float* hb, float* out_f32, uint8_t* out_u8, size_t nvox, hipStream_t s) {
  if (cpad > 64) BSMI_FAIL(BSMI_ERR_INVALID, "head kernel supports at most 64 input channels (got %d)", cpad);
  if (precision == BSMI_PREC_F32) return launch_head_t<float, false>((const float*)z, cpad, cin, cout, hw, hb, out_f32, out_u8, nvox, s);
  if (precision == BSMI_PREC_BF16X3) return launch_head_t<uint16_t, true>((const uint16_t*)z, cpad, cin, cout, hw, hb, out_f32, out_u8, nvox, s);
  return launch_head_t<uint16_t, false>((const uint16_t*)z, cpad, cin, cout, hw, hb, out_f32, out_u8, nvox, s);
}

// ---- reflect-padded block read ---------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {
  // numpy.pad mode='reflect' (edge not repeated), iterated for offsets larger than n
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i < n ? i : p - i;
}

__global__ void extract_block_reflect_kernel(const uint8_t* vol, int VD, int VH, int VW, int oz, int oy,
                                             int ox, int BD, int BH, int BW, uint8_t* block) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)BD * BH * BW;
  if (i >= total) return;
  const int x = (int)(i % BW);
  const size_t v = i / BW;
  const int y = (int)(v % BH);
  const int z = (int)(v / BH);
  const int sz = reflect_idx(z + oz, VD), sy = reflect_idx(y + oy, VH), sx = reflect_idx(x + ox, VW);
  block[i] = vol[((size_t)sz * VH + sy) * VW + sx];
}

int launch_extract_block_reflect(const uint8_t* vol, const int64_t vs[3], const int64_t off[3],
                                 const int64_t bs3[3], uint8_t* block, hipStream_t s) {
  const size_t total = (size_t)bs3[0] * bs3[1] * bs3[2];
  const int bs = 256;
  hipLaunchKernelGGL(extract_block_reflect_kernel, dim3((unsigned)ceil_div64((int64_t)total, bs)), dim3(bs), 0, s,
                     vol, (int)vs[0], (int)vs[1], (int)vs[2], (int)off[0], (int)off[1], (int)off[2],
                     (int)bs3[0], (int)bs3[1], (int)bs3[2], block);
  BSMI_HIP(hipGetLastError());
  return BSMI_OK;
}

}  // namespace bsmi

// ---- development aid: an LDS canary ---------------------------------------------------------------------------------------
// Workgroups that fill their LDS with a pattern, idle, and check it: run beside an engine on another stream, they tell whether
// any co-resident kernel writes outside its own LDS allocation (the concurrent-forward defect).
namespace bsmi {
__global__ void lds_canary_kernel(int words, int spins, unsigned long long* mismatches) {
  extern __shared__ uint32_t canary[];
  const uint32_t tag = 0xC0DE0000u ^ (blockIdx.x * 2654435761u);
  for (int i = threadIdx.x; i < words; i += blockDim.x) canary[i] = tag + (uint32_t)i;
  __syncthreads();
  for (int s = 0; s < spins; ++s) __builtin_amdgcn_s_sleep(127);
  __syncthreads();
  unsigned long long bad = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) bad += canary[i] != tag + (uint32_t)i;
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace bsmi

extern "C" int bsmi_debug_lds_canary(int lds_bytes, int blocks, int spins, unsigned long long* mismatches_dev, void* stream) {
  if (lds_bytes < 4 || lds_bytes > 64 * 1024 || blocks < 1 || !mismatches_dev) return BSMI_ERR_INVALID;
  hipLaunchKernelGGL(bsmi::lds_canary_kernel, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, lds_bytes / 4, spins, mismatches_dev);
  return hipGetLastError() == hipSuccess ? BSMI_OK : BSMI_ERR_HIP;
}
