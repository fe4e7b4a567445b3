// Dev probe: shader cycles per v_mfma_f32_16x16x32_bf16 in a few register arrangements (one wave per SIMD, or two).
// build: hipcc -O3 --offload-arch=gfx950 -o ab/mfma_probe tools/mfma_probe.hip ; run on the GPU box
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4_t* in, float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  u32x4_t a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa[2][4], fb[2][NACC / 4 > 0 ? NACC / 4 : 1];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[q][i] = in[(tid + 37 * (q * 4 + i)) & 1023];
#pragma unroll
    for (int i = 0; i < (NACC / 4 > 0 ? NACC / 4 : 1); ++i) fb[q][i] = in[(tid + 91 * (q * 8 + i) + 5) & 1023];
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const u32x4_t av = (r == 1) ? a1 : a0;
        u32x4_t bv = (r == 2) ? b1 : b0;
        if (MODE == 1 && (i % 3) == 1)  // operand through v_alignbyte, as the tap-1 operand
          bv = u32x4_t{__builtin_amdgcn_alignbyte(bv.y, bv.x, 2), __builtin_amdgcn_alignbyte(bv.z, bv.y, 2), __builtin_amdgcn_alignbyte(bv.w, bv.z, 2),
                       __builtin_amdgcn_alignbyte(bv.x, bv.w, 2)};
        if (MODE == 3) {  // as 2, B changes every instruction and A every NACC / 4
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[r & 1][i / (NACC / 4)]), __builtin_bit_cast(bf16x8_t, fb[r >> 1][i % (NACC / 4)]), acc[i], 0, 0, 0);
          continue;
        }
        if (MODE == 4) {  // as 2 with the accumulators pinned to AGPRs, in place
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(fa[r & 1][i & 3]), "v"(fb[r >> 1][i >> 2]));
          continue;
        }
        if (MODE == 5) {  // as 2 with the accumulators pinned to VGPRs, in place
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa[r & 1][i & 3]), "v"(fb[r >> 1][i >> 2]));
          continue;
        }
        if (MODE == 2) {  // distinct fragments: 4 A x NACC / 4 B, as a register tile
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[r & 1][i & 3]), __builtin_bit_cast(bf16x8_t, fb[r >> 1][i >> 2]), acc[i], 0, 0, 0);
          continue;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv), acc[i], 0, 0, 0);
      }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
static void run(const char* label, const u32x4_t* in, float* out, long long* cyc, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 3 * NACC;
  printf("%-52s blocks %4d: %7.2f shader cycles per MFMA (wave 0), wall %.3f ms = %.1f ns per MFMA per wave\n", label, blocks, c / n, ms, ms * 1e6 / n);
}

int main() {
  std::vector<uint32_t> h(1024 * 4);
  u32x4_t* in; float* out; long long* cyc;
  hipMalloc(&in, 1024 * 16); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t i = 0; i < h.size(); ++i) {
      // pass 0: zeros; pass 1: bf16 pairs of small magnitudes (1.0 .. 2.0 x 2^-20, as gradients are)
      const uint32_t lo = 0x3580 + (uint32_t)((i * 7919u) & 0x7f), hi = 0x3580 + (uint32_t)((i * 104729u) & 0x7f);
      h[i] = pass == 0 ? 0u : (lo | (hi << 16));
    }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("== operands: %s\n", pass == 0 ? "zeros" : "random small bf16");
    run<12, 0>("12 accumulators in place, 1 wave/SIMD", in, out, cyc, 256, 20000);
    run<24, 0>("24 accumulators in place, 1 wave/SIMD", in, out, cyc, 256, 10000);
    run<24, 1>("24 accumulators, every third B through alignbyte", in, out, cyc, 256, 10000);
    run<24, 0>("24 accumulators in place, 2 waves/SIMD", in, out, cyc, 512, 10000);
    run<4, 0>("4 accumulators in place (dependent after 4)", in, out, cyc, 256, 40000);
    run<24, 2>("24 accumulators, 4 x 6 distinct fragments, 1 wave", in, out, cyc, 256, 10000);
    run<24, 2>("24 accumulators, 4 x 6 distinct fragments, 2 waves", in, out, cyc, 512, 10000);
    run<24, 3>("4 x 6 distinct, B fastest, 1 wave", in, out, cyc, 256, 10000);
    run<24, 4>("4 x 6 distinct, accumulators in AGPRs, 1 wave", in, out, cyc, 256, 10000);
    run<24, 4>("4 x 6 distinct, accumulators in AGPRs, 2 waves", in, out, cyc, 512, 10000);
    run<24, 5>("4 x 6 distinct, accumulators in VGPRs, 1 wave", in, out, cyc, 256, 10000);
    run<24, 5>("4 x 6 distinct, accumulators in VGPRs, 2 waves", in, out, cyc, 512, 10000);
  }
  return 0;
}
