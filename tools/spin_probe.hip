// Dev probe: occupy `blocks` CUs with one workgroup each for `ms` milliseconds.
// mode 0: pure ALU/sleep loop; mode 1: dependent global loads (latency-bound pointer chase); `lds` bytes of LDS held.
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void spin_kernel(unsigned long long ticks, int mode, uint32_t* buf, uint32_t mask) {
  extern __shared__ char sm[];
  const unsigned long long t0 = wall_clock64();
  uint32_t p = threadIdx.x + blockIdx.x * 977;
  while (wall_clock64() - t0 < ticks) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(8);
    } else {
      for (int i = 0; i < 16; ++i) p = buf[p & mask] + i;
    }
  }
  if (p == 0xdeadbeef) sm[0] = 1, buf[0] = p;
}
extern "C" int spin_launch(int blocks, int threads, int lds, double ms, int mode, void* buf, unsigned mask, void* stream) {
  static bool set = false;
  if (!set) { hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); set = true; }
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (unsigned long long)(ms * 1e5), mode, (uint32_t*)buf, mask);
  return (int)hipGetLastError();
}
