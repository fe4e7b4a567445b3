// Dev probe: one lane chases `n` dependent loads through `buf` (a random cycle), another kernel does `n` dependent ALU ops:
// how long a latency-bound kernel takes in different power states of the chip.
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void chase_kernel(const uint32_t* buf, int n, uint32_t* out) {
  uint32_t p = 0;
  for (int i = 0; i < n; ++i) p = buf[p];
  out[0] = p;
}
extern "C" __global__ void alu_kernel(int n, uint32_t* out) {
  uint32_t p = 1;
  for (int i = 0; i < n; ++i) p = p * 1664525u + 1013904223u;
  out[1] = p;
}
extern "C" int chase_launch(const void* buf, int n, void* out, void* stream) {
  hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const uint32_t*)buf, n, (uint32_t*)out);
  return (int)hipGetLastError();
}
extern "C" int alu_launch(int n, void* out, void* stream) {
  hipLaunchKernelGGL(alu_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, n, (uint32_t*)out);
  return (int)hipGetLastError();
}
