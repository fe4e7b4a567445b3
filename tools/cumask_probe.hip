// Dev probe: which (XCC, SE, CU) do the workgroups of a CU-masked stream land on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
  // stay resident a little so that the grid spreads
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main(int argc, char** argv) {
  int nbits = argc > 1 ? atoi(argv[1]) : 32;
  int pattern = argc > 2 ? atoi(argv[2]) : 0;  // 0: first nbits, 1: every 8th bit
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  std::vector<uint32_t> mask(8, 0);
  for (int i = 0; i < 256; ++i) {
    bool on = pattern == 0 ? i < nbits : (i % (256 / nbits) == 0);
    if (on) mask[i / 32] |= 1u << (i % 32);
  }
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
  printf("create: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  unsigned* d; hipMalloc(&d, 2 * 1024 * 4);
  where<<<1024, 64, 0, s>>>(d);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2048);
  hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::map<unsigned, int>> cnt;
  for (int b = 0; b < 1024; ++b) {
    unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cnt[xcc][(se << 8) | (sh << 4) | cu]++;
  }
  int total = 0;
  for (auto& x : cnt) { printf("xcc %u: %zu CUs:", x.first, x.second.size()); total += x.second.size(); for (auto& c : x.second) printf(" %x", c.first); printf("\n"); }
  printf("distinct CUs used: %d\n", total);
  return 0;
}
