// Shared helpers for libbsmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsmi.h"

namespace bsmi {

void set_error(const char* fmt, ...);

#define BSMI_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      bsmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                      __LINE__);                                                         \
      return BSMI_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

#define BSMI_FAIL(code, ...)      \
  do {                            \
    bsmi::set_error(__VA_ARGS__); \
    return (code);                \
  } while (0)

// "Once per kernel" set-up such as hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function object of the
// CURRENT device, so the done-flag is kept per device (a handle created on a second GPU of the same process would otherwise
// launch without the attribute), under a lock (two host threads may reach a kernel's first launch together).
//   static DeviceOnce once;  int rc = once.run([&]() -> int { BSMI_HIP(hipFuncSetAttribute(...)); return BSMI_OK; });
struct DeviceOnce {
  std::mutex m;
  uint64_t done[4] = {0, 0, 0, 0};  // up to 256 devices
  template <class F>
  int run(F&& f) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return BSMI_ERR_HIP;
    d &= 255;
    std::lock_guard<std::mutex> g(m);
    if (done[d >> 6] >> (d & 63) & 1) return BSMI_OK;
    const int rc = f();
    if (rc == BSMI_OK) done[d >> 6] |= 1ull << (d & 63);
    return rc;
  }
};

// f(i) for i in [0, n) on up to 16 host threads (weight packing at finalize: independent rows of a packed image)
template <class F>
static inline void host_parallel_for(size_t n, F&& f) {
  const size_t nt = std::min<size_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), n);
  if (nt <= 1) {
    for (size_t i = 0; i < n; ++i) f(i);
    return;
  }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (size_t i = t; i < n; i += nt) f(i);
    });
  for (auto& x : th) x.join();
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// channel padding unit of every channels-last activation tensor (elements)
constexpr int kChanPad = 16;

}  // namespace bsmi
