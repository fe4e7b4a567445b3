// Shared helpers for libbsmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// channel padding unit of every channels-last activation tensor (elements)
constexpr int kChanPad = 16;

}  // namespace bsmi
