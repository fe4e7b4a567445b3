// Dev aid: guarded device allocations for the forward engine.  With BSMI_GUARD_MB=<n> in the environment every
// allocation of the files that include this header sits between two n-MiB zones filled with 0xFF (a NaN in every
// float format in use): a read past a buffer that reaches a result poisons it, a write past a buffer is found by
// bsmi_debug_check_guards().  Unset: plain hipMalloc / hipFree.  Include it AFTER every other header.
#pragma once
#include <hip/hip_runtime.h>

namespace bsmi {
hipError_t guarded_malloc(void** p, size_t bytes, const char* file, int line);
hipError_t guarded_free(void* p);
}  // namespace bsmi

#define hipMalloc(p, n) ::bsmi::guarded_malloc((void**)(p), (n), __FILE__, __LINE__)
#define hipFree(p) ::bsmi::guarded_free((void*)(p))
