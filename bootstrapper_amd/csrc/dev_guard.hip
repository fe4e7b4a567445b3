// Guarded device allocations (dev_guard.h): the registry and the checker.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/bsmi.h"

namespace bsmi {

namespace {
struct GuardRec {
  char* base;
  size_t bytes;
  const char* file;
  int line;
};
std::mutex g_guard_mu;
std::map<void*, GuardRec> g_guards;

size_t guard_bytes() {
  static const size_t g = [] {
    const char* e = getenv("BSMI_GUARD_MB");
    return e ? (size_t)atol(e) << 20 : (size_t)0;
  }();
  return g;
}
}  // namespace

hipError_t guarded_malloc(void** p, size_t bytes, const char* file, int line) {
  const size_t g = guard_bytes();
  if (!g) return ::hipMalloc(p, bytes);
  char* base = nullptr;
  hipError_t e = ::hipMalloc((void**)&base, bytes + 2 * g);
  if (e != hipSuccess) return e;
  if ((e = hipMemset(base, 0xFF, g)) != hipSuccess) return e;
  if ((e = hipMemset(base + g + bytes, 0xFF, g)) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guards[base + g] = GuardRec{base, bytes, file, line};
  *p = base + g;
  return hipSuccess;
}

hipError_t guarded_free(void* p) {
  if (!guard_bytes() || !p) return ::hipFree(p);
  std::lock_guard<std::mutex> lk(g_guard_mu);
  auto it = g_guards.find(p);
  if (it == g_guards.end()) return ::hipFree(p);
  void* base = it->second.base;
  g_guards.erase(it);
  return ::hipFree(base);
}

}  // namespace bsmi

// Number of guard zones found written to (0 = all intact; -1 = a HIP error); each is reported on stderr with the
// allocation site, the buffer's size and the first / last touched offset relative to the buffer.
extern "C" int bsmi_debug_check_guards(void) {
  using namespace bsmi;
  const size_t g = guard_bytes();
  if (!g) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  std::lock_guard<std::mutex> lk(g_guard_mu);
  std::vector<unsigned char> host(g);
  int bad = 0;
  for (const auto& kv : g_guards) {
    const GuardRec& r = kv.second;
    for (int side = 0; side < 2; ++side) {
      const char* zone = side ? r.base + g + r.bytes : r.base;
      if (hipMemcpy(host.data(), zone, g, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      size_t first = g, last = 0, n = 0;
      for (size_t i = 0; i < g; ++i)
        if (host[i] != 0xFF) {
          if (first == g) first = i;
          last = i;
          ++n;
        }
      if (!n) continue;
      ++bad;
      if (side)
        fprintf(stderr, "guard: %s:%d buffer of %zu bytes: %zu bytes written PAST its end, at +%zu .. +%zu beyond the end\n", r.file,
                r.line, r.bytes, n, first, last);
      else
        fprintf(stderr, "guard: %s:%d buffer of %zu bytes: %zu bytes written BEFORE its start, at -%zu .. -%zu\n", r.file, r.line,
                r.bytes, n, g - first, g - last);
    }
  }
  return bad;
}
