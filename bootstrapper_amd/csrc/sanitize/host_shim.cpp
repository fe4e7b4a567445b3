// Sanitizer build of libbsmi's HOST code only (`make -C bootstrapper_amd/csrc asan` -> ../libbsmi_host_asan.so).
// GPU AddressSanitizer is not available on the MI355X pool, and the host files -- the hand-written LZ4 / blosclz decoders of
// chunk_codec.cpp, the merge loops of agglo_host.cpp, the heap of flood_host.cpp -- are the pointer-heavy C++ that wants one:
// they are compiled here with g++ -fsanitize=address,undefined, without the device half of the library.  This file supplies
// the two things those sources take from the device half (the per-thread error message) and C entry points for the functions
// that the product only reaches through seg.hip.  Test infrastructure: nothing here is part of libbsmi.so.
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../../include/bsmi.h"

namespace bsmi {
static thread_local char g_err[1024];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
void host_flood3(int D, int H, int W, const uint8_t* mask, const int32_t* d2, int32_t* lab);
}  // namespace bsmi

extern "C" const char* bsmi_last_error(void) { return bsmi::g_err; }

extern "C" int bsmi_san_host_flood3(int D, int H, int W, const uint8_t* mask, const int32_t* d2, int32_t* lab) {
  bsmi::host_flood3(D, H, W, mask, d2, lab);
  return 0;
}
