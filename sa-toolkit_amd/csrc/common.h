// Shared host-side helpers for libsatools_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/satools_hip.h"

namespace sat {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SAT_ERR_HIP;
  }
  return SAT_OK;
}

#define SAT_HIP(call)                                   \
  do {                                                  \
    int _s = ::sat::check_hip((call), #call);           \
    if (_s != SAT_OK) return _s;                        \
  } while (0)

#define SAT_LAUNCH_CHECK(name) SAT_HIP(hipGetLastError())

#define SAT_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::sat::set_error(__VA_ARGS__);    \
      return SAT_ERR_INVALID;           \
    }                                   \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a launcher that raises it remembers which
// devices it has done so on.  `done` = one bit per device ordinal; the bit is published only after the attribute
// calls have returned (they are idempotent, so two host threads racing here both make them).
inline bool attr_needed_on_current_device(const std::atomic<uint64_t>& done, int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  *dev_out = dev;
  return dev >= 64 || !(done.load(std::memory_order_acquire) & (1ull << dev));
}
inline void attr_done_on_device(std::atomic<uint64_t>& done, int dev) {
  if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace sat
