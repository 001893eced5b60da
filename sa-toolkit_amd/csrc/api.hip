// Library-level entry points: ABI version, error string, device probe.
#include <cstring>

#include "common.h"

namespace sat {
static thread_local char g_err[512] = "";

static thread_local char g_dispatch[384] = "";

void note_dispatch(const char* kernel, const char* launcher) {
  // "... launch_lean(...) [KS = 11, TG = 6]" -> "kernel<KS = 11, TG = 6>"
  const char* lb = launcher ? strrchr(launcher, '[') : nullptr;
  if (lb && strchr(lb, ']')) {
    const int n = (int)(strchr(lb, ']') - lb - 1);
    snprintf(g_dispatch, sizeof(g_dispatch), "%s<%.*s>", kernel, n, lb + 1);
  } else {
    snprintf(g_dispatch, sizeof(g_dispatch), "%s", kernel);
  }
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sat

extern "C" int sat_abi_version(void) { return SAT_ABI_VERSION; }

extern "C" const char* sat_last_error(void) { return sat::g_err; }

extern "C" const char* sat_last_dispatch_name(void) { return sat::g_dispatch; }

extern "C" int sat_device_info(char* name, int name_len, int* cu_count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    sat::set_error("no HIP device visible");
    return SAT_ERR_NO_DEVICE;
  }
  int dev = 0;
  SAT_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SAT_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  return SAT_OK;
}


// ---- diagnostic: the shader clock the chip holds while other work runs ------------------------------------------
// One wave samples (s_memtime, s_memrealtime) every `period_us` for `n` samples; clock between two samples =
// d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X guide, "DVFS give-back" item 6).  Launched on its own stream beside
// the path's kernels by tools/clock_probe.py; occupies one wave slot of one CU and issues almost nothing (s_sleep).
namespace sat {
__global__ void __launch_bounds__(64) clock_probe_kernel(long long* out, int n, int period_ticks) {
  if (threadIdx.x != 0) return;
  long long next = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    while ((long long)__builtin_amdgcn_s_memrealtime() < next) __builtin_amdgcn_s_sleep(32);
    out[2 * i] = (long long)__builtin_amdgcn_s_memtime();
    out[2 * i + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    next += period_ticks;
  }
}
}  // namespace sat

extern "C" int sat_clock_probe(int64_t* samples, int n, int period_us, void* stream) {
  SAT_REQUIRE(samples && n > 1 && period_us > 0, "clock_probe: bad arguments");
  hipLaunchKernelGGL(sat::clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)samples, n, period_us * 100);
  SAT_LAUNCH_CHECK("clock_probe_kernel");
  return SAT_OK;
}
