// Library-level entry points: ABI version, error string, device probe.
#include <cstring>

#include "common.h"

namespace sat {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sat

extern "C" int sat_abi_version(void) { return SAT_ABI_VERSION; }

extern "C" const char* sat_last_error(void) { return sat::g_err; }

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
