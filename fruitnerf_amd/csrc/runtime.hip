// runtime.hip — error plumbing and device queries for the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace fnr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cu_count() {
  static thread_local int cached_dev = -1;
  static thread_local int cached_cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return cached_cus;
  if (dev != cached_dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
      cached_cus = cus;
      cached_dev = dev;
    }
  }
  return cached_cus;
}

}  // namespace fnr

extern "C" int fnr_abi_version(void) { return FNR_ABI_VERSION; }

extern "C" const char* fnr_last_error(void) { return fnr::g_err; }

extern "C" int fnr_device_check(int* cu_count_out, char* name_out, int name_len) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    fnr::set_error("no HIP device visible (%s); this library has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return FNR_ERR_HIP;
  }
  int dev = 0;
  FNR_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  FNR_HIP(hipGetDeviceProperties(&prop, dev));
  if (name_out && name_len > 0) {
    strncpy(name_out, prop.gcnArchName, (size_t)name_len - 1);
    name_out[name_len - 1] = 0;
  }
  if (cu_count_out) *cu_count_out = prop.multiProcessorCount;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fnr::set_error("device arch %s is not gfx950; kernels are built for MI355X only", prop.gcnArchName);
    return FNR_ERR_UNSUPPORTED;
  }
  return FNR_OK;
}
