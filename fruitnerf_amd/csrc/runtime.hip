// runtime.hip — error plumbing and device queries for the C ABI.
#include <stdarg.h>
#include <string.h>

#include <vector>

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

// ---- event timing ---------------------------------------------------------------------------------
// Events are pooled: hipEventCreate costs ~10 us of host time, a profiled step records a dozen scopes.
struct ProfRec {
  hipEvent_t a, b;
  int op;
  long long units;
};
static bool g_prof_on = false;
static bool g_prof_paused = false;
static unsigned long long g_prof_mask = ~0ull;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;

static bool prof_event(hipEvent_t* e) {
  if (!g_prof_pool.empty()) {
    *e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return true;
  }
  return hipEventCreate(e) == hipSuccess;
}

ProfScope::ProfScope(int op, long long units, void* stream) : slot(-1), st(as_stream(stream)) {
  if (op < 0 || !g_prof_on || g_prof_paused || !((g_prof_mask >> op) & 1ull) || g_prof.size() >= (1u << 20)) return;   // op < 0: a scope that is part of its caller's
  ProfRec r;
  r.op = op;
  r.units = units;
  if (!prof_event(&r.a)) return;
  if (!prof_event(&r.b)) {
    g_prof_pool.push_back(r.a);
    return;
  }
  (void)hipEventRecord(r.a, st);
  slot = (int)g_prof.size();
  g_prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st);
}

}  // namespace fnr

extern "C" int fnr_profile_enable(int on, uint64_t op_mask) {
  for (auto& r : fnr::g_prof) {  // back to the pool (a pending event is re-recorded by its next user)
    fnr::g_prof_pool.push_back(r.a);
    fnr::g_prof_pool.push_back(r.b);
  }
  fnr::g_prof.clear();
  fnr::g_prof_on = on != 0;
  fnr::g_prof_paused = false;
  fnr::g_prof_mask = op_mask;
  return FNR_OK;
}

extern "C" int fnr_profile_pause(int paused) {
  fnr::g_prof_paused = paused != 0;
  return FNR_OK;
}

extern "C" int64_t fnr_profile_collect(int32_t* ops, int64_t* units, float* ms, int64_t capacity) {
  int64_t n = 0;
  for (auto& r : fnr::g_prof) {
    if (n >= capacity) break;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ops[n] = r.op;
    units[n] = r.units;
    ms[n] = t;
    ++n;
  }
  return n;
}

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
