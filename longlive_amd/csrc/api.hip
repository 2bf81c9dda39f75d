// Error plumbing + version for the C ABI (include/longlive_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "build/asm_knobs.h"
#include "common.h"

static thread_local char g_err[512] = "";

void ll_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ll_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ll_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return LL_ERR_LAUNCH;
  }
  return LL_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, DEVICE): the attribute belongs to the device's copy of the function,
// so a process that drives two GPUs must set it on each (a process-wide "done" flag made the second device's first launch fail).
int ll_lds_attr(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  int& have = done[std::make_pair(fn, dev)];
  if (have >= bytes) return LL_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    ll_set_error("hipFuncSetAttribute(%d bytes of LDS) failed on device %d: %s", bytes, dev, hipGetErrorString(e));
    return LL_ERR_LAUNCH;
  }
  have = bytes;
  return LL_OK;
}

// Generator knobs the library was built with (build/asm_knobs.h; empty for the default schedule): appended to the ll_*_plan strings
// so that a record made with a diagnostic or re-scheduled build says so.
const char* ll_asm_knobs(void) { return LL_ASM_KNOBS; }
void ll_plan_append_knobs(char* out, int cap) {
  if (LL_ASM_KNOBS[0] == 0 || out == nullptr) return;
  size_t n = strlen(out);
  if ((int)n + 12 < cap) snprintf(out + n, (size_t)cap - n, " [generator knobs: %s]", LL_ASM_KNOBS);
}

extern "C" int ll_version(void) { return LL_ABI_VERSION; }
extern "C" const char* ll_last_error(void) { return g_err; }
