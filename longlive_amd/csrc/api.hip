// Error plumbing + version for the C ABI (include/longlive_hip.h).
#include <stdarg.h>
#include <stdio.h>

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

extern "C" int ll_version(void) { return 100; }
extern "C" const char* ll_last_error(void) { return g_err; }
