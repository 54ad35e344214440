// Error channel and library-level entry points of the C ABI (include/missm_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "missm_internal.h"

static thread_local char g_err[512] = "";

void missm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* missm_last_error(void) { return g_err; }

extern "C" int missm_abi_version(void) { return 11; }

extern "C" int missm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
