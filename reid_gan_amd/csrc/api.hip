// Error state and library identification for the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ried_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ried_last_error(void) { return g_err; }
extern "C" int ried_abi_version(void) { return 1; }

// Number of compute units of the current device (used by hosts to size split-K); 0 on failure.
extern "C" int ried_device_cus(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}
