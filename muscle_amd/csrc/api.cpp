// Error plumbing and version for the C ABI (include/muscle_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void mx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mx_last_error(void) { return g_err; }
extern "C" int mx_version(void) { return 100; }
