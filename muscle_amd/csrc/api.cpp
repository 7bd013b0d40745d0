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
extern "C" int mx_version(void) { return 101; }
// Hash of include/muscle_hip.h at build time (muscle_amd/_build.py passes it): the Python binding derives its ctypes
// signatures from the header on disk and refuses to call a library built from a different one.
#ifndef MX_ABI_HASH
#define MX_ABI_HASH 0
#endif
extern "C" int mx_abi_hash(void) { return MX_ABI_HASH; }
