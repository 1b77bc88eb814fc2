// Error reporting + ABI version for libmla_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mla_hip.h"

static thread_local char g_err[512] = "";

void mla_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mla_abi_version(void) { return 3; }
extern "C" const char* mla_last_error(void) { return g_err; }
