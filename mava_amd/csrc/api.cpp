// libmavahip.so: error channel and version of the C ABI declared in include/mava_hip.h.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mava_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mava_last_error(void) { return g_err; }

extern "C" int mava_abi_version(void) { return 2; }  // 2: round-2 signatures (grad_scale, y_ld, packed acting step, comm, general layers)
