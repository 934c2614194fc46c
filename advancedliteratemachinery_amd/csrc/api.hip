// Error plumbing + ABI version for libomp355.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void omp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* omp_last_error(void) { return g_err; }
extern "C" int omp_abi_version(void) { return OMP_ABI_VERSION; }
