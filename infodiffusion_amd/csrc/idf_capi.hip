// Error channel + version for the C ABI (include/infodiff_hip.h).
#include "idf_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void idf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* idf_last_error(void) { return g_err; }
extern "C" int idf_version(void) { return 100; }
