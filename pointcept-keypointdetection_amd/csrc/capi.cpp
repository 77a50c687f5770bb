// Error reporting + version of the C ABI (include/ptv3_hip.h). Kernels live in the .hip files.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ptv3_hip.h"

namespace ptv3 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ptv3

extern "C" const char* ptv3_last_error(void) { return ptv3::g_err; }
extern "C" int ptv3_version(void) { return 300; }
