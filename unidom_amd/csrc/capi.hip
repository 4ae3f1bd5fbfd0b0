// libunidom_hip: error channel + version of the C ABI (include/unidom_hip.h).
#include "common.h"

namespace ud {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ud

extern "C" {
const char* ud_last_error(void) { return ud::g_err; }
const char* ud_version(void) { return "0.1.0 gfx950"; }
}
