// Error reporting and version of the C ABI (include/nvt_hip.h).
#include <cstdarg>

#include "nvt_common.hpp"

namespace nvt {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace nvt

extern "C" {
int nvt_version(void) { return 100; }  // 0.1.0
const char *nvt_last_error(void) { return nvt::g_err; }
}
