// Error reporting and ABI version of libwanq_hip.
#include "wanq_common.h"

namespace wanq {
static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace wanq

extern "C" const char* wanq_last_error(void) { return wanq::g_err; }
extern "C" int wanq_abi_version(void) { return 6; }
