#include <stdarg.h>
#include "common.h"
#include "../../include/seedhip.h"

namespace seedhip {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace seedhip

extern "C" const char* seedhip_last_error(void) { return seedhip::err_buf(); }
extern "C" int seedhip_abi_version(void) { return SEEDHIP_ABI_VERSION; }
