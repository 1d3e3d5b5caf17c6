// Thread-local error string + version for the C-ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lidal_amd.h"

namespace lidal {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lidal

extern "C" const char* lidal_last_error(void) { return lidal::g_err; }
extern "C" int lidal_version(void) { return 141; }   // 1.41 (round 4): launch plans (lidal_plan_run), row-wise helpers, conv offset split (_ws entry points), NN grid cell in the header, block-tail BatchNorm sums, BatchNorm merges inside their consumers; the first-generation convolution left the library
