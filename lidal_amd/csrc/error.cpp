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
extern "C" int lidal_version(void) { return 130; }   // 1.30 (round 3): own radix sort, batched maps / orders / pyramid, BatchNorm backward sums, 16-byte point<->voxel kernels, lidal_revoxelize_coords; built without packed f32 instructions
