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
extern "C" int lidal_version(void) { return 161; }   // 1.61 (round 6): the weight gradient on rule streams (lidal_wgrad_streams_build, lidal_conv_wgrad_streams, plan op 34), device-side item counts in sort.hip; 1.60 (round 6): data-gradient images in the split form (f32 training: lidal_conv_weight_image_job with LIDAL_F32_SPLIT and an img_bwd), the 32 x 32 x 16 MFMA forms of the split and lean kernels (LIDAL_SPLIT_MFMA=32, LIDAL_LEAN32=1); 1.41 (round 4): launch plans, row-wise helpers, conv offset split, NN grid cell in the header, block-tail BatchNorm sums, BatchNorm merges inside their consumers
