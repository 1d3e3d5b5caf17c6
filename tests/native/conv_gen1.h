/* conv_gen1.h -- C-ABI of tests/native/liblidal_gen1.so: the first-generation fused sparse convolution (rounds
 * 1-3 of liblidal_amd.so), kept OUT of the product library as an independent implementation for bitwise
 * cross-checks of lidal_conv_apply_image (tests/test_ops_gpu.py).  Conventions as include/lidal_amd.h. */
#ifndef LIDAL_CONV_GEN1_H
#define LIDAL_CONV_GEN1_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Weight re-layout (+ optional cast): W [k][ci][co] -> Wt [k][co][ci] in wt_dtype; wc (may be NULL) receives W cast to
 * wt_dtype in the original layout. */
int lidal_conv_weight_pack(const void* w, int w_dtype, void* wt, void* wc, int wt_dtype, int k, int ci, int co,
                           void* stream);
/* out[row(j), :] = sum_k in[nbr[kk][j], :] * Wk[k]^T with Wk laid out [k][co][ci]; arguments as
 * lidal_conv_apply_image (include/lidal_amd.h) without the tile statistics. */
int lidal_conv_apply(const void* in, const void* wk, const int32_t* nbr, const int32_t* perm,
                     const uint32_t* tile_masks, void* out, int64_t n_in, int64_t n_out, int ci, int co, int k,
                     int kflip, int dtype, const float* ep_scale, const float* ep_shift, int ep_relu,
                     const void* ep_residual, void* stream);
#ifdef __cplusplus
}
#endif
#endif
