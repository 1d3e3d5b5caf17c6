// lidal_amd/csrc/voxel.hip's ti_weights_kernel with a diagnostic record per (point, corner): the index the kernel
// saw (both halves), the weight before and after the `idx == -1 -> 0` select.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void __launch_bounds__(256) tiw_dbg_kernel(const float* __restrict__ coords, int cstride,
                                                                 const int64_t* __restrict__ idx, int64_t n, float scale,
                                                                 float* __restrict__ w, int* __restrict__ idx32,
                                                                 uint4* __restrict__ dbg) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = coords[i * cstride + 0], y = coords[i * cstride + 1], z = coords[i * cstride + 2];
  float xf, yf, zf;
  if (scale != 1.f) {
    xf = floorf(x / scale) * scale; yf = floorf(y / scale) * scale; zf = floorf(z / scale) * scale;
  } else {
    xf = floorf(x); yf = floorf(y); zf = floorf(z);
  }
  float xc = xf + scale, yc = yf + scale, zc = zf + scale;
  float ww[8];
  ww[0] = (xc - x) * (yc - y) * (zc - z);
  ww[1] = (xc - x) * (yc - y) * (z - zf);
  ww[2] = (xc - x) * (y - yf) * (zc - z);
  ww[3] = (xc - x) * (y - yf) * (z - zf);
  ww[4] = (x - xf) * (yc - y) * (zc - z);
  ww[5] = (x - xf) * (yc - y) * (z - zf);
  ww[6] = (x - xf) * (y - yf) * (zc - z);
  ww[7] = (x - xf) * (y - yf) * (z - zf);
  float s3 = scale * scale * scale;
  float sum = 0.f;
  int id[8];
  int64_t qs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qs[k] = idx[(unsigned)k * (unsigned)n + (unsigned)i];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t q = qs[k];
    id[k] = (int)q;
    if (scale != 1.f) ww[k] /= s3;
    const float before = ww[k];
    if (q == -1) ww[k] = 0.f;
    dbg[i * 8 + k] = make_uint4((unsigned)q, (unsigned)((uint64_t)q >> 32), __float_as_uint(before), __float_as_uint(ww[k]));
    sum += ww[k];
  }
  sum += 1e-8f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    w[i * 8 + k] = ww[k] / sum;
    idx32[i * 8 + k] = id[k];
  }
}

extern "C" int tiw_dbg_launch(const float* coords, int cstride, const int64_t* idx, int64_t n, float scale, float* w,
                              int* idx32, void* dbg, void* stream) {
  tiw_dbg_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(coords, cstride, idx, n, scale, w, idx32,
                                                                              (uint4*)dbg);
  return (int)hipGetLastError();
}
