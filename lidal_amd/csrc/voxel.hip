// Point <-> voxel exchange kernels (count, voxelize, devoxelize, trilinear weights) for gfx950.
//
// All HBM-bound gathers / scatters over [rows, C] f32 feature matrices.  A thread owns a float4
// chunk of one row (C is a multiple of 4 on this path: 4, 32, 96, 128, 256), so every access is
// a 16-byte load/store and a row is covered by C/4 consecutive lanes.
#include "common.h"

using namespace lidal;

namespace {

__global__ void __launch_bounds__(256) count_kernel(const int* __restrict__ idx, int64_t n,
                                                    int* __restrict__ out, int64_t m) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = idx[i];
  if (v >= 0 && v < m) atomicAdd(&out[v], 1);
}

// out[idx[i]] += feat[i] / counts[idx[i]]   (mean pooling; out pre-zeroed)
template <int VEC>
__global__ void __launch_bounds__(256) voxelize_fwd_kernel(const float* __restrict__ feat,
                                                           const int* __restrict__ idx,
                                                           const int* __restrict__ counts,
                                                           float* __restrict__ out, int64_t n,
                                                           int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  int pos = idx[i];
  if (pos < 0 || pos >= m) return;
  int cnt = counts[pos];
  if (cnt == 0) return;
  float div = (float)cnt;
  const float* src = feat + i * c + j;
  float* dst = out + (int64_t)pos * c + j;
  if (cnt == 1) {          // sole contributor: plain store, no atomic
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = src[v] / div;
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) atomicAdd(&dst[v], src[v] / div);
  }
}

// gin[i] = gout[idx[i]] / counts[idx[i]]
template <int VEC>
__global__ void __launch_bounds__(256) voxelize_bwd_kernel(const float* __restrict__ gout,
                                                           const int* __restrict__ idx,
                                                           const int* __restrict__ counts,
                                                           float* __restrict__ gin, int64_t n,
                                                           int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  int pos = idx[i];
  float* dst = gin + i * c + j;
  if (pos < 0 || pos >= m || counts[pos] == 0) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = 0.f;
    return;
  }
  float div = (float)counts[pos];
  const float* src = gout + (int64_t)pos * c + j;
#pragma unroll
  for (int v = 0; v < VEC; ++v) dst[v] = src[v] / div;
}

// out[i] = sum_k w[i,k] * feat[idx[i,k]]
template <int VEC>
__global__ void __launch_bounds__(256) devoxelize_fwd_kernel(const float* __restrict__ feat,
                                                             const int* __restrict__ idx,
                                                             const float* __restrict__ w,
                                                             float* __restrict__ out, int64_t n,
                                                             int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int pos = idx[i * 8 + k];
    if (pos < 0 || pos >= m) continue;
    float wk = w[i * 8 + k];
    if (wk == 0.f) continue;        // exact zeros: points on a cell face/corner (all of stride 1)
    const float* src = feat + (int64_t)pos * c + j;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += wk * src[v];
  }
  float* dst = out + i * c + j;
#pragma unroll
  for (int v = 0; v < VEC; ++v) dst[v] = acc[v];
}

// gin[idx[i,k]] += w[i,k] * gout[i]    (gin pre-zeroed)
template <int VEC>
__global__ void __launch_bounds__(256) devoxelize_bwd_kernel(const float* __restrict__ gout,
                                                             const int* __restrict__ idx,
                                                             const float* __restrict__ w,
                                                             float* __restrict__ gin, int64_t n,
                                                             int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  float g[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) g[v] = gout[i * c + j + v];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int pos = idx[i * 8 + k];
    if (pos < 0 || pos >= m) continue;
    float wk = w[i * 8 + k];
    if (wk == 0.f) continue;
    float* dst = gin + (int64_t)pos * c + j;
#pragma unroll
    for (int v = 0; v < VEC; ++v) atomicAdd(&dst[v], wk * g[v]);
  }
}

// torchsparse calc_ti_weights, fused with the two transposes of network/utils.py:78-79.
__global__ void __launch_bounds__(256) ti_weights_kernel(const float* __restrict__ coords,
                                                         int cstride,
                                                         const int64_t* __restrict__ idx,
                                                         int64_t n, float scale,
                                                         float* __restrict__ w,
                                                         int* __restrict__ idx32) {
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
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int64_t q = idx[(int64_t)k * n + i];
    id[k] = (int)q;
    if (scale != 1.f) ww[k] /= s3;
    if (q == -1) ww[k] = 0.f;
    sum += ww[k];                 // torch.sum over dim 0 of an [8,n] tensor: k ascending
  }
  sum += 1e-8f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    w[i * 8 + k] = ww[k] / sum;
    idx32[i * 8 + k] = id[k];
  }
}


}  // namespace

#define DISPATCH_VEC(kernel, c, total_rows, ...)                                             \
  do {                                                                                       \
    if ((c) % 4 == 0) {                                                                      \
      int64_t t__ = (total_rows) * ((c) / 4);                                                \
      kernel<4><<<(unsigned)cdiv(t__, 256), 256, 0, s>>>(__VA_ARGS__);                       \
    } else {                                                                                 \
      int64_t t__ = (total_rows) * (c);                                                      \
      kernel<1><<<(unsigned)cdiv(t__, 256), 256, 0, s>>>(__VA_ARGS__);                       \
    }                                                                                        \
  } while (0)

extern "C" int lidal_count(const int32_t* idx, int64_t n, int32_t* out, int64_t m, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (m > 0) LIDAL_HIP(hipMemsetAsync(out, 0, 4 * m, s));
  if (n == 0 || m == 0) return 0;
  count_kernel<<<(unsigned)cdiv(n, 256), 256, 0, s>>>(idx, n, out, m);
  LIDAL_CHECK_LAUNCH("lidal_count");
  return 0;
}

extern "C" int lidal_voxelize_fwd(const float* feat, const int32_t* idx, const int32_t* counts,
                                  float* out, int64_t n, int64_t m, int c, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (m > 0 && c > 0) LIDAL_HIP(hipMemsetAsync(out, 0, 4 * m * c, s));
  if (n == 0 || m == 0 || c == 0) return 0;
  DISPATCH_VEC(voxelize_fwd_kernel, c, n, feat, idx, counts, out, n, m, c);
  LIDAL_CHECK_LAUNCH("lidal_voxelize_fwd");
  return 0;
}

extern "C" int lidal_voxelize_bwd(const float* gout, const int32_t* idx, const int32_t* counts,
                                  float* gin, int64_t n, int64_t m, int c, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || c == 0) return 0;
  DISPATCH_VEC(voxelize_bwd_kernel, c, n, gout, idx, counts, gin, n, m, c);
  LIDAL_CHECK_LAUNCH("lidal_voxelize_bwd");
  return 0;
}

extern "C" int lidal_devoxelize_fwd(const float* feat, const int32_t* idx, const float* w,
                                    float* out, int64_t n, int64_t m, int c, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || c == 0) return 0;
  DISPATCH_VEC(devoxelize_fwd_kernel, c, n, feat, idx, w, out, n, m, c);
  LIDAL_CHECK_LAUNCH("lidal_devoxelize_fwd");
  return 0;
}

extern "C" int lidal_devoxelize_bwd(const float* gout, const int32_t* idx, const float* w,
                                    float* gin, int64_t n, int64_t m, int c, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (m > 0 && c > 0) LIDAL_HIP(hipMemsetAsync(gin, 0, 4 * m * c, s));
  if (n == 0 || m == 0 || c == 0) return 0;
  DISPATCH_VEC(devoxelize_bwd_kernel, c, n, gout, idx, w, gin, n, m, c);
  LIDAL_CHECK_LAUNCH("lidal_devoxelize_bwd");
  return 0;
}

extern "C" int lidal_ti_weights(const float* coords, int cstride, const int64_t* idx, int64_t n,
                                float scale, float* w, int32_t* idx32, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return 0;
  LIDAL_REQUIRE(cstride >= 3, "ti_weights: coords need >= 3 columns");
  ti_weights_kernel<<<(unsigned)cdiv(n, 256), 256, 0, s>>>(coords, cstride, idx, n, scale, w, idx32);
  LIDAL_CHECK_LAUNCH("lidal_ti_weights");
  return 0;
}
