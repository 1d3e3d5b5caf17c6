// Point <-> voxel exchange kernels (count, voxelize, devoxelize, trilinear weights) for gfx950.
//
// All HBM-bound gathers / scatters over [rows, C] f32 feature matrices.  A thread owns a float4
// chunk of one row (C is a multiple of 4 on this path: 4, 32, 96, 128, 256), so every access is
// a 16-byte load/store and a row is covered by C/4 consecutive lanes.
#include <cstring>


#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

// 4 consecutive row elements as f32 (16-byte access for f32 rows, 8-byte for bf16 rows)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  bf16x4_t o;
  o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4_t*>(p) = o;
}

// VEC consecutive row elements as f32: one 16-byte access for 4 f32 or 8 bf16, 8 bytes for 4 bf16, scalar otherwise
template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&a)[VEC]) {
  if constexpr (VEC == 4) { float4 v = ld4(p); a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
  else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[v] = p[v];
  }
}
template <int VEC>
__device__ __forceinline__ void ldv(const __bf16* p, float (&a)[VEC]) {
  if constexpr (VEC == 8) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned q[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[2 * v] = __uint_as_float(q[v] << 16);
      a[2 * v + 1] = __uint_as_float(q[v] & 0xFFFF0000u);
    }
  } else if constexpr (VEC == 4) { float4 v = ld4(p); a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
  else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[v] = (float)p[v];
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&a)[VEC]) {
  if constexpr (VEC == 4) st4(p, make_float4(a[0], a[1], a[2], a[3]));
  else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) p[v] = a[v];
  }
}
template <int VEC>
__device__ __forceinline__ void stv(__bf16* p, const float (&a)[VEC]) {
  if constexpr (VEC == 8) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    bf16x8_t o;
#pragma unroll
    for (int v = 0; v < 8; ++v) o[v] = (__bf16)a[v];
    *reinterpret_cast<bf16x8_t*>(p) = o;
  } else if constexpr (VEC == 4) st4(p, make_float4(a[0], a[1], a[2], a[3]));
  else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) p[v] = (__bf16)a[v];
  }
}

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

// gin[i] = gout[idx[i]] / counts[idx[i]]  (+ res[i]: the gradient that reaches the same point rows
// through a second consumer, added here instead of by a separate accumulation pass)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) voxelize_bwd_kernel(const T* __restrict__ gout,
                                                           const int* __restrict__ idx,
                                                           const int* __restrict__ counts,
                                                           const T* __restrict__ res,
                                                           T* __restrict__ gin, int64_t n,
                                                           int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  int pos = idx[i];
  const bool dead = pos < 0 || pos >= m || counts[pos] == 0;
  const float div = dead ? 1.f : (float)counts[pos];
  float x[VEC], r[VEC];
  if (dead) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[v] = 0.f;
  } else {
    ldv<VEC>(gout + (int64_t)pos * c + j, x);
  }
  if (res != nullptr) ldv<VEC>(res + i * c + j, r);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    x[v] = x[v] / div;
    // the quotient is rounded to T first, as the stand-alone result would be, then the sum
    if (res != nullptr) x[v] = (float)(T)x[v] + r[v];
  }
  stv<VEC>(gin + i * c + j, x);
}

// 64 bytes of zeros: corners without a voxel or with an exactly zero weight read from here, so the
// eight corner loads of a point are unconditional and all in flight together (a `continue` around
// each load had serialised them)
__device__ __attribute__((aligned(16))) unsigned char g_zero_row[64];

// out[i] = sum_k w[i,k] * feat[idx[i,k]]
template <typename T, int VEC>
__global__ void __launch_bounds__(256) devoxelize_fwd_kernel(const T* __restrict__ feat,
                                                             const int* __restrict__ idx,
                                                             const float* __restrict__ w,
                                                             T* __restrict__ out, int64_t n,
                                                             int64_t m, int c) {
  const int cv = c / VEC;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  int64_t i = t / cv;
  int j = (int)(t - i * cv) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if constexpr (VEC >= 4) {
    // the point's 8 indices and weights as four 16-byte loads, then 8 row loads in flight
    const int4 p0 = *reinterpret_cast<const int4*>(idx + i * 8), p1 = *reinterpret_cast<const int4*>(idx + i * 8 + 4);
    const float4 w0 = *reinterpret_cast<const float4*>(w + i * 8), w1 = *reinterpret_cast<const float4*>(w + i * 8 + 4);
    const int pos[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float x[8][VEC];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // exact zeros: points on a cell face/corner (all of stride 1) -- skipped as upstream's sum
      // would add 0 * row (finite rows; the reference never holds inf/nan features here)
      const bool live = pos[k] >= 0 && pos[k] < m && wk[k] != 0.f;
      const T* src = live ? feat + (int64_t)pos[k] * c + j : reinterpret_cast<const T*>(g_zero_row);
      ldv<VEC>(src, x[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool live = pos[k] >= 0 && pos[k] < m && wk[k] != 0.f;
      if (live) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += wk[k] * x[k][v];
      }
    }
    stv<VEC>(out + i * c + j, acc);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int pos = idx[i * 8 + k];
      if (pos < 0 || pos >= m) continue;
      float wk = w[i * 8 + k];
      if (wk == 0.f) continue;
      const T* src = feat + (int64_t)pos * c + j;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += wk * (float)src[v];
    }
    T* dst = out + i * c + j;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = (T)acc[v];
  }
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
  // all eight corner indices first (eight loads in flight), through 32-bit element offsets
  int64_t qs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qs[k] = idx[(unsigned)k * (unsigned)n + (unsigned)i];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t q = qs[k];
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



// ---------------- inverse (contributor) lists: atomic-free, reproducible scatter sums ----------
// A point->voxel index idx[e] (e = point, or point*8+corner) is transposed once into per-voxel
// lists: `order` = entries sorted by voxel (stable radix sort => ascending e inside a voxel),
// seg_ptr[v] = first position of voxel v.  Scatter-adds then become per-voxel gathers of whole
// rows in a fixed order: no float atomics (chip-wide atomic rate is ~1.3 TB/s against ~5.5 TB/s
// for gathered rows) and bitwise reproducible sums.
__global__ void __launch_bounds__(256) inv_keys_kernel(const int* __restrict__ idx,
                                                       const float* __restrict__ w, int64_t n,
                                                       int64_t m, unsigned* __restrict__ keys,
                                                       int* __restrict__ vals) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int v = idx[e];
  bool ok = v >= 0 && v < m && (w == nullptr || w[e] != 0.f);
  keys[e] = ok ? (unsigned)v : (unsigned)m;
  vals[e] = (int)e;
}

__global__ void __launch_bounds__(256) inv_segptr_kernel(const unsigned* __restrict__ skeys,
                                                         int64_t n, int64_t m,
                                                         int64_t* __restrict__ seg_ptr) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > m) return;
  int64_t lo = 0, hi = n;                      // lower_bound(skeys, v)
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)skeys[mid] < v) lo = mid + 1; else hi = mid;
  }
  seg_ptr[v] = lo;
}

// out[v][:] = sum_{j in list(v)} scale(j) * src[row(j)][:]
//   voxelize:   row = e,      scale = 1 / counts[v]
//   devox bwd:  row = e >> 3, scale = w[e]
// Wave-level body: LPR lanes cover a row with VEC elements (16 bytes: 4 f32 / 8 bf16) each, the 64/LPR lane
// groups take list elements round-robin (4 independent row loads in flight per lane), fixed xor-tree at the end:
// lanes 0..LPR-1 end with sum_{j in [beg, end)} scale(j) * src[row(j)][VEC l .. VEC l + VEC - 1].
template <typename T, int LPR, int VEC, bool DEVOX>
__device__ __forceinline__ void segment_wave_sum(const T* __restrict__ src, const int* __restrict__ order,
                                                 const float* __restrict__ w, int64_t beg, int64_t end,
                                                 float inv, int c, int lane, float (&acc)[VEC]) {
  constexpr int RPW = 64 / LPR;
  const int l = lane % LPR, grp = lane / LPR;
  const bool act = VEC * l < c;
#pragma unroll
  for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
  auto add = [&](const float (&x)[VEC], float sc) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] += DEVOX ? sc * x[q] : x[q] / inv;
  };
  // 64 entries of the list per load, handed out by lane shuffle (round 5: the index load of a step no longer sits in
  // front of its row load); a lane group still takes entries beg + grp, beg + grp + RPW, ... in this order, four rows in
  // flight -- the sums of the earlier loop bit for bit.  The shuffles are wave-wide: the step counters do not depend on
  // the lane.
  for (int64_t base = beg; base < end; base += 64) {
    const int cnt = (int)((end - base < 64) ? (end - base) : 64);
    const int mine = lane < cnt ? order[base + lane] : 0;
    int ib = 0;
    for (; ib + 4 * RPW <= cnt; ib += 4 * RPW) {
      int e[4]; float sc[4]; float x[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = __shfl(mine, ib + grp + u * RPW, 64);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = DEVOX ? (e[u] >> 3) : e[u];
        sc[u] = DEVOX ? w[e[u]] : 1.f;
        if (act) ldv<VEC>(src + row * c + VEC * l, x[u]);
        else {
#pragma unroll
          for (int q = 0; q < VEC; ++q) x[u][q] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(x[u], sc[u]);
    }
    for (; ib < cnt; ib += RPW) {
      const bool ok = ib + grp < cnt;
      const int e = __shfl(mine, ok ? ib + grp : 0, 64);
      if (ok) {
        const int64_t row = DEVOX ? (e >> 3) : e;
        float x[VEC];
        if (act) ldv<VEC>(src + row * c + VEC * l, x);
        else {
#pragma unroll
          for (int q = 0; q < VEC; ++q) x[q] = 0.f;
        }
        add(x, DEVOX ? w[e] : 1.f);
      }
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
  }
}

// One wave per voxel (tens of contributors per voxel).
template <typename T, int LPR, int VEC, bool DEVOX>
__global__ void __launch_bounds__(256) segment_sum_kernel(const T* __restrict__ src,
                                                          const int* __restrict__ order,
                                                          const int64_t* __restrict__ seg_ptr,
                                                          const float* __restrict__ w,
                                                          const int* __restrict__ counts,
                                                          T* __restrict__ out, int64_t m,
                                                          int c) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const int l = lane % LPR, grp = lane / LPR;
  float inv = 1.f;
  if (!DEVOX) { int cv = counts[v]; inv = cv > 0 ? (float)cv : 1.f; }
  float acc[VEC];
  segment_wave_sum<T, LPR, VEC, DEVOX>(src, order, w, seg_ptr[v], seg_ptr[v + 1], inv, c, lane, acc);
  if (grp == 0 && VEC * l < c) stv<VEC>(out + v * c + VEC * l, acc);
}

// The coarse levels (hundreds of contributors per voxel, few voxels): `parts` workgroups per
// voxel, each wave sums a contiguous 1/(4*parts) of the list, the 4 waves are folded through LDS
// in wave order; with parts == 1 the result is final, otherwise it lands in an f32 partial row
// (v * parts + part) that segment_fold_kernel adds up in part order.  No atomics anywhere.
template <typename T, int LPR, int VEC, bool DEVOX>
__global__ void __launch_bounds__(256) segment_sum_wg_kernel(const T* __restrict__ src,
                                                             const int* __restrict__ order,
                                                             const int64_t* __restrict__ seg_ptr,
                                                             const float* __restrict__ w,
                                                             const int* __restrict__ counts,
                                                             T* __restrict__ out,
                                                             float* __restrict__ partial,
                                                             int64_t m, int c, int parts) {
  __shared__ float red[4][LPR][VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v = blockIdx.x / parts;
  const int part = (int)(blockIdx.x - v * parts);
  const int l = lane % LPR, grp = lane / LPR;
  const int64_t beg = seg_ptr[v], end = seg_ptr[v + 1];
  const int64_t chunk = (end - beg + 4 * parts - 1) / (4 * parts);
  int64_t b0 = beg + (int64_t)(part * 4 + wave) * chunk;
  int64_t b1 = b0 + chunk;
  if (b0 > end) b0 = end;
  if (b1 > end) b1 = end;
  float inv = 1.f;
  if (!DEVOX) { int cv = counts[v]; inv = cv > 0 ? (float)cv : 1.f; }
  float acc[VEC];
  segment_wave_sum<T, LPR, VEC, DEVOX>(src, order, w, b0, b1, inv, c, lane, acc);
  if (grp == 0) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) red[wave][l][q] = acc[q];
  }
  __syncthreads();
  if (wave == 0 && grp == 0 && VEC * l < c) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = red[0][l][q];
#pragma unroll
    for (int u = 1; u < 4; ++u) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] += red[u][l][q];
    }
    if (parts == 1) stv<VEC>(out + v * c + VEC * l, acc);
    else stv<VEC>(partial + ((int64_t)v * parts + part) * c + VEC * l, acc);
  }
}

// The fine levels (stride 1: every voxel has ONE contributor, the trilinear weights of integer points being
// (1, 0, .. 0); stride 2-4: a handful): a wave per voxel leaves most lanes idle and costs a wave launch per
// 64-256 bytes.  Here LPR lanes own a voxel -- 16 bytes of the row each -- and walk its list in order, four rows
// in flight; 64 / LPR voxels per wave.
template <typename T, int LPR, int VEC, bool DEVOX>
__global__ void __launch_bounds__(256) segment_sum_group_kernel(const T* __restrict__ src,
                                                                const int* __restrict__ order,
                                                                const int64_t* __restrict__ seg_ptr,
                                                                const float* __restrict__ w,
                                                                const int* __restrict__ counts,
                                                                T* __restrict__ out, int64_t m, int c) {
  const int l = threadIdx.x % LPR;
  const int64_t v = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  if (v >= m || l * VEC >= c) return;
  const int64_t beg = seg_ptr[v], end = seg_ptr[v + 1];
  float inv = 1.f;
  if (!DEVOX) { int cv = counts[v]; inv = cv > 0 ? (float)cv : 1.f; }
  float acc[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) acc[u] = 0.f;
  auto add = [&](const float (&x)[VEC], float sc) {
#pragma unroll
    for (int u = 0; u < VEC; ++u) acc[u] += DEVOX ? sc * x[u] : x[u] / inv;
  };
  int64_t j = beg;
  for (; j + 3 < end; j += 4) {
    int e[4]; float sc[4]; float x[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = order[j + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sc[u] = DEVOX ? w[e[u]] : 1.f;
      ldv<VEC>(src + (int64_t)(DEVOX ? (e[u] >> 3) : e[u]) * c + l * VEC, x[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add(x[u], sc[u]);
  }
  for (; j < end; ++j) {
    const int e = order[j];
    float x[VEC];
    ldv<VEC>(src + (int64_t)(DEVOX ? (e >> 3) : e) * c + l * VEC, x);
    add(x, DEVOX ? w[e] : 1.f);
  }
  stv<VEC>(out + v * c + l * VEC, acc);
}

template <typename T>
__global__ void __launch_bounds__(256) segment_fold_kernel(const float* __restrict__ partial,
                                                           T* __restrict__ out, int64_t m, int c,
                                                           int parts) {
  const int cv = c / 4;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * cv) return;
  int64_t v = t / cv;
  int j = (int)(t - v * cv) * 4;
  float4 a = ld4(partial + (v * parts) * c + j);
  for (int p = 1; p < parts; ++p) {
    float4 b = ld4(partial + (v * parts + p) * c + j);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  st4(out + v * c + j, a);
}

// Which form: a wave sums RPW = 64 / LPR rows per step, 4 steps in flight.  One wave per voxel (0) while the lists
// are short -- or while there are voxels enough to fill the chip with waves; the workgroup kernel otherwise, with
// enough workgroups per voxel that a wave sees ~4 rounds.  Measured on the bench batch (bf16, us; parts 0 / 1 / 2 / 4):
//   voxelize forward, stride 16, 16 730 voxels x 24 rows of 256:     158 /  87 /  95 / 136
//   devoxelize backward, stride 16, 16 730 voxels x ~170 rows of 256: 371 / 196 / 177 / 188
//   devoxelize backward, stride 4, 105 363 voxels x ~20 rows of 128:   81 / 149 / 282 / 520
static inline int row_lanes(int c, int vec) {        // lanes per row: the power of two that covers c / vec
  const int need = (c + vec - 1) / vec;
  int lpr = 4;
  while (lpr < need) lpr <<= 1;
  return lpr > 64 ? 64 : lpr;
}
static inline int segment_vec(int c, int esz) { return (esz == 2 && c % 8 == 0) ? 8 : 4; }
static inline int segment_parts(int64_t n_entries, int64_t m, int c, int esz) {
  if (m <= 0) return 0;
  const int lpr = row_lanes(c, segment_vec(c, esz));
  const int64_t rpw = 64 / lpr, avg = n_entries / m;
  if (avg < 20 * rpw && (m >= 65536 || avg < 8 * rpw)) return 0;
  const int64_t waves = (avg + 16 * rpw - 1) / (16 * rpw);
  int parts = 1;
  while (parts < 4 && (int64_t)parts * 4 < waves) parts <<= 1;
  return parts;
}

// Lists of at most this many entries per voxel (on average; an upper bound for devoxelize backward, whose zero
// weights are not in the lists) take the lane-group kernel -- on levels with many voxels.  A lane group walks its
// list alone, so the launch is as long as the longest list: at stride 16 (16 730 voxels, 24 points on average but
// hundreds in the cells next to the sensor) it took 241 us against the workgroup kernel's 106; at stride 4
// (105 363 voxels, <= 30 entries) 62 against the wave kernel's 89.
constexpr int64_t GROUP_MAX_AVG = 32, GROUP_MAX_AVG_FEW = 12, GROUP_MANY_VOXELS = 65536;

template <typename T, int VEC, bool DEVOX>
void launch_group(const T* src, const int* order, const int64_t* seg_ptr, const float* w, const int* counts,
                  T* out, int64_t m, int c, hipStream_t s) {
  const int need = (c + VEC - 1) / VEC;           // lanes per row
#define LIDAL_GROUP(LPR)                                                                               \
  segment_sum_group_kernel<T, LPR, VEC, DEVOX><<<(unsigned)cdiv(m, 256 / LPR), 256, 0, s>>>(src, order, seg_ptr, w, \
                                                                                          counts, out, m, c)
  if (need <= 4) LIDAL_GROUP(4);
  else if (need <= 8) LIDAL_GROUP(8);
  else if (need <= 16) LIDAL_GROUP(16);
  else if (need <= 32) LIDAL_GROUP(32);
  else LIDAL_GROUP(64);
#undef LIDAL_GROUP
}

template <typename T, bool DEVOX>
int launch_segment_sum(const T* src, const int* order, const int64_t* seg_ptr, const float* w,
                       const int* counts, T* out, int64_t m, int c, int64_t n_entries, void* ws,
                       int64_t ws_bytes, hipStream_t s) {
  // a row is covered by at most 64 lanes of VEC channels each
  LIDAL_REQUIRE(c <= 64 * segment_vec(c, (int)sizeof(T)), "segment_sum: at most %d channels per row for this element type",
                64 * segment_vec(c, (int)sizeof(T)));
  if (m > 0 && n_entries / m <= (m >= GROUP_MANY_VOXELS ? GROUP_MAX_AVG : GROUP_MAX_AVG_FEW)) {
    if (sizeof(T) == 2 && c % 8 == 0) launch_group<T, 8, DEVOX>(src, order, seg_ptr, w, counts, out, m, c, s);
    else launch_group<T, 4, DEVOX>(src, order, seg_ptr, w, counts, out, m, c, s);
    LIDAL_CHECK_LAUNCH("segment_sum_group");
    return 0;
  }
  const int parts = segment_parts(n_entries, m, c, (int)sizeof(T));
  float* partial = (float*)ws;
  if (parts > 1)
    LIDAL_REQUIRE(ws != nullptr && ws_bytes >= (int64_t)m * parts * c * 4, "segment_sum workspace too small");
  const bool wide = segment_vec(c, (int)sizeof(T)) == 8;
  const int lpr = row_lanes(c, wide ? 8 : 4);
#define LIDAL_SEG(LPR, VEC)                                                                              \
  do {                                                                                                   \
    if (parts == 0)                                                                                      \
      segment_sum_kernel<T, LPR, VEC, DEVOX><<<(unsigned)cdiv(m, 4), 256, 0, s>>>(src, order, seg_ptr, w, counts, \
                                                                                 out, m, c);             \
    else                                                                                                 \
      segment_sum_wg_kernel<T, LPR, VEC, DEVOX><<<(unsigned)(m * parts), 256, 0, s>>>(                   \
          src, order, seg_ptr, w, counts, out, partial, m, c, parts);                                    \
  } while (0)
  if (wide) {
    if (lpr <= 4) LIDAL_SEG(4, 8); else if (lpr == 8) LIDAL_SEG(8, 8); else if (lpr == 16) LIDAL_SEG(16, 8);
    else if (lpr == 32) LIDAL_SEG(32, 8); else LIDAL_SEG(64, 8);
  } else {
    if (lpr <= 8) LIDAL_SEG(8, 4); else if (lpr == 16) LIDAL_SEG(16, 4); else if (lpr == 32) LIDAL_SEG(32, 4);
    else LIDAL_SEG(64, 4);
  }
#undef LIDAL_SEG
  LIDAL_CHECK_LAUNCH("segment_sum");
  if (parts > 1) {
    segment_fold_kernel<T><<<(unsigned)cdiv(m * (c / 4), 256), 256, 0, s>>>(partial, out, m, c, parts);
    LIDAL_CHECK_LAUNCH("segment_fold");
  }
  return 0;
}

// ---- devoxelize backward through the CELLS (round 5; the coarse levels, where a voxel has hundreds of contributors) ----
// Every point of a cell -- the voxel its own coordinates floor to -- interpolates from the same eight corner voxels
// (voxel_to_point of network/utils.py: floor(pc / s) * s + {0, s}^3), only the weights differ.  So
//   stage 1  cs[cell][j][:] = sum over the cell's points p, in list order, of w[p][j] * g[p][:]          (f32)
//   stage 2  gin[v][:]      = sum over the (cell, j) whose corner j is v, in list order, of cs[cell][j][:]
// reads every gradient row ONCE (P rows) plus 8 f32 rows per cell written and read back, where the per-voxel lists of
// lidal_devoxelize_bwd_sorted gather every row 8 times: at stride 16 (16 730 cells of ~24 points, 256 channels) 0.48 GB
// instead of 1.46 GB.  Deterministic (fixed orders, no atomics); differs from the per-voxel form by the association of
// the f32 sums only.  Zero weights add nothing, as in the lists of the per-voxel form (which leave them out).
// One workgroup of four waves per cell, a contiguous quarter of the cell's list each (a cell next to the sensor holds
// hundreds of points: with two waves the launch was as long as that one list); a wave reads 64 entries of the list with ONE
// load and hands them out by lane shuffle (no dependent index load per step), four rows in flight per lane group.
template <typename T, int LPR, int VEC>
__global__ void __launch_bounds__(256) devox_cell_sums_kernel(const T* __restrict__ g, const int* __restrict__ vorder,
                                                              const int64_t* __restrict__ vseg,
                                                              const float* __restrict__ w8, float* __restrict__ cs,
                                                              int64_t m, int c) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[3][8][LPR * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane % LPR, grp = lane / LPR;
  const int64_t cell = blockIdx.x;
  const int64_t beg = vseg[cell], end = vseg[cell + 1];
  const int64_t quarter = (end - beg + 3) / 4;
  int64_t b0 = beg + (int64_t)wave * quarter, b1 = b0 + quarter;
  if (b0 > end) b0 = end;
  if (b1 > end) b1 = end;
  float acc[8][VEC];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[k][q] = 0.f;
  auto add = [&](const float (&x)[VEC], const float4& wa, const float4& wb) {
    const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (wv[k] != 0.f) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[k][q] += wv[k] * x[q];
      }
  };
  for (int64_t base = b0; base < b1; base += 64) {
    const int cnt = (int)((b1 - base < 64) ? (b1 - base) : 64);
    const int mine = lane < cnt ? vorder[base + lane] : 0;
    int ib = 0;                                  // first entry of the step (the same in every lane: the shuffles are wave-wide)
    for (; ib + 4 * RPW <= cnt; ib += 4 * RPW) {
      const int i = ib + grp;
      int p[4]; float x[4][VEC]; float4 wa[4], wb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = __shfl(mine, i + u * RPW, 64);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ldv<VEC>(g + (int64_t)p[u] * c + VEC * l, x[u]);
        wa[u] = ld4(w8 + (int64_t)p[u] * 8); wb[u] = ld4(w8 + (int64_t)p[u] * 8 + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(x[u], wa[u], wb[u]);
    }
    // (every lane takes part in the shuffle; rows past the count add nothing)
    for (; ib < cnt; ib += RPW) {
      const int i = ib + grp;
      const bool ok = i < cnt;
      const int p0 = __shfl(mine, ok ? i : 0, 64);
      if (ok) {
        float x0[VEC];
        ldv<VEC>(g + (int64_t)p0 * c + VEC * l, x0);
        add(x0, ld4(w8 + (int64_t)p0 * 8), ld4(w8 + (int64_t)p0 * 8 + 4));
      }
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[k][q] += __shfl_xor(acc[k][q], off, 64);
  }
  if (wave > 0 && grp == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < VEC; ++q) red[wave - 1][k][l * VEC + q] = acc[k][q];
  }
  __syncthreads();
  if (wave == 0 && grp == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float* dst = cs + ((int64_t)cell * 8 + k) * c + VEC * l;
#pragma unroll
      for (int q = 0; q < VEC; q += 4) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = ((acc[k][q + e] + red[0][k][l * VEC + q + e]) + red[1][k][l * VEC + q + e]) + red[2][k][l * VEC + q + e];
        st4(dst + q, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) devox_cell_fold_kernel(const float* __restrict__ cs, const int* __restrict__ corder,
                                                              const int64_t* __restrict__ cseg, T* __restrict__ gin,
                                                              int64_t m, int c) {
  const int cv = c / 4;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * cv) return;
  const int64_t v = t / cv;
  const int jj = (int)(t - v * cv) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = cseg[v]; i < cseg[v + 1]; ++i) {
    const float4 b = ld4(cs + (int64_t)corder[i] * c + jj);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  st4(gin + v * c + jj, a);
}

template <typename T>
int launch_devox_cells(const T* g, const int* vorder, const int64_t* vseg, const float* w8, const int* corder,
                       const int64_t* cseg, T* gin, int64_t m, int c, float* cs, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int lpr = c / VEC;
#define LIDAL_CELLS(LPR) devox_cell_sums_kernel<T, LPR, VEC><<<(unsigned)m, 256, 0, s>>>(g, vorder, vseg, w8, cs, m, c)
  if (lpr == 4) LIDAL_CELLS(4); else if (lpr == 8) LIDAL_CELLS(8); else if (lpr == 16) LIDAL_CELLS(16);
  else if (lpr == 32) LIDAL_CELLS(32); else LIDAL_CELLS(64);
#undef LIDAL_CELLS
  LIDAL_CHECK_LAUNCH("devoxelize_bwd_cells(sums)");
  devox_cell_fold_kernel<T><<<(unsigned)cdiv(m * (c / 4), 256), 256, 0, s>>>(cs, corder, cseg, gin, m, c);
  LIDAL_CHECK_LAUNCH("devoxelize_bwd_cells(fold)");
  return 0;
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

#define DISPATCH_TVEC(kernel, T, c, total_rows, ...)                                         \
  do {                                                                                       \
    if (sizeof(T) == 2 && (c) % 8 == 0) {     /* 16-byte accesses on bf16 rows */             \
      int64_t t__ = (total_rows) * ((c) / 8);                                                \
      kernel<T, 8><<<(unsigned)cdiv(t__, 256), 256, 0, s>>>(__VA_ARGS__);                    \
    } else if ((c) % 4 == 0) {                                                                      \
      int64_t t__ = (total_rows) * ((c) / 4);                                                \
      kernel<T, 4><<<(unsigned)cdiv(t__, 256), 256, 0, s>>>(__VA_ARGS__);                    \
    } else {                                                                                 \
      int64_t t__ = (total_rows) * (c);                                                      \
      kernel<T, 1><<<(unsigned)cdiv(t__, 256), 256, 0, s>>>(__VA_ARGS__);                    \
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

// Every voxel has exactly ONE point (LiDAL: the dataset already voxelised the scan, network/spvcnn.py:114 with
// pres == vres): the mean over a voxel's points is the point row itself, `out[idx[i]] = feat[i]` -- a row
// permutation, no contributor lists (their sort) and no wave per voxel.  idx must be a permutation of 0..n-1
// (the caller knows: as many distinct voxel hashes as points).
namespace {
template <typename T, int VEC>
__global__ void __launch_bounds__(256) voxelize_1to1_kernel(const T* __restrict__ feat, const int* __restrict__ idx,
                                                            T* __restrict__ out, int64_t n, int c) {
  const int cv = c / VEC;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cv) return;
  const int64_t i = t / cv;
  const int j = (int)(t - i * cv) * VEC;
  const int pos = idx[i];
  if (pos < 0 || pos >= n) return;
  const T* src = feat + i * c + j;
  T* dst = out + (int64_t)pos * c + j;
#pragma unroll
  for (int v = 0; v < VEC; ++v) dst[v] = src[v];
}
}  // namespace

extern "C" int lidal_voxelize_fwd_1to1(const void* feat, const int32_t* idx, void* out, int64_t n, int c,
                                       int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || c == 0) return 0;
  if (dtype == LIDAL_F32)
    DISPATCH_TVEC(voxelize_1to1_kernel, float, c, n, (const float*)feat, idx, (float*)out, n, c);
  else if (dtype == LIDAL_BF16)
    DISPATCH_TVEC(voxelize_1to1_kernel, __bf16, c, n, (const __bf16*)feat, idx, (__bf16*)out, n, c);
  else { set_error("voxelize_fwd_1to1: bad dtype %d", dtype); return 2; }
  LIDAL_CHECK_LAUNCH("lidal_voxelize_fwd_1to1");
  return 0;
}

extern "C" int lidal_voxelize_bwd(const void* gout, const int32_t* idx, const int32_t* counts,
                                  const void* residual, void* gin, int64_t n, int64_t m, int c,
                                  int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || c == 0) return 0;
  if (dtype == LIDAL_F32)
    DISPATCH_TVEC(voxelize_bwd_kernel, float, c, n, (const float*)gout, idx, counts,
                  (const float*)residual, (float*)gin, n, m, c);
  else if (dtype == LIDAL_BF16)
    DISPATCH_TVEC(voxelize_bwd_kernel, __bf16, c, n, (const __bf16*)gout, idx, counts,
                  (const __bf16*)residual, (__bf16*)gin, n, m, c);
  else { set_error("voxelize_bwd: bad dtype %d", dtype); return 2; }
  LIDAL_CHECK_LAUNCH("lidal_voxelize_bwd");
  return 0;
}

extern "C" int lidal_devoxelize_fwd(const void* feat, const int32_t* idx, const float* w,
                                    void* out, int64_t n, int64_t m, int c, int dtype,
                                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || c == 0) return 0;
  if (dtype == LIDAL_F32)
    DISPATCH_TVEC(devoxelize_fwd_kernel, float, c, n, (const float*)feat, idx, w, (float*)out, n, m, c);
  else if (dtype == LIDAL_BF16)
    DISPATCH_TVEC(devoxelize_fwd_kernel, __bf16, c, n, (const __bf16*)feat, idx, w, (__bf16*)out, n, m, c);
  else { set_error("devoxelize_fwd: bad dtype %d", dtype); return 2; }
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

static int64_t inv_tmp_bytes(int64_t q) { return sort_pairs_ws_bytes(q); }

extern "C" int64_t lidal_invlist_workspace_bytes(int64_t n_entries) {
  int64_t q = n_entries > 0 ? n_entries : 1;
  return 2 * align_up(4 * q, 256) + align_up(4 * q, 256) + align_up(inv_tmp_bytes(q), 256) + 256;
}

extern "C" int lidal_invlist_build(const int32_t* idx, const float* w, int64_t n_entries, int64_t m,
                                   int32_t* order, int64_t* seg_ptr, void* ws, int64_t ws_bytes,
                                   void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(m >= 0 && m < 0x7FFFFFFF, "invlist: bad voxel count");
  if (n_entries == 0) {
    LIDAL_HIP(hipMemsetAsync(seg_ptr, 0, 8 * (m + 1), s));
    return 0;
  }
  LIDAL_REQUIRE(ws_bytes >= lidal_invlist_workspace_bytes(n_entries), "invlist workspace too small");
  int64_t q = n_entries;
  unsigned* keys = (unsigned*)ws;
  unsigned* skeys = (unsigned*)((char*)ws + align_up(4 * q, 256));
  int* vals = (int*)((char*)ws + 2 * align_up(4 * q, 256));
  void* tmp = (char*)ws + 3 * align_up(4 * q, 256);
  inv_keys_kernel<<<(unsigned)cdiv(q, 256), 256, 0, s>>>(idx, w, q, m, keys, vals);
  LIDAL_CHECK_LAUNCH("inv_keys");
  int bits = 1;
  while ((1ll << bits) <= m) ++bits;
  if (int rc = sort_pairs_u32(keys, vals, skeys, order, q, bits, tmp, inv_tmp_bytes(q), s)) return rc;
  inv_segptr_kernel<<<(unsigned)cdiv(m + 1, 256), 256, 0, s>>>(skeys, q, m, seg_ptr);
  LIDAL_CHECK_LAUNCH("inv_segptr");
  return 0;
}

extern "C" int64_t lidal_segment_workspace_bytes(int64_t n_entries, int64_t m, int c) {
  // (f32 rows never need less than bf16 rows: the split is chosen per element size, the larger one is reported)
  int parts = segment_parts(n_entries, m, c, 2), parts4 = segment_parts(n_entries, m, c, 4);
  if (parts4 > parts) parts = parts4;
  return parts > 1 ? (int64_t)m * parts * c * 4 : 0;
}

extern "C" int lidal_voxelize_fwd_sorted(const void* feat, const int32_t* order,
                                         const int64_t* seg_ptr, const int32_t* counts, void* out,
                                         int64_t m, int c, int dtype, int64_t n_entries, void* ws,
                                         int64_t ws_bytes, void* stream) {
  if (m == 0 || c == 0) return 0;
  LIDAL_REQUIRE(c % 4 == 0, "voxelize_fwd_sorted: channels must be a multiple of 4");
  if (dtype == LIDAL_F32)
    return launch_segment_sum<float, false>((const float*)feat, order, seg_ptr, nullptr, counts,
                                            (float*)out, m, c, n_entries, ws, ws_bytes,
                                            (hipStream_t)stream);
  if (dtype == LIDAL_BF16)
    return launch_segment_sum<__bf16, false>((const __bf16*)feat, order, seg_ptr, nullptr, counts,
                                             (__bf16*)out, m, c, n_entries, ws, ws_bytes,
                                             (hipStream_t)stream);
  set_error("voxelize_fwd_sorted: bad dtype %d", dtype);
  return 2;
}

// devoxelize backward through the cells (see devox_cell_sums_kernel): vorder / vseg = the points of every cell (the
// inverse lists of the points' own voxel index, idx8[:, 0]), w8 [P, 8] the trilinear weights, corder / cseg = for every
// voxel the entries cell * 8 + j whose corner j it is (the inverse lists of the cells' [m, 8] corner indices).  c: whole
// 16-byte steps per row, a power of two of them up to 64 (bf16: 32 .. 512 channels).  ws >= m * 8 * c * 4 bytes.
extern "C" int64_t lidal_devoxelize_bwd_cells_workspace_bytes(int64_t m, int c) { return m * 8 * (int64_t)c * 4; }
extern "C" int lidal_devoxelize_bwd_cells(const void* gout, const int32_t* vorder, const int64_t* vseg, const float* w8,
                                          const int32_t* corder, const int64_t* cseg, void* gin, int64_t m, int c,
                                          int dtype, void* ws, int64_t ws_bytes, void* stream) {
  if (m == 0 || c == 0) return 0;
  const int vec = dtype == LIDAL_F32 ? 4 : 8;
  const int lpr = c / vec;
  LIDAL_REQUIRE(dtype == LIDAL_F32 || dtype == LIDAL_BF16, "devoxelize_bwd_cells: bad dtype %d", dtype);
  LIDAL_REQUIRE(c % vec == 0 && lpr >= 4 && lpr <= 64 && (lpr & (lpr - 1)) == 0,
                "devoxelize_bwd_cells: %d channels (a power of two of 16-byte steps, 4 .. 64 of them)", c);
  LIDAL_REQUIRE(ws != nullptr && ws_bytes >= lidal_devoxelize_bwd_cells_workspace_bytes(m, c), "devoxelize_bwd_cells ws too small");
  if (dtype == LIDAL_F32)
    return launch_devox_cells<float>((const float*)gout, vorder, vseg, w8, corder, cseg, (float*)gin, m, c, (float*)ws,
                                     (hipStream_t)stream);
  return launch_devox_cells<__bf16>((const __bf16*)gout, vorder, vseg, w8, corder, cseg, (__bf16*)gin, m, c, (float*)ws,
                                    (hipStream_t)stream);
}

extern "C" int lidal_devoxelize_bwd_sorted(const void* gout, const int32_t* order,
                                           const int64_t* seg_ptr, const float* w, void* gin,
                                           int64_t m, int c, int dtype, int64_t n_entries, void* ws,
                                           int64_t ws_bytes, void* stream) {
  if (m == 0 || c == 0) return 0;
  LIDAL_REQUIRE(c % 4 == 0, "devoxelize_bwd_sorted: channels must be a multiple of 4");
  if (dtype == LIDAL_F32)
    return launch_segment_sum<float, true>((const float*)gout, order, seg_ptr, w, nullptr,
                                           (float*)gin, m, c, n_entries, ws, ws_bytes,
                                           (hipStream_t)stream);
  if (dtype == LIDAL_BF16)
    return launch_segment_sum<__bf16, true>((const __bf16*)gout, order, seg_ptr, w, nullptr,
                                            (__bf16*)gin, m, c, n_entries, ws, ws_bytes,
                                            (hipStream_t)stream);
  set_error("devoxelize_bwd_sorted: bad dtype %d", dtype);
  return 2;
}
