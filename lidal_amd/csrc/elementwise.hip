// Small fused row-wise ops of the training step for gfx950:
//   * add + ReLU of the residual blocks (network/utils.py:171), forward and backward;
//   * cross-entropy with ignore_index and mean reduction (train.py:136): forward (per-block loss
//     partials, fixed-order final sum) and backward (softmax - onehot) / n_valid in one pass each.
// All HBM-bound; 16-byte lane accesses where rows allow, f32 arithmetic.
#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <typename T> struct EW;
template <> struct EW<float> {
  static constexpr int VEC = 4;
  typedef float4 vec;
  __device__ static void unpack(const vec& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  __device__ static vec pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct EW<__bf16> {
  static constexpr int VEC = 8;
  typedef bf16x8_t vec;
  __device__ static void unpack(const vec& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  __device__ static vec pack(const float (&f)[8]) {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    return v;
  }
};

// y = max(a + b, 0)                 (n = number of VEC-wide chunks)
template <typename T>
__global__ void __launch_bounds__(256) add_relu_fwd_kernel(const T* __restrict__ a,
                                                           const T* __restrict__ b,
                                                           T* __restrict__ y, int64_t n) {
  constexpr int VEC = EW<T>::VEC;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    float fa[VEC], fb[VEC];
    EW<T>::unpack(reinterpret_cast<const typename EW<T>::vec*>(a)[i], fa);
    EW<T>::unpack(reinterpret_cast<const typename EW<T>::vec*>(b)[i], fb);
#pragma unroll
    for (int e = 0; e < VEC; ++e) fa[e] = fmaxf(fa[e] + fb[e], 0.f);
    reinterpret_cast<typename EW<T>::vec*>(y)[i] = EW<T>::pack(fa);
  }
}

// g_in = g * (y > 0)  (the same tensor is the gradient of both summands)
template <typename T>
__global__ void __launch_bounds__(256) add_relu_bwd_kernel(const T* __restrict__ y,
                                                           const T* __restrict__ g,
                                                           T* __restrict__ gin, int64_t n) {
  constexpr int VEC = EW<T>::VEC;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    float fy[VEC], fg[VEC];
    EW<T>::unpack(reinterpret_cast<const typename EW<T>::vec*>(y)[i], fy);
    EW<T>::unpack(reinterpret_cast<const typename EW<T>::vec*>(g)[i], fg);
#pragma unroll
    for (int e = 0; e < VEC; ++e) fg[e] = fy[e] > 0.f ? fg[e] : 0.f;
    reinterpret_cast<typename EW<T>::vec*>(gin)[i] = EW<T>::pack(fg);
  }
}

constexpr int CE_MAXC = 32;

// per-row -log softmax(logits)[label]; block partial sums of loss and of the valid-row count
template <typename T>
__global__ void __launch_bounds__(256) ce_fwd_kernel(const T* __restrict__ logits,
                                                     const int64_t* __restrict__ labels, int64_t n,
                                                     int c, int64_t ignore_index,
                                                     float* __restrict__ part /*[blocks][2]*/) {
  __shared__ float red[2][256];
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (i < n) {
    const int64_t y = labels[i];
    if (y != ignore_index && y >= 0 && y < c) {
      const T* row = logits + i * c;
      float x[CE_MAXC], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < CE_MAXC; ++j)
        if (j < c) { x[j] = (float)row[j]; mx = fmaxf(mx, x[j]); }
      float s = 0.f, xy = 0.f;
#pragma unroll
      for (int j = 0; j < CE_MAXC; ++j)
        if (j < c) { s += expf(x[j] - mx); if (j == (int)y) xy = x[j]; }
      loss = logf(s) + mx - xy;
      cnt = 1.f;
    }
  }
  red[0][threadIdx.x] = loss; red[1][threadIdx.x] = cnt;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = red[0][0]; part[blockIdx.x * 2 + 1] = red[1][0]; }
}

// out[0] = sum(loss) / n_valid, out[1] = n_valid        (single block, fixed order)
__global__ void __launch_bounds__(256) ce_final_kernel(const float* __restrict__ part,
                                                       int64_t nblocks, float* __restrict__ out) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < nblocks; i += 256) { a += part[i * 2]; b += part[i * 2 + 1]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[1][0] > 0.0 ? (float)(red[0][0] / red[1][0]) : NAN;   // torch: nan if all ignored
    out[1] = (float)red[1][0];
  }
}

// dlogits = (softmax - onehot) * gscale / n_valid   (0 for ignored rows)
template <typename T>
__global__ void __launch_bounds__(256) ce_bwd_kernel(const T* __restrict__ logits,
                                                     const int64_t* __restrict__ labels, int64_t n,
                                                     int c, int64_t ignore_index,
                                                     const float* __restrict__ fwd_out,
                                                     const float* __restrict__ gscale,
                                                     T* __restrict__ dlogits) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t y = labels[i];
  T* drow = dlogits + i * c;
  if (y == ignore_index || y < 0 || y >= c) {
#pragma unroll
    for (int j = 0; j < CE_MAXC; ++j)
      if (j < c) drow[j] = (T)0.f;
    return;
  }
  const T* row = logits + i * c;
  const float k = gscale[0] / fwd_out[1];
  float x[CE_MAXC], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < CE_MAXC; ++j)
    if (j < c) { x[j] = (float)row[j]; mx = fmaxf(mx, x[j]); }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CE_MAXC; ++j)
    if (j < c) { x[j] = expf(x[j] - mx); s += x[j]; }
#pragma unroll
  for (int j = 0; j < CE_MAXC; ++j)
    if (j < c) drow[j] = (T)((x[j] / s - (j == (int)y ? 1.f : 0.f)) * k);
}

static inline unsigned ew_grid(int64_t n) {
  int64_t g = cdiv(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" int lidal_add_relu_fwd(const void* a, const void* b, void* y, int64_t numel, int dtype,
                                  void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (numel == 0) return 0;
  if (dtype == LIDAL_F32) {
    LIDAL_REQUIRE(numel % 4 == 0, "add_relu: element count must be a multiple of 4");
    add_relu_fwd_kernel<float><<<ew_grid(numel / 4), 256, 0, s>>>((const float*)a, (const float*)b,
                                                                  (float*)y, numel / 4);
  } else if (dtype == LIDAL_BF16) {
    LIDAL_REQUIRE(numel % 8 == 0, "add_relu: element count must be a multiple of 8");
    add_relu_fwd_kernel<__bf16><<<ew_grid(numel / 8), 256, 0, s>>>((const __bf16*)a, (const __bf16*)b,
                                                                   (__bf16*)y, numel / 8);
  } else {
    set_error("add_relu: bad dtype %d", dtype);
    return 2;
  }
  LIDAL_CHECK_LAUNCH("lidal_add_relu_fwd");
  return 0;
}

extern "C" int lidal_add_relu_bwd(const void* y, const void* g, void* gin, int64_t numel, int dtype,
                                  void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (numel == 0) return 0;
  if (dtype == LIDAL_F32) {
    LIDAL_REQUIRE(numel % 4 == 0, "add_relu: element count must be a multiple of 4");
    add_relu_bwd_kernel<float><<<ew_grid(numel / 4), 256, 0, s>>>((const float*)y, (const float*)g,
                                                                  (float*)gin, numel / 4);
  } else if (dtype == LIDAL_BF16) {
    LIDAL_REQUIRE(numel % 8 == 0, "add_relu: element count must be a multiple of 8");
    add_relu_bwd_kernel<__bf16><<<ew_grid(numel / 8), 256, 0, s>>>((const __bf16*)y, (const __bf16*)g,
                                                                   (__bf16*)gin, numel / 8);
  } else {
    set_error("add_relu: bad dtype %d", dtype);
    return 2;
  }
  LIDAL_CHECK_LAUNCH("lidal_add_relu_bwd");
  return 0;
}

extern "C" int64_t lidal_ce_workspace_bytes(int64_t n) { return cdiv(n > 0 ? n : 1, 256) * 8 + 256; }

extern "C" int lidal_ce_fwd(const void* logits, int dtype, const int64_t* labels, int64_t n, int c,
                            int64_t ignore_index, float* out2, void* ws, int64_t ws_bytes,
                            void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(c > 0 && c <= CE_MAXC, "cross_entropy: classes must be in 1..%d", CE_MAXC);
  LIDAL_REQUIRE(ws_bytes >= lidal_ce_workspace_bytes(n), "cross_entropy workspace too small");
  int64_t blocks = cdiv(n > 0 ? n : 1, 256);
  if (dtype == LIDAL_F32)
    ce_fwd_kernel<float><<<(unsigned)blocks, 256, 0, s>>>((const float*)logits, labels, n, c,
                                                          ignore_index, (float*)ws);
  else if (dtype == LIDAL_BF16)
    ce_fwd_kernel<__bf16><<<(unsigned)blocks, 256, 0, s>>>((const __bf16*)logits, labels, n, c,
                                                           ignore_index, (float*)ws);
  else {
    set_error("cross_entropy: bad dtype %d", dtype);
    return 2;
  }
  LIDAL_CHECK_LAUNCH("ce_fwd");
  ce_final_kernel<<<1, 256, 0, s>>>((const float*)ws, blocks, out2);
  LIDAL_CHECK_LAUNCH("ce_final");
  return 0;
}

extern "C" int lidal_ce_bwd(const void* logits, int dtype, const int64_t* labels, int64_t n, int c,
                            int64_t ignore_index, const float* fwd_out2, const float* grad_scale,
                            void* dlogits, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(c > 0 && c <= CE_MAXC, "cross_entropy: classes must be in 1..%d", CE_MAXC);
  if (n == 0) return 0;
  if (dtype == LIDAL_F32)
    ce_bwd_kernel<float><<<(unsigned)cdiv(n, 256), 256, 0, s>>>(
        (const float*)logits, labels, n, c, ignore_index, fwd_out2, grad_scale, (float*)dlogits);
  else if (dtype == LIDAL_BF16)
    ce_bwd_kernel<__bf16><<<(unsigned)cdiv(n, 256), 256, 0, s>>>(
        (const __bf16*)logits, labels, n, c, ignore_index, fwd_out2, grad_scale, (__bf16*)dlogits);
  else {
    set_error("cross_entropy: bad dtype %d", dtype);
    return 2;
  }
  LIDAL_CHECK_LAUNCH("ce_bwd");
  return 0;
}
