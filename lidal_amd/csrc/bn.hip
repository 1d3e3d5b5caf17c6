// BatchNorm over the rows of a [N, C] feature matrix (spnn.BatchNorm = nn.BatchNorm1d on .feats,
// network/utils.py:115, 49 instances per model) for gfx950.
//
// HBM-bound: training forward reads x twice (statistics, normalise) and writes y once; backward
// reads x and dy twice and writes dx once.  All accesses are 16-byte lane loads covering whole
// rows; per-channel reductions are hierarchical (thread -> LDS tree -> per-workgroup partial ->
// one combining workgroup) with Chan's parallel-variance merge on SHIFTED sums, so the variance does
// not cancel when |mean| >> std, and the order is fixed (bitwise reproducible, no atomics).
#include "common.h"
#include <type_traits>

using namespace lidal;

namespace {

constexpr int NT = 256;
#ifndef LIDAL_BN_EW_THREADS
#define LIDAL_BN_EW_THREADS 256
#endif
constexpr int EW_THREADS = LIDAL_BN_EW_THREADS;   // threads of the element-wise kernels that take it as a parameter
constexpr int UNR = 4;      // row loads in flight per thread (per operand)
// per kernel family (scripts/build_variant.py -DLIDAL_BN_UNR_x=8: the sweep of round 5, profiles/README.md): statistics
// pass, normalising pass, backward sums, dx
#ifndef LIDAL_BN_UNR_S
#define LIDAL_BN_UNR_S UNR
#endif
#ifndef LIDAL_BN_UNR_A
#define LIDAL_BN_UNR_A UNR
#endif
#ifndef LIDAL_BN_UNR_P
#define LIDAL_BN_UNR_P UNR
#endif
#ifndef LIDAL_BN_UNR_D
#define LIDAL_BN_UNR_D UNR
#endif
constexpr int UNR_S = LIDAL_BN_UNR_S, UNR_A = LIDAL_BN_UNR_A, UNR_P = LIDAL_BN_UNR_P, UNR_D = LIDAL_BN_UNR_D;
constexpr int MIN_ROWS_PER_WG = 32;    // rows per workgroup (one statistics partial each), at least

template <typename T> struct IO;
template <> struct IO<float> {
  static constexpr int VEC = 4;
  typedef float4 vec;
  __device__ static void unpack(const vec& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  __device__ static vec pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <> struct IO<__bf16> {
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

// VEC consecutive per-channel values (a thread's channel group), or `dflt` for a NULL array.  One uniform branch per array
// and 16-byte loads where the array allows: rounds 1-4 loaded them one element at a time behind a NULL test each -- and
// tested the ReLU / residual flags per element inside the row loops (36 scalar branches per 16-byte piece in the ISA of
// the apply kernel).  The flags are template parameters of the element-wise kernels since round 5.
template <int VEC>
__device__ __forceinline__ void load_channels(const float* __restrict__ p, int off, float dflt, float (&o)[VEC]) {
  if (p == nullptr) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = dflt;
  } else if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + off + i);
      o[i] = v.x; o[i + 1] = v.y; o[i + 2] = v.z; o[i + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = p[off + i];
  }
}

// merge (nb, mb, M2b) into (na, ma, M2a)  -- Chan et al.
__device__ __forceinline__ void chan_merge(double& na, double& ma, double& m2a, double nb, double mb,
                                           double m2b) {
  if (nb == 0.) return;
  double n = na + nb;
  double d = mb - ma;
  ma = ma + d * (nb / n);
  m2a = m2a + m2b + d * d * (na * nb / n);
  na = n;
}

// Thread layout shared by every kernel below: thread t of a 256-thread workgroup owns channel
// group cg = t % CG (VEC consecutive channels = one 16-byte access) and row lane rl = t / CG; the
// workgroup owns rows [blockIdx.x * rpw, ...) and row lane rl walks rows rl, rl+RPI, ...
// (RPI = 256 / CG rows per sweep).  Consecutive threads touch consecutive 16-byte pieces of a row,
// then the next row: fully coalesced, and no per-element div/mod.

// LDS tree sum over the row lanes of each channel group: v[tid*VEC+i] += ... ; result in rl == 0
template <int VEC, typename A>
__device__ __forceinline__ void tree_sum_rows(A* v, int tid, int cg_n, int rpi, int rl) {
  int span = 1;
  while (span < rpi) span <<= 1;
  for (int s = span >> 1; s >= 1; s >>= 1) {
    __syncthreads();
    if (rl < s && rl + s < rpi) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[tid * VEC + i] += v[(tid + s * cg_n) * VEC + i];
    }
  }
  __syncthreads();
}

// ---- statistics: per-workgroup partial (count, mean, M2) per channel ----
// All threads of the workgroup shift by the SAME value (the slab's first row), so inside the slab
// plain sums of (x-K) and (x-K)^2 can be added; only slabs are merged with Chan's formula.
// Every sum of this file is ACCUMULATED IN F64 (per thread, in the LDS tree, across slabs): the
// kernels are HBM-bound, the f64 adds are free, and the reductions then carry no summation error
// at all -- what is left is the f32 rounding of each term.  (With f32 accumulation the gradient of
// one BatchNorm gamma of the 49-layer golden model was off by 1.2e-3 against the f64 reference,
// 4x the error of the f32 CPU oracle: an ill-conditioned sum(dy * xhat), see tests/test_ops_gpu.py.)
template <typename T>
__global__ void __launch_bounds__(NT) bn_stats_partial_kernel(const T* __restrict__ x, int64_t n,
                                                              int c, double* __restrict__ part, int rpw) {
  constexpr int VEC = IO<T>::VEC;
  extern __shared__ double sh[];                // [2][NT][VEC]
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  float shift[VEC];
  double s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { shift[i] = 0.f; s1[i] = 0.; s2[i] = 0.; }
  if (rl < rpi) {
    IO<T>::unpack(*reinterpret_cast<const typename IO<T>::vec*>(x + r_beg * c + cg * VEC), shift);
    int64_t r = r_beg + rl;
    for (; r + (UNR_S - 1) * rpi < r_end; r += UNR_S * rpi) {        // UNR_S independent 16-byte loads in flight
      typename IO<T>::vec v[UNR_S];
#pragma unroll
      for (int u = 0; u < UNR_S; ++u)
        v[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
#pragma unroll
      for (int u = 0; u < UNR_S; ++u) {
        float f[VEC];
        IO<T>::unpack(v[u], f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { float d = f[i] - shift[i]; s1[i] += (double)d; s2[i] += (double)d * (double)d; }
      }
    }
    for (; r < r_end; r += rpi) {
      float f[VEC];
      IO<T>::unpack(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { float d = f[i] - shift[i]; s1[i] += (double)d; s2[i] += (double)d * (double)d; }
    }
  }
  double* a1 = sh; double* a2 = sh + NT * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a1[tid * VEC + i] = s1[i]; a2[tid * VEC + i] = s2[i]; }
  tree_sum_rows<VEC, double>(a1, tid, cg_n, rpi, rl);
  tree_sum_rows<VEC, double>(a2, tid, cg_n, rpi, rl);
  if (rl == 0) {
    const double cnt = (double)(r_end - r_beg);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double t1 = a1[tid * VEC + i], t2 = a2[tid * VEC + i];
      double d = t1 / cnt, m2 = t2 - t1 * d;
      double* dst = part + ((int64_t)blockIdx.x * c + cg * VEC + i) * 3;
      dst[0] = cnt; dst[1] = (double)shift[i] + d; dst[2] = m2 < 0. ? 0. : m2;
    }
  }
}

// One workgroup folds the slab partials of 8 channels: 32 lanes per channel each merge a strided
// subset in order, then a fixed LDS tree merges the 32 lanes.  Writes mean / invstd and updates the
// running statistics exactly as torch.nn.BatchNorm1d (biased var to normalise, unbiased for
// running_var).
template <typename P>
__global__ void __launch_bounds__(NT) bn_stats_final_kernel(const P* __restrict__ part,
                                                            int nparts, int c, float eps,
                                                            float momentum,
                                                            float* __restrict__ mean,
                                                            float* __restrict__ invstd,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var,
                                                            long long* __restrict__ num_batches) {
  __shared__ double sn[NT], sm[NT], sq[NT];
  if (num_batches != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches += 1;
  const int tid = threadIdx.x, cl = tid & 7, pl = tid >> 3;      // 8 channels x 32 lanes
  const int ch = blockIdx.x * 8 + cl;
  double na = 0., ma = 0., qa = 0.;
  if (ch < c)
    for (int p = pl; p < nparts; p += 32) {
      const P* s = part + ((int64_t)p * c + ch) * 3;
      if (na == 0.) { na = (double)s[0]; ma = (double)s[1]; qa = (double)s[2]; }
      else chan_merge(na, ma, qa, (double)s[0], (double)s[1], (double)s[2]);
    }
  sn[tid] = na; sm[tid] = ma; sq[tid] = qa;
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (pl < s) {
      double nb = sn[tid + s * 8], mb = sm[tid + s * 8], qb = sq[tid + s * 8];
      double n0 = sn[tid], m0 = sm[tid], q0 = sq[tid];
      if (n0 == 0.) { n0 = nb; m0 = mb; q0 = qb; }
      else chan_merge(n0, m0, q0, nb, mb, qb);
      sn[tid] = n0; sm[tid] = m0; sq[tid] = q0;
    }
  }
  __syncthreads();
  if (pl == 0 && ch < c) {
    na = sn[tid]; ma = sm[tid]; qa = sq[tid];
    const double var = na > 0. ? qa / na : 0.;
    mean[ch] = (float)ma;
    invstd[ch] = (float)(1. / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const float unbiased = (float)(na > 1. ? qa / (na - 1.) : var);
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)ma;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
  }
}

// y = (x - mean) * invstd * gamma + beta  ==  x * sc + sh   (per-channel sc, sh held in registers)
// `res` (may be null): y = act(...) rounded to T, + res -- the sum of the point branch
// (network/spvcnn.py:104 `z1.F = z1.F + point_transforms(z.F)`) without a pass of its own; relu bit 1 =
// ReLU before that sum, bit 2 = ReLU after it (network/utils.py:171 `relu(net(x) + downsample(x))`)
// (the ReLU / residual flags select one of six specialised copies of the row loops: F bit 0 = ReLU on the normalised
// value, bit 1 = a residual is added, bit 2 = ReLU after the sum)
template <typename T, bool VAR_IN>
__global__ void __launch_bounds__(NT) bn_apply_kernel(const T* __restrict__ x, int64_t n, int c,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ istd_or_var,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      int relu, const T* __restrict__ res,
                                                      T* __restrict__ y, int rpw) {
  constexpr int VEC = IO<T>::VEC;
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  if (rl >= rpi) return;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  float mu[VEC], sc[VEC], sh[VEC], iv[VEC];
  load_channels<VEC>(mean, cg * VEC, 0.f, mu);
  load_channels<VEC>(istd_or_var, cg * VEC, 0.f, iv);
  load_channels<VEC>(gamma, cg * VEC, 1.f, sc);
  load_channels<VEC>(beta, cg * VEC, 0.f, sh);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float is = VAR_IN ? 1.f / sqrtf(iv[i] + eps) : iv[i];
    sc[i] = is * sc[i];
  }
  auto run = [&](auto fc) {
  constexpr int F = decltype(fc)::value;
  constexpr bool RELU1 = (F & 1) != 0, HAS_RES = (F & 2) != 0, RELU2 = (F & 4) != 0;
  auto one = [&](const typename IO<T>::vec& v, const typename IO<T>::vec& vr, int64_t r) {
    float f[VEC], fr[VEC];
    IO<T>::unpack(v, f);
    if (HAS_RES) IO<T>::unpack(vr, fr);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      f[i] = (f[i] - mu[i]) * sc[i] + sh[i];
      if (RELU1) f[i] = fmaxf(f[i], 0.f);
      if (HAS_RES) {
        f[i] = (float)(T)f[i] + fr[i];       // as the stand-alone sum of two T rows
        if (RELU2) f[i] = fmaxf(f[i], 0.f);          // relu(bn(x) + shortcut): the end of a residual block
      }
    }
    *reinterpret_cast<typename IO<T>::vec*>(y + r * c + cg * VEC) = IO<T>::pack(f);
  };
  const T* rsrc = HAS_RES ? res : x;
  int64_t r = r_beg + rl;
  for (; r + (UNR_A - 1) * rpi < r_end; r += UNR_A * rpi) {
    typename IO<T>::vec v[UNR_A], vr[UNR_A];
#pragma unroll
    for (int u = 0; u < UNR_A; ++u)
      v[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
    if (HAS_RES) {
#pragma unroll
      for (int u = 0; u < UNR_A; ++u)
        vr[u] = *reinterpret_cast<const typename IO<T>::vec*>(rsrc + (r + u * rpi) * c + cg * VEC);
    } else {
#pragma unroll
      for (int u = 0; u < UNR_A; ++u) vr[u] = v[u];
    }
#pragma unroll
    for (int u = 0; u < UNR_A; ++u) one(v[u], vr[u], r + u * rpi);
  }
  for (; r < r_end; r += rpi) {
    const typename IO<T>::vec v = *reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC);
    one(v, HAS_RES ? *reinterpret_cast<const typename IO<T>::vec*>(res + r * c + cg * VEC) : v, r);
  }
  };
  switch ((relu & 1) | (res != nullptr ? 2 : 0) | ((res != nullptr && (relu & 2)) ? 4 : 0)) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    case 3: run(std::integral_constant<int, 3>{}); break;
    case 6: run(std::integral_constant<int, 6>{}); break;
    default: run(std::integral_constant<int, 7>{}); break;
  }
}

// backward partials: per workgroup and channel  sum(dy), sum(dy * xhat)
template <typename T>
__global__ void __launch_bounds__(NT) bn_bwd_partial_kernel(const T* __restrict__ x,
                                                            const T* __restrict__ dy, int64_t n,
                                                            int c, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            int relu, double* __restrict__ part, int rpw,
                                                            int64_t ldy) {
  constexpr int VEC = IO<T>::VEC;
  extern __shared__ double sh[];                // [2][NT][VEC]
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  double a[VEC], b[VEC];
  float mu[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.; b[i] = 0.; mu[i] = 0.f; is[i] = 0.f; ga[i] = 1.f; be[i] = 0.f; }
  if (rl < rpi) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mu[i] = mean[cg * VEC + i]; is[i] = invstd[cg * VEC + i];
      if (gamma) ga[i] = gamma[cg * VEC + i];
      if (beta) be[i] = beta[cg * VEC + i];
    }
    auto one = [&](const typename IO<T>::vec& vx, const typename IO<T>::vec& vd) {
      float fx[VEC], fd[VEC];
      IO<T>::unpack(vx, fx);
      IO<T>::unpack(vd, fd);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = (fx[i] - mu[i]) * is[i];
        if (relu && !(xhat * ga[i] + be[i] > 0.f)) fd[i] = 0.f;      // fused ReLU: dy where y > 0
        a[i] += (double)fd[i]; b[i] += (double)fd[i] * (double)xhat;
      }
    };
    int64_t r = r_beg + rl;
#ifdef LIDAL_BN_NO_PIPELINE
    for (; r + (UNR_P - 1) * rpi < r_end; r += UNR_P * rpi) {
      typename IO<T>::vec vx[UNR_P], vd[UNR_P];
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) {
        vx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
        vd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
      }
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) one(vx[u], vd[u]);
    }
#else
    // two batches in flight: the loads of batch i + 1 are requested before batch i is summed -- with ONE wave per SIMD
    // (256 workgroups of 4 waves) nothing else hides the arithmetic of a batch (~90 VALU instructions per 16-byte pair,
    // f64 sums) behind the memory round trip.  Same rows, same order per thread: the sums are those of the plain loop.
    if (r + (UNR_P - 1) * rpi < r_end) {
      typename IO<T>::vec vx[UNR_P], vd[UNR_P], wx[UNR_P], wd[UNR_P];
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) {
        vx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
        vd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
      }
      r += UNR_P * rpi;
      while (r + (UNR_P - 1) * rpi < r_end) {
#pragma unroll
        for (int u = 0; u < UNR_P; ++u) {
          wx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
          wd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
        }
        r += UNR_P * rpi;
#pragma unroll
        for (int u = 0; u < UNR_P; ++u) one(vx[u], vd[u]);
#pragma unroll
        for (int u = 0; u < UNR_P; ++u) { vx[u] = wx[u]; vd[u] = wd[u]; }
      }
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) one(vx[u], vd[u]);
    }
#endif
    for (; r < r_end; r += rpi)
      one(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC),
          *reinterpret_cast<const typename IO<T>::vec*>(dy + r * ldy + cg * VEC));
  }
  double* sa = sh; double* sb = sh + NT * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sa[tid * VEC + i] = a[i]; sb[tid * VEC + i] = b[i]; }
  tree_sum_rows<VEC, double>(sa, tid, cg_n, rpi, rl);
  tree_sum_rows<VEC, double>(sb, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double* dst = part + ((int64_t)blockIdx.x * c + cg * VEC + i) * 2;
      dst[0] = sa[tid * VEC + i]; dst[1] = sb[tid * VEC + i];
    }
  }
}

// bf16 (round 5): the same sums as an ELEMENT-WISE-shaped launch -- rows_per_wg_ew slabs (512 workgroups on the large
// levels: two waves per SIMD), f32 sums per thread (~40 rows) and through the LDS tree, one f32 pair per (channel, slab)
// in the layout of the convolutions' tile sums ([channel][part][2]); bn_bwd_dx_merge_kernel<T, true> merges the slabs
// in f64.  The f64 kernel above runs one wave per SIMD and streams (x, dy) at ~3.7 TB/s; this one at the ~5 TB/s of the
// dx pass (tail_tile_sums_kernel below is the same idea with the ReLU mask of a block's tail in front).  The f32 parity
// mode keeps the f64 sums.
template <typename T, bool RELU>
__global__ void __launch_bounds__(NT) bn_bwd_slab_sums_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n,
                                                              int c, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int relu,
                                                              float* __restrict__ sums, int rpw, int64_t ldy) {
  constexpr int VEC = IO<T>::VEC;
  __shared__ float sh[NT * VEC];
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  float a[VEC], b[VEC];
  float mu[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.f; b[i] = 0.f; mu[i] = 0.f; is[i] = 0.f; ga[i] = 1.f; be[i] = 0.f; }
  if (rl < rpi) {
    load_channels<VEC>(mean, cg * VEC, 0.f, mu);
    load_channels<VEC>(invstd, cg * VEC, 0.f, is);
    load_channels<VEC>(gamma, cg * VEC, 1.f, ga);
    load_channels<VEC>(beta, cg * VEC, 0.f, be);
    auto one = [&](const typename IO<T>::vec& vx, const typename IO<T>::vec& vd) {
      float fx[VEC], fd[VEC];
      IO<T>::unpack(vx, fx);
      IO<T>::unpack(vd, fd);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = (fx[i] - mu[i]) * is[i];
        if (RELU && !(xhat * ga[i] + be[i] > 0.f)) fd[i] = 0.f;
        a[i] += fd[i]; b[i] += fd[i] * xhat;
      }
    };
    int64_t r = r_beg + rl;
    for (; r + (UNR - 1) * rpi < r_end; r += UNR * rpi) {
      typename IO<T>::vec vx[UNR], vd[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        vx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
        vd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(vx[u], vd[u]);
    }
    for (; r < r_end; r += rpi)
      one(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC),
          *reinterpret_cast<const typename IO<T>::vec*>(dy + r * ldy + cg * VEC));
  }
  const int nparts = (int)gridDim.x, part = (int)blockIdx.x;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = a[i];
  tree_sum_rows<VEC, float>(sh, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sums[((int64_t)(cg * VEC + i) * nparts + part) * 2] = sh[tid * VEC + i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = b[i];
  tree_sum_rows<VEC, float>(sh, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sums[((int64_t)(cg * VEC + i) * nparts + part) * 2 + 1] = sh[tid * VEC + i];
  }
}

// The tail of a residual block backwards (network/blocks.py _Residual.backward): gm = g * (out > 0) is the gradient of
// BOTH summands of relu(bn2(x_a) + shortcut).  Written once here and, in the same pass, summed against xhat of bn2 (and,
// DUAL, of the shortcut's BatchNorm over x_b): the loop of bn_bwd_partial_kernel over (x, gm) with relu = 0, term by term
// and in the same order -- the partials are bit for bit those of the separate pass, which no longer reads gm and x back.
template <typename T, bool DUAL>
__global__ void __launch_bounds__(NT) bn_bwd_partial_masked_kernel(
    const T* __restrict__ out, const T* __restrict__ g, T* __restrict__ gm, int64_t n, int c,
    const T* __restrict__ xa, const float* __restrict__ mean_a, const float* __restrict__ invstd_a, double* __restrict__ part_a,
    const T* __restrict__ xb, const float* __restrict__ mean_b, const float* __restrict__ invstd_b, double* __restrict__ part_b,
    int rpw) {
  constexpr int VEC = IO<T>::VEC;
  extern __shared__ double sh[];                // [2][NT][VEC]
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  double a[VEC], b[VEC], a2[VEC], b2[VEC];
  float mu[VEC], is[VEC], mu2[VEC], is2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.; b[i] = 0.; a2[i] = 0.; b2[i] = 0.; mu[i] = 0.f; is[i] = 0.f; mu2[i] = 0.f; is2[i] = 0.f; }
  if (rl < rpi) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mu[i] = mean_a[cg * VEC + i]; is[i] = invstd_a[cg * VEC + i];
      if (DUAL) { mu2[i] = mean_b[cg * VEC + i]; is2[i] = invstd_b[cg * VEC + i]; }
    }
    typedef typename IO<T>::vec V;
    auto one = [&](int64_t r, const V& vo, const V& vg, const V& vx, const V& vx2) {
      float fo[VEC], fd[VEC], fx[VEC], fx2[VEC];
      IO<T>::unpack(vo, fo);
      IO<T>::unpack(vg, fd);
      IO<T>::unpack(vx, fx);
      if (DUAL) IO<T>::unpack(vx2, fx2);
#pragma unroll
      for (int i = 0; i < VEC; ++i) fd[i] = fo[i] > 0.f ? fd[i] : 0.f;
      *reinterpret_cast<V*>(gm + r * c + cg * VEC) = IO<T>::pack(fd);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = (fx[i] - mu[i]) * is[i];
        a[i] += (double)fd[i]; b[i] += (double)fd[i] * (double)xhat;
        if (DUAL) {
          const float xhat2 = (fx2[i] - mu2[i]) * is2[i];
          a2[i] += (double)fd[i]; b2[i] += (double)fd[i] * (double)xhat2;
        }
      }
    };
    int64_t r = r_beg + rl;
    for (; r + (UNR_P - 1) * rpi < r_end; r += UNR_P * rpi) {
      V vo[UNR_P], vg[UNR_P], vx[UNR_P], vx2[UNR_P];
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) {
        const int64_t off = (r + u * rpi) * c + cg * VEC;
        vo[u] = *reinterpret_cast<const V*>(out + off);
        vg[u] = *reinterpret_cast<const V*>(g + off);
        vx[u] = *reinterpret_cast<const V*>(xa + off);
        if (DUAL) vx2[u] = *reinterpret_cast<const V*>(xb + off);
      }
#pragma unroll
      for (int u = 0; u < UNR_P; ++u) one(r + u * rpi, vo[u], vg[u], vx[u], vx2[u]);
    }
    for (; r < r_end; r += rpi) {
      const int64_t off = r * c + cg * VEC;
      V vx2 = *reinterpret_cast<const V*>(xa + off);
      if (DUAL) vx2 = *reinterpret_cast<const V*>(xb + off);
      one(r, *reinterpret_cast<const V*>(out + off), *reinterpret_cast<const V*>(g + off), *reinterpret_cast<const V*>(xa + off), vx2);
    }
  }
  double* sa = sh; double* sb = sh + NT * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sa[tid * VEC + i] = a[i]; sb[tid * VEC + i] = b[i]; }
  tree_sum_rows<VEC, double>(sa, tid, cg_n, rpi, rl);
  tree_sum_rows<VEC, double>(sb, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double* dst = part_a + ((int64_t)blockIdx.x * c + cg * VEC + i) * 2;
      dst[0] = sa[tid * VEC + i]; dst[1] = sb[tid * VEC + i];
    }
  }
  if (DUAL) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sa[tid * VEC + i] = a2[i]; sb[tid * VEC + i] = b2[i]; }
    tree_sum_rows<VEC, double>(sa, tid, cg_n, rpi, rl);
    tree_sum_rows<VEC, double>(sb, tid, cg_n, rpi, rl);
    if (rl == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        double* dst = part_b + ((int64_t)blockIdx.x * c + cg * VEC + i) * 2;
        dst[0] = sa[tid * VEC + i]; dst[1] = sb[tid * VEC + i];
      }
    }
  }
}

// The same tail on the levels with many rows (round 5): the mask as an ELEMENT-WISE launch -- 512 workgroups, two waves
// per SIMD -- that also leaves, per workgroup and channel, the f32 sums of its slab of rows in the layout of the
// convolutions' tile sums ([channel][part][2]); lidal_bn_bwd_tiles merges them (f64 across the parts) in front of its dx
// pass.  bn_bwd_partial_masked_kernel above keeps f64 sums per thread on 256 workgroups and streams at ~2.5 TB/s on
// the 397 k-row level (115 / 156 us against 44 + 41 (+ 41) for the separate passes: profiles/README.md, round 5), which
// is why rounds 3-4 kept the separate passes there.  A thread sums ~40 rows in f32, a slab is ~775 rows: the error of a
// part is that of a convolution's 128-row tile sum.
template <typename T, bool DUAL>
__global__ void __launch_bounds__(NT) tail_tile_sums_kernel(
    const T* __restrict__ out, const T* __restrict__ g, T* __restrict__ gm, int64_t n, int c,
    const T* __restrict__ xa, const float* __restrict__ mean_a, const float* __restrict__ invstd_a, float* __restrict__ sums_a,
    const T* __restrict__ xb, const float* __restrict__ mean_b, const float* __restrict__ invstd_b, float* __restrict__ sums_b,
    int rpw) {
  constexpr int VEC = IO<T>::VEC;
  __shared__ float sh[NT * VEC];
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  float a[VEC], b[VEC], b2[VEC];
  float mu[VEC], is[VEC], mu2[VEC], is2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = 0.f; b[i] = 0.f; b2[i] = 0.f; mu[i] = 0.f; is[i] = 0.f; mu2[i] = 0.f; is2[i] = 0.f; }
  if (rl < rpi) {
    load_channels<VEC>(mean_a, cg * VEC, 0.f, mu);
    load_channels<VEC>(invstd_a, cg * VEC, 0.f, is);
    if (DUAL) {
      load_channels<VEC>(mean_b, cg * VEC, 0.f, mu2);
      load_channels<VEC>(invstd_b, cg * VEC, 0.f, is2);
    }
    typedef typename IO<T>::vec V;
    auto one = [&](int64_t r, const V& vo, const V& vg, const V& vx, const V& vx2) {
      float fo[VEC], fd[VEC], fx[VEC], fx2[VEC];
      IO<T>::unpack(vo, fo);
      IO<T>::unpack(vg, fd);
      IO<T>::unpack(vx, fx);
      if (DUAL) IO<T>::unpack(vx2, fx2);
#pragma unroll
      for (int i = 0; i < VEC; ++i) fd[i] = fo[i] > 0.f ? fd[i] : 0.f;
      *reinterpret_cast<V*>(gm + r * c + cg * VEC) = IO<T>::pack(fd);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        a[i] += fd[i];
        b[i] += fd[i] * ((fx[i] - mu[i]) * is[i]);
        if (DUAL) b2[i] += fd[i] * ((fx2[i] - mu2[i]) * is2[i]);
      }
    };
    int64_t r = r_beg + rl;
    for (; r + (UNR - 1) * rpi < r_end; r += UNR * rpi) {
      V vo[UNR], vg[UNR], vx[UNR], vx2[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t off = (r + u * rpi) * c + cg * VEC;
        vo[u] = *reinterpret_cast<const V*>(out + off);
        vg[u] = *reinterpret_cast<const V*>(g + off);
        vx[u] = *reinterpret_cast<const V*>(xa + off);
        vx2[u] = DUAL ? *reinterpret_cast<const V*>(xb + off) : vx[u];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(r + u * rpi, vo[u], vg[u], vx[u], vx2[u]);
    }
    for (; r < r_end; r += rpi) {
      const int64_t off = r * c + cg * VEC;
      const V vx = *reinterpret_cast<const V*>(xa + off);
      one(r, *reinterpret_cast<const V*>(out + off), *reinterpret_cast<const V*>(g + off), vx,
          DUAL ? *reinterpret_cast<const V*>(xb + off) : vx);
    }
  }
  const int nparts = (int)gridDim.x, part = (int)blockIdx.x;
  // three LDS trees over the row lanes (sum gm, sum gm xhat_a, sum gm xhat_b), each left with rl == 0
#pragma unroll
  for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = a[i];
  tree_sum_rows<VEC, float>(sh, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int64_t o = ((int64_t)(cg * VEC + i) * nparts + part) * 2;
      sums_a[o] = sh[tid * VEC + i];
      if (DUAL) sums_b[o] = sh[tid * VEC + i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = b[i];
  tree_sum_rows<VEC, float>(sh, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sums_a[((int64_t)(cg * VEC + i) * nparts + part) * 2 + 1] = sh[tid * VEC + i];
  }
  if (DUAL) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = b2[i];
    tree_sum_rows<VEC, float>(sh, tid, cg_n, rpi, rl);
    if (rl == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) sums_b[((int64_t)(cg * VEC + i) * nparts + part) * 2 + 1] = sh[tid * VEC + i];
    }
  }
}

__global__ void __launch_bounds__(NT) bn_bwd_final_kernel(const double* __restrict__ part,
                                                          int nparts, int c,
                                                          float* __restrict__ sum_dy,
                                                          float* __restrict__ sum_dy_xhat) {
  __shared__ double sa[NT], sb[NT];
  const int tid = threadIdx.x, cl = tid & 7, pl = tid >> 3;
  const int ch = blockIdx.x * 8 + cl;
  double a = 0., b = 0.;
  if (ch < c)
    for (int p = pl; p < nparts; p += 32) {
      a += part[((int64_t)p * c + ch) * 2];
      b += part[((int64_t)p * c + ch) * 2 + 1];
    }
  sa[tid] = a; sb[tid] = b;
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (pl < s) { sa[tid] += sa[tid + s * 8]; sb[tid] += sb[tid + s * 8]; }
  }
  __syncthreads();
  if (pl == 0 && ch < c) {
    sum_dy[ch] = (float)sa[tid];            // = grad_beta
    sum_dy_xhat[ch] = (float)sb[tid];       // = grad_gamma
  }
}

// dx = gamma * invstd * (dy - sum_dy/n - xhat * sum_dy_xhat/n)
template <typename T>
__global__ void __launch_bounds__(NT) bn_bwd_dx_kernel(const T* __restrict__ x,
                                                       const T* __restrict__ dy, int64_t n, int c,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu,
                                                       const float* __restrict__ sum_dy,
                                                       const float* __restrict__ sum_dy_xhat,
                                                       T* __restrict__ dx, int rpw, int64_t ldy) {
  constexpr int VEC = IO<T>::VEC;
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  if (rl >= rpi) return;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  const float inv_n = 1.f / (float)n;
  float mu[VEC], is[VEC], ga[VEC], be[VEC], k1[VEC], k2[VEC];
  load_channels<VEC>(mean, cg * VEC, 0.f, mu);
  load_channels<VEC>(invstd, cg * VEC, 0.f, is);
  load_channels<VEC>(gamma, cg * VEC, 1.f, ga);
  load_channels<VEC>(beta, cg * VEC, 0.f, be);
  load_channels<VEC>(sum_dy, cg * VEC, 0.f, k1);
  load_channels<VEC>(sum_dy_xhat, cg * VEC, 0.f, k2);
#pragma unroll
  for (int i = 0; i < VEC; ++i) { k1[i] = k1[i] * inv_n; k2[i] = k2[i] * inv_n; }
  auto run = [&](auto rc) {
  constexpr bool RELU = decltype(rc)::value;
  auto one = [&](const typename IO<T>::vec& vx, const typename IO<T>::vec& vd, int64_t r) {
    float fx[VEC], fd[VEC];
    IO<T>::unpack(vx, fx);
    IO<T>::unpack(vd, fd);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float xhat = (fx[i] - mu[i]) * is[i];
      if (RELU && !(xhat * ga[i] + be[i] > 0.f)) fd[i] = 0.f;
      fd[i] = ga[i] * is[i] * (fd[i] - k1[i] - xhat * k2[i]);
    }
    *reinterpret_cast<typename IO<T>::vec*>(dx + r * c + cg * VEC) = IO<T>::pack(fd);
  };
  int64_t r = r_beg + rl;
  for (; r + (UNR_D - 1) * rpi < r_end; r += UNR_D * rpi) {
    typename IO<T>::vec vx[UNR_D], vd[UNR_D];
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) {
      vx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
      vd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
    }
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) one(vx[u], vd[u], r + u * rpi);
  }
  for (; r < r_end; r += rpi)
    one(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC),
        *reinterpret_cast<const typename IO<T>::vec*>(dy + r * ldy + cg * VEC), r);
  };
  if (relu) run(std::true_type{}); else run(std::false_type{});
}

// Slab size = rows / 256: one workgroup per CU on every level.  Each workgroup pays a fixed set-up
// (per-channel parameters, and an LDS tree in the reducing kernels), so fewer and larger slabs win
// as long as every CU has one; and an equal share per CU matters: with 512-row slabs the 396k-row
// level ran as 775 workgroups, 3.03 per CU (measured sweep: profiles/README.md).
constexpr int BN_WGS = 256;
static inline int slab_rows(int64_t n) {
  int64_t rpw = (n + BN_WGS - 1) / BN_WGS;
  if (rpw < MIN_ROWS_PER_WG) rpw = MIN_ROWS_PER_WG;
  return (int)rpw;
}
static inline int rows_per_wg(int64_t n) { return slab_rows(n); }
// The element-wise kernels (apply, dx) have no tree and a light set-up, and at one workgroup per CU the
// large levels are short of bytes in flight: 512 workgroups took dx at 396662 x 96 bf16 from 61.7 to
// 44.3 us (3.7 -> 5.2 TB/s), while the small levels lost (43000 x 128: 14.4 -> 18.3 us) and 1024 lost
// everywhere.  The reducing kernels are best at 256 on every level (sweep: profiles/README.md).
constexpr int64_t EW_WIDE_BYTES = 12ll << 20;
static inline int rows_per_wg_ew(int64_t n, int64_t row_bytes) {
#ifndef LIDAL_BN_EW_WGS
#define LIDAL_BN_EW_WGS (2 * BN_WGS)
#endif
  const int wgs = n * row_bytes >= EW_WIDE_BYTES ? LIDAL_BN_EW_WGS : BN_WGS;
  int64_t rpw = (n + wgs - 1) / wgs;
  if (rpw < MIN_ROWS_PER_WG) rpw = MIN_ROWS_PER_WG;
  return (int)rpw;
}
// the forward (apply) kernels alone: A/B knob for their grid (LIDAL_BN_APPLY_WGS, 0 = as the other element-wise kernels)
#ifndef LIDAL_BN_APPLY_WGS
#define LIDAL_BN_APPLY_WGS 0
#endif
static inline int rows_per_wg_apply(int64_t n, int64_t row_bytes) {
  if (LIDAL_BN_APPLY_WGS == 0 || n * row_bytes < EW_WIDE_BYTES / 4) return rows_per_wg_ew(n, row_bytes);
  int64_t rpw = (n + LIDAL_BN_APPLY_WGS - 1) / LIDAL_BN_APPLY_WGS;
  if (rpw < MIN_ROWS_PER_WG) rpw = MIN_ROWS_PER_WG;
  return (int)rpw;
}
static inline int nslabs_apply(int64_t n, int64_t row_bytes) {
  return (int)cdiv(n > 0 ? n : 1, rows_per_wg_apply(n, row_bytes));
}
static inline int nslabs_ew(int64_t n, int64_t row_bytes) {
  return (int)cdiv(n > 0 ? n : 1, rows_per_wg_ew(n, row_bytes));
}
static inline int nparts_for(int64_t n) { return (int)cdiv(n > 0 ? n : 1, rows_per_wg(n)); }
template <typename T>
int bn_train_fwd(const void* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
                 float momentum, float* rm, float* rv, long long* nbt, int relu, const void* res,
                 void* y, float* mean, float* invstd, double* part, hipStream_t s) {
  constexpr int VEC = IO<T>::VEC;
  int np = nparts_for(n);
  bn_stats_partial_kernel<T><<<np, NT, 2 * NT * VEC * sizeof(double), s>>>((const T*)x, n, c, part,
                                                                           rows_per_wg(n));
  LIDAL_CHECK_LAUNCH("bn_stats_partial");
  bn_stats_final_kernel<double><<<(unsigned)cdiv(c, 8), NT, 0, s>>>(part, np, c, eps, momentum, mean,
                                                             invstd, rm, rv, nbt);
  LIDAL_CHECK_LAUNCH("bn_stats_final");
  bn_apply_kernel<T, false><<<nslabs_apply(n, (int64_t)c * sizeof(T)), NT, 0, s>>>((const T*)x, n, c, mean, invstd, gamma,
                                                        beta, eps, relu, (const T*)res, (T*)y,
                                                        rows_per_wg_apply(n, (int64_t)c * sizeof(T)));
  LIDAL_CHECK_LAUNCH("bn_apply");
  return 0;
}

// (defined with the merge-in-consumer kernels further down)
template <typename T, bool TILES>
bool bn_bwd_merge_dx(const void* x, const void* dy, int64_t ldy, int64_t n, int c, const float* gamma, const float* beta,
                     int relu, const float* mean, const float* invstd, void* dx, float* ggamma, float* gbeta,
                     const void* part, int nparts, hipStream_t s);

// LIDAL_BN_SLAB_SUMS=0 / lidal_bn_set_slab_sums(0): the f64 partial sums for bf16 too (A/B, and the bitwise tests against
// the fused tail of the f32 mode)
static int g_slab_sums = -1;            // -1: not read yet
static inline bool slab_sums_enabled() {
  if (g_slab_sums < 0) { const char* e = getenv("LIDAL_BN_SLAB_SUMS"); g_slab_sums = (e && e[0] == '0') ? 0 : 1; }
  return g_slab_sums != 0;
}
__global__ void __launch_bounds__(NT) bn_bwd_tiles_final_kernel(const float* __restrict__ part, int nparts, int c,
                                                                float* __restrict__ sum_dy,
                                                                float* __restrict__ sum_dy_xhat) {
  __shared__ double sa[NT], sb[NT];
  const int tid = threadIdx.x, ch = blockIdx.x;
  double a = 0., b = 0.;
  for (int p = tid; p < nparts; p += NT) {
    const float* s = part + ((int64_t)ch * nparts + p) * 2;
    a += (double)s[0]; b += (double)s[1];
  }
  sa[tid] = a; sb[tid] = b;
  for (int st = NT / 2; st >= 1; st >>= 1) {
    __syncthreads();
    if (tid < st) { sa[tid] += sa[tid + st]; sb[tid] += sb[tid + st]; }
  }
  if (tid == 0) {
    sum_dy[ch] = (float)sa[0];            // = grad_beta
    sum_dy_xhat[ch] = (float)sb[0];       // = grad_gamma
  }
}

template <typename T>
int bn_bwd(const void* x, const void* dy, int64_t ldy, int64_t n, int c, const float* gamma,
           const float* beta, int relu, const float* mean, const float* invstd, void* dx, float* ggamma,
           float* gbeta, double* part, hipStream_t s) {
  constexpr int VEC = IO<T>::VEC;
  if (sizeof(T) == 2 && slab_sums_enabled()) {
    // bf16: f32 slab sums on the element-wise grid, merged like the convolutions' tile sums (bn_bwd_slab_sums_kernel)
    const int64_t rb = (int64_t)c * sizeof(T);
    const int parts = nslabs_ew(n, rb);
    float* sums = (float*)part;               // c * parts * 8 bytes <= lidal_bn_workspace_bytes (parts <= 2 * nparts_for)
    if (relu)
      bn_bwd_slab_sums_kernel<T, true><<<parts, NT, 0, s>>>((const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu,
                                                           sums, rows_per_wg_ew(n, rb), ldy);
    else
      bn_bwd_slab_sums_kernel<T, false><<<parts, NT, 0, s>>>((const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu,
                                                            sums, rows_per_wg_ew(n, rb), ldy);
    LIDAL_CHECK_LAUNCH("bn_bwd_slab_sums");
    if (bn_bwd_merge_dx<T, true>(x, dy, ldy, n, c, gamma, beta, relu, mean, invstd, dx, ggamma, gbeta, sums, parts, s)) {
      LIDAL_CHECK_LAUNCH("bn_bwd_dx(slab sums merged in the launch)");
      return 0;
    }
    bn_bwd_tiles_final_kernel<<<(unsigned)c, NT, 0, s>>>(sums, parts, c, gbeta, ggamma);
    LIDAL_CHECK_LAUNCH("bn_bwd_final(slabs)");
    if (dx != nullptr) {
      bn_bwd_dx_kernel<T><<<parts, NT, 0, s>>>((const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu, gbeta,
                                               ggamma, (T*)dx, rows_per_wg_ew(n, rb), ldy);
      LIDAL_CHECK_LAUNCH("bn_bwd_dx");
    }
    return 0;
  }
  int np = nparts_for(n);
  bn_bwd_partial_kernel<T><<<np, NT, 2 * NT * VEC * sizeof(double), s>>>(
      (const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu, part, rows_per_wg(n), ldy);
  LIDAL_CHECK_LAUNCH("bn_bwd_partial");
  if (bn_bwd_merge_dx<T, false>(x, dy, ldy, n, c, gamma, beta, relu, mean, invstd, dx, ggamma, gbeta, part, np, s)) {
    LIDAL_CHECK_LAUNCH("bn_bwd_dx(sums merged in the launch)");
    return 0;
  }
  bn_bwd_final_kernel<<<(unsigned)cdiv(c, 8), NT, 0, s>>>(part, np, c, gbeta, ggamma);
  LIDAL_CHECK_LAUNCH("bn_bwd_final");
  if (dx != nullptr) {
    bn_bwd_dx_kernel<T><<<nslabs_ew(n, (int64_t)c * sizeof(T)), NT, 0, s>>>((const T*)x, (const T*)dy, n, c, mean, invstd,
                                                    gamma, beta, relu, gbeta, ggamma, (T*)dx,
                                                    rows_per_wg_ew(n, (int64_t)c * sizeof(T)), ldy);
    LIDAL_CHECK_LAUNCH("bn_bwd_dx");
  }
  return 0;
}

template <typename T>
int bn_bwd_masked_partials(const void* out, const void* g, void* gm, int64_t n, int c, const void* xa, const float* mean_a,
                           const float* invstd_a, double* part_a, const void* xb, const float* mean_b,
                           const float* invstd_b, double* part_b, hipStream_t s) {
  constexpr int VEC = IO<T>::VEC;
  const int np = nparts_for(n);
  if (xb != nullptr)
    bn_bwd_partial_masked_kernel<T, true><<<np, NT, 2 * NT * VEC * sizeof(double), s>>>(
        (const T*)out, (const T*)g, (T*)gm, n, c, (const T*)xa, mean_a, invstd_a, part_a, (const T*)xb, mean_b, invstd_b,
        part_b, rows_per_wg(n));
  else
    bn_bwd_partial_masked_kernel<T, false><<<np, NT, 2 * NT * VEC * sizeof(double), s>>>(
        (const T*)out, (const T*)g, (T*)gm, n, c, (const T*)xa, mean_a, invstd_a, part_a, nullptr, nullptr, nullptr,
        nullptr, rows_per_wg(n));
  LIDAL_CHECK_LAUNCH("bn_bwd_partial_masked");
  return 0;
}

// bn_bwd without its first pass: the partials are in `part` already
template <typename T>
int bn_bwd_from_partials(const void* x, const void* dy, int64_t ldy, int64_t n, int c, const float* gamma,
                         const float* beta, int relu, const float* mean, const float* invstd, void* dx, float* ggamma,
                         float* gbeta, const double* part, hipStream_t s) {
  if (bn_bwd_merge_dx<T, false>(x, dy, ldy, n, c, gamma, beta, relu, mean, invstd, dx, ggamma, gbeta, part, nparts_for(n), s)) {
    LIDAL_CHECK_LAUNCH("bn_bwd_dx(sums merged in the launch)");
    return 0;
  }
  bn_bwd_final_kernel<<<(unsigned)cdiv(c, 8), NT, 0, s>>>(part, nparts_for(n), c, gbeta, ggamma);
  LIDAL_CHECK_LAUNCH("bn_bwd_final");
  if (dx != nullptr) {
    bn_bwd_dx_kernel<T><<<nslabs_ew(n, (int64_t)c * sizeof(T)), NT, 0, s>>>((const T*)x, (const T*)dy, n, c, mean, invstd,
                                                    gamma, beta, relu, gbeta, ggamma, (T*)dx,
                                                    rows_per_wg_ew(n, (int64_t)c * sizeof(T)), ldy);
    LIDAL_CHECK_LAUNCH("bn_bwd_dx");
  }
  return 0;
}

// eval-mode BatchNorm as a per-channel affine map: y = x * scale + shift
__global__ void __launch_bounds__(256) bn_fold_kernel(const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ var, float eps,
                                                      int c, float* __restrict__ scale,
                                                      float* __restrict__ shift) {
  int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  float sc = (1.f / sqrtf(var[ch] + eps)) * (gamma ? gamma[ch] : 1.f);
  scale[ch] = sc;
  shift[ch] = (beta ? beta[ch] : 0.f) - mean[ch] * sc;
}

}  // namespace

extern "C" int lidal_bn_check_device(void);
static int bn_check(int64_t n, int c, int dtype) {
  if (int rc = lidal_bn_check_device()) return rc;       // a fused launch of this device timed out earlier: not a silent NaN
  int vec = dtype == LIDAL_BF16 ? 8 : 4;
  LIDAL_REQUIRE(dtype == LIDAL_F32 || dtype == LIDAL_BF16, "bn: bad dtype %d", dtype);
  LIDAL_REQUIRE(c > 0 && c % vec == 0 && c / vec <= NT, "bn: channels %d must be a multiple of %d and <= %d",
                c, vec, NT * vec);
  LIDAL_REQUIRE(n >= 0, "bn: bad row count");
  return 0;
}

extern "C" int64_t lidal_bn_workspace_bytes(int64_t n, int c) {
  return (int64_t)nparts_for(n) * c * 3 * sizeof(double) + 256;
}

extern "C" int lidal_bn_train_fwd(const void* x, int dtype, int64_t n, int c, const float* gamma,
                                  const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, int relu, const void* residual,
                                  void* y, float* save_mean, float* save_invstd, void* ws,
                                  int64_t ws_bytes, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "bn_train_fwd: needs at least one row");
  LIDAL_REQUIRE(ws_bytes >= lidal_bn_workspace_bytes(n, c), "bn workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32)
    return bn_train_fwd<float>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var,
                               (long long*)num_batches_tracked, relu, residual, y, save_mean,
                               save_invstd, (double*)ws, s);
  return bn_train_fwd<__bf16>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var,
                              (long long*)num_batches_tracked, relu, residual, y, save_mean,
                              save_invstd, (double*)ws, s);
}

// Merge of the per-tile (count, mean, M2) f32 triples a convolution left (thousands of tiles): one
// workgroup per channel, plain f64 sums N, S1 = sum n m, S2 = sum (M2 + n m^2) -- in f64 the
// cancellation of S2/N - mean^2 is harmless, and no per-element division (Chan's formula) is needed.
__global__ void __launch_bounds__(NT) bn_tiles_final_kernel(const float* __restrict__ part, int nparts,
                                                            int c, float eps, float momentum,
                                                            float* __restrict__ mean,
                                                            float* __restrict__ invstd,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var,
                                                            long long* __restrict__ num_batches) {
  __shared__ double sn[NT], s1[NT], s2[NT];
  const int tid = threadIdx.x, ch = blockIdx.x;
  if (num_batches != nullptr && ch == 0 && tid == 0) *num_batches += 1;
  double n = 0., a = 0., b = 0.;
  for (int p = tid; p < nparts; p += NT) {
    const float* s = part + ((int64_t)ch * nparts + p) * 3;         // [c][tiles][3]: a channel's triples are one run
    const double pn = (double)s[0], pm = (double)s[1];
    n += pn; a += pn * pm; b += (double)s[2] + pn * pm * pm;
  }
  sn[tid] = n; s1[tid] = a; s2[tid] = b;
  for (int st = NT / 2; st >= 1; st >>= 1) {
    __syncthreads();
    if (tid < st) { sn[tid] += sn[tid + st]; s1[tid] += s1[tid + st]; s2[tid] += s2[tid + st]; }
  }
  if (tid == 0) {
    n = sn[0];
    const double m = n > 0. ? s1[0] / n : 0.;
    double m2 = s2[0] - n * m * m;
    if (m2 < 0.) m2 = 0.;
    const double var = n > 0. ? m2 / n : 0.;
    mean[ch] = (float)m;
    invstd[ch] = (float)(1. / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const float unbiased = (float)(n > 1. ? m2 / (n - 1.) : var);
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
  }
}

// ---- merge kernels inside their consumers ------------------------------------------------------------------
// A BatchNorm layer was three launches forwards (tile merge, apply) and backwards (partial / tile merge, dx); the merges
// are 5-7 us launches that compute for 2 us (104 per step: on one scan every eighth microsecond of the step).  Here
// the consumer kernel's first workgroups do the merge -- workgroup b the channels b, b + grid, ... with the arithmetic
// of the stand-alone merge kernel, term by term -- and PUBLISH the two f32 values of a channel as two 64-bit words
// (value | launch token << 32; agent-scope atomics, no fence needed: the token travels with the value); every
// workgroup then fetches the words of all channels (spinning until the token matches) and streams its rows.  Forward
// progress: the merging workgroups have the lowest ids of the launch and are dispatched first (the assumption the
// radix sort's look-back makes, sort.hip).  The slots of a launch are one of TWO buffers that belong to the launch's
// STREAM (a fixed pool of 64 buffers per device, 2 MiB, allocated by this library at the first fused launch: 32
// streams per device; a 33rd stream, or a pool that cannot be allocated, takes the separate merge launches): launches
// of one stream execute in order, so a buffer is never rewritten while an earlier launch still reads it, whatever
// other streams or host threads do; the token is a process-wide counter, so a stale word never matches.  A fetch
// gives up after ~30 s (a broken dispatch-order assumption must fail a test, not hang a GPU): it poisons its channel
// with NaN AND raises the pool's error word (pinned host memory), which the next BatchNorm entry point or
// lidal_plan_run on that device turns into an error return -- never a silent NaN.
struct Slots { unsigned long long* v; unsigned token; unsigned* error; };
constexpr int SLOT_RING = 64, SLOT_CH = 2048, SLOTS_PER_STREAM = 2;
constexpr unsigned SPIN_LIMIT = 1u << 25;     // ~1 us per probe: half a minute (a time-sliced device may stall a launch for seconds)

__device__ __forceinline__ void publish(const Slots& s, int ch, float a, float b) {
  const unsigned long long t = (unsigned long long)s.token << 32;
  __hip_atomic_store(s.v + 2 * ch, t | (unsigned long long)__float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(s.v + 2 * ch + 1, t | (unsigned long long)__float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float fetch_one(const Slots& s, int idx) {
  unsigned long long v = __hip_atomic_load(s.v + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((unsigned)(v >> 32) != s.token) {
    if (++spins > SPIN_LIMIT) {
      __hip_atomic_store(s.error, s.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return __uint_as_float(0x7fc00000u);
    }
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(s.v + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)v);
}

// bn_tiles_final_kernel + bn_apply_kernel<T, false> in one launch.  NTT threads (EW_THREADS: sixteen waves -- the streaming
// part wants waves in flight, not workgroups: every workgroup pays the wait for the merged values once); the merge runs on
// the first NT of them, the others add exact zeros to its tree: the sums of bn_tiles_final_kernel bit for bit.
// F: bit 0 = ReLU on the normalised value, bit 1 = a residual is added, bit 2 = ReLU after the sum
template <typename T, int NTT, int F>
__global__ void __launch_bounds__(NTT) bn_apply_tiles_kernel(const T* __restrict__ x, int64_t n, int c,
                                                            const float* __restrict__ part, int nparts, float eps,
                                                            float momentum, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var,
                                                            long long* __restrict__ num_batches,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const T* __restrict__ res, T* __restrict__ y, int rpw,
                                                            Slots slots) {
  constexpr bool RELU1 = (F & 1) != 0, HAS_RES = (F & 2) != 0, RELU2 = (F & 4) != 0;
  constexpr int VEC = IO<T>::VEC;
  __shared__ double sn[NTT], s1[NTT], s2[NTT];
  __shared__ float smu[SLOT_CH], sis[SLOT_CH];
  const int tid = threadIdx.x;
  if (num_batches != nullptr && blockIdx.x == 0 && tid == 0) *num_batches += 1;
  // ---- the merge of bn_tiles_final_kernel, for this workgroup's channels
  for (int ch = blockIdx.x; ch < c; ch += gridDim.x) {
    double nn = 0., a = 0., b = 0.;
    // (eight tiles' triples requested at once, summed in tile order: the sums of the one-at-a-time loop, with the
    // loads of a batch in flight together -- the merge is on the layer's critical path)
    for (int p0 = tid; tid < NT && p0 < nparts; p0 += 8 * NT) {
      float t0[8], t1[8], t2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * NT;
        const float* sp = part + ((int64_t)ch * nparts + (p < nparts ? p : p0)) * 3;
        t0[u] = sp[0]; t1[u] = sp[1]; t2[u] = sp[2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + u * NT < nparts) {
          const double pn = (double)t0[u], pm = (double)t1[u];
          nn += pn; a += pn * pm; b += (double)t2[u] + pn * pm * pm;
        }
    }
    sn[tid] = nn; s1[tid] = a; s2[tid] = b;
    for (int st = NTT / 2; st >= 1; st >>= 1) {
      __syncthreads();
      if (tid < st) { sn[tid] += sn[tid + st]; s1[tid] += s1[tid + st]; s2[tid] += s2[tid + st]; }
    }
    if (tid == 0) {
      nn = sn[0];
      const double m = nn > 0. ? s1[0] / nn : 0.;
      double m2 = s2[0] - nn * m * m;
      if (m2 < 0.) m2 = 0.;
      const double var = nn > 0. ? m2 / nn : 0.;
      const float fm = (float)m, fi = (float)(1. / sqrt(var + (double)eps));
      mean[ch] = fm;
      invstd[ch] = fi;
      if (running_mean != nullptr) {
        const float unbiased = (float)(nn > 1. ? m2 / (nn - 1.) : var);
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * fm;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
      }
      publish(slots, ch, fm, fi);
    }
    __syncthreads();
  }
  // ---- the first rows of this workgroup's slab are requested BEFORE the statistics are waited for: the merge's
  //      latency (a strided read of the tiles, an 8-step LDS tree) hides behind them
  const int cg_n = c / VEC, rpi = NTT / cg_n;
  const int cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  const T* rsrc = HAS_RES ? res : x;
  int64_t r = r_beg + rl;
  const bool first = rl < rpi && r + (UNR_A - 1) * rpi < r_end;
  typename IO<T>::vec v0[UNR_A], vr0[UNR_A];
  if (first) {
#pragma unroll
    for (int u = 0; u < UNR_A; ++u)
      v0[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
    if (HAS_RES) {
#pragma unroll
      for (int u = 0; u < UNR_A; ++u)
        vr0[u] = *reinterpret_cast<const typename IO<T>::vec*>(rsrc + (r + u * rpi) * c + cg * VEC);
    }
  }
  // ---- every channel's (mean, invstd), from whichever workgroup merged it
  for (int ch = tid; ch < c; ch += NTT) {
    smu[ch] = fetch_one(slots, 2 * ch);
    sis[ch] = fetch_one(slots, 2 * ch + 1);
  }
  __syncthreads();
  // ---- bn_apply_kernel<T, false>
  if (rl >= rpi) return;
  float mu[VEC], sc[VEC], sh[VEC];
  load_channels<VEC>(gamma, cg * VEC, 1.f, sc);
  load_channels<VEC>(beta, cg * VEC, 0.f, sh);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = cg * VEC + i;
    mu[i] = smu[ch];
    sc[i] = sis[ch] * sc[i];
  }
  auto one = [&](const typename IO<T>::vec& v, const typename IO<T>::vec& vr, int64_t r) {
    float f[VEC], fr[VEC];
    IO<T>::unpack(v, f);
    if (HAS_RES) IO<T>::unpack(vr, fr);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      f[i] = (f[i] - mu[i]) * sc[i] + sh[i];
      if (RELU1) f[i] = fmaxf(f[i], 0.f);
      if (HAS_RES) {
        f[i] = (float)(T)f[i] + fr[i];
        if (RELU2) f[i] = fmaxf(f[i], 0.f);
      }
    }
    *reinterpret_cast<typename IO<T>::vec*>(y + r * c + cg * VEC) = IO<T>::pack(f);
  };
  if (first) {
#pragma unroll
    for (int u = 0; u < UNR_A; ++u) one(v0[u], HAS_RES ? vr0[u] : v0[u], r + u * rpi);
    r += UNR_A * rpi;
  }
  for (; r + (UNR_A - 1) * rpi < r_end; r += UNR_A * rpi) {
    typename IO<T>::vec v[UNR_A], vr[UNR_A];
#pragma unroll
    for (int u = 0; u < UNR_A; ++u)
      v[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
    if (HAS_RES) {
#pragma unroll
      for (int u = 0; u < UNR_A; ++u)
        vr[u] = *reinterpret_cast<const typename IO<T>::vec*>(rsrc + (r + u * rpi) * c + cg * VEC);
    } else {
#pragma unroll
      for (int u = 0; u < UNR_A; ++u) vr[u] = v[u];
    }
#pragma unroll
    for (int u = 0; u < UNR_A; ++u) one(v[u], vr[u], r + u * rpi);
  }
  for (; r < r_end; r += rpi) {
    const typename IO<T>::vec v = *reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC);
    one(v, HAS_RES ? *reinterpret_cast<const typename IO<T>::vec*>(res + r * c + cg * VEC) : v, r);
  }
}

// bn_bwd_final_kernel (TILES = false: f64 partial pairs of bn_bwd_partial_kernel, 8 channels x 32 lanes per unit) or
// bn_bwd_tiles_final_kernel (TILES = true: f32 pairs per 128-row tile, one channel per unit) + bn_bwd_dx_kernel
template <typename T, bool TILES, bool RELU>
__global__ void __launch_bounds__(NT) bn_bwd_dx_merge_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n,
                                                             int c, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int relu, const void* __restrict__ part, int nparts,
                                                             float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat,
                                                             T* __restrict__ dx, int rpw, int64_t ldy, Slots slots) {
  constexpr int VEC = IO<T>::VEC;
  __shared__ double sa[NT], sb[NT];
  __shared__ float sk1[SLOT_CH], sk2[SLOT_CH];
  const int tid = threadIdx.x;
  if (TILES) {
    const float* pt = (const float*)part;
    for (int ch = blockIdx.x; ch < c; ch += gridDim.x) {
      double a = 0., b = 0.;
      for (int p0 = tid; p0 < nparts; p0 += 8 * NT) {       // (batched loads, sums in tile order: bn_apply_tiles_kernel)
        float t0[8], t1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int p = p0 + u * NT;
          const float* sp = pt + ((int64_t)ch * nparts + (p < nparts ? p : p0)) * 2;
          t0[u] = sp[0]; t1[u] = sp[1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (p0 + u * NT < nparts) { a += (double)t0[u]; b += (double)t1[u]; }
      }
      sa[tid] = a; sb[tid] = b;
      for (int st = NT / 2; st >= 1; st >>= 1) {
        __syncthreads();
        if (tid < st) { sa[tid] += sa[tid + st]; sb[tid] += sb[tid + st]; }
      }
      if (tid == 0) {
        const float fa = (float)sa[0], fb = (float)sb[0];
        sum_dy[ch] = fa; sum_dy_xhat[ch] = fb;
        publish(slots, ch, fa, fb);
      }
      __syncthreads();
    }
  } else {
    const double* pd = (const double*)part;
    const int cl = tid & 7, pl = tid >> 3;
    for (int u = blockIdx.x; u * 8 < c; u += gridDim.x) {
      const int ch = u * 8 + cl;
      double a = 0., b = 0.;
      if (ch < c)
        for (int p = pl; p < nparts; p += 32) {
          a += pd[((int64_t)p * c + ch) * 2];
          b += pd[((int64_t)p * c + ch) * 2 + 1];
        }
      sa[tid] = a; sb[tid] = b;
      for (int st = 16; st >= 1; st >>= 1) {
        __syncthreads();
        if (pl < st) { sa[tid] += sa[tid + st * 8]; sb[tid] += sb[tid + st * 8]; }
      }
      __syncthreads();
      if (pl == 0 && ch < c) {
        const float fa = (float)sa[tid], fb = (float)sb[tid];
        sum_dy[ch] = fa; sum_dy_xhat[ch] = fb;
        publish(slots, ch, fa, fb);
      }
      __syncthreads();
    }
  }
  if (dx == nullptr) return;            // (only the parameter gradients were wanted)
  // ---- the first rows are requested before the sums are waited for (as in bn_apply_tiles_kernel)
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  int64_t r = r_beg + rl;
  const bool first = rl < rpi && r + (UNR_D - 1) * rpi < r_end;
  typename IO<T>::vec vx0[UNR_D], vd0[UNR_D];
  if (first) {
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) {
      vx0[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
      vd0[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
    }
  }
  for (int ch = tid; ch < c; ch += NT) {
    sk1[ch] = fetch_one(slots, 2 * ch);
    sk2[ch] = fetch_one(slots, 2 * ch + 1);
  }
  __syncthreads();
  // ---- bn_bwd_dx_kernel
  if (rl >= rpi) return;
  const float inv_n = 1.f / (float)n;
  float mu[VEC], is[VEC], ga[VEC], be[VEC], k1[VEC], k2[VEC];
  load_channels<VEC>(mean, cg * VEC, 0.f, mu);
  load_channels<VEC>(invstd, cg * VEC, 0.f, is);
  load_channels<VEC>(gamma, cg * VEC, 1.f, ga);
  load_channels<VEC>(beta, cg * VEC, 0.f, be);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = cg * VEC + i;
    k1[i] = sk1[ch] * inv_n; k2[i] = sk2[ch] * inv_n;
  }
  auto one = [&](const typename IO<T>::vec& vx, const typename IO<T>::vec& vd, int64_t r) {
    float fx[VEC], fd[VEC];
    IO<T>::unpack(vx, fx);
    IO<T>::unpack(vd, fd);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float xhat = (fx[i] - mu[i]) * is[i];
      if (RELU && !(xhat * ga[i] + be[i] > 0.f)) fd[i] = 0.f;
      fd[i] = ga[i] * is[i] * (fd[i] - k1[i] - xhat * k2[i]);
    }
    *reinterpret_cast<typename IO<T>::vec*>(dx + r * c + cg * VEC) = IO<T>::pack(fd);
  };
  if (first) {
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) one(vx0[u], vd0[u], r + u * rpi);
    r += UNR_D * rpi;
  }
  for (; r + (UNR_D - 1) * rpi < r_end; r += UNR_D * rpi) {
    typename IO<T>::vec vx[UNR_D], vd[UNR_D];
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) {
      vx[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
      vd[u] = *reinterpret_cast<const typename IO<T>::vec*>(dy + (r + u * rpi) * ldy + cg * VEC);
    }
#pragma unroll
    for (int u = 0; u < UNR_D; ++u) one(vx[u], vd[u], r + u * rpi);
  }
  for (; r < r_end; r += rpi)
    one(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC),
        *reinterpret_cast<const typename IO<T>::vec*>(dy + r * ldy + cg * VEC), r);
}

// ---- host side of the slots: a pool of buffers per device, two per stream, one token per launch
#include <atomic>
#include <mutex>
#include <unordered_map>
namespace {
std::atomic<unsigned> g_token{1};
std::atomic<int> g_fused{-1};           // -1: not read yet (LIDAL_BN_FUSED, default on)
struct SlotPool {
  unsigned long long* base = nullptr;   // SLOT_RING buffers of 2 * SLOT_CH words
  unsigned* error_host = nullptr;       // pinned, read by the host without a synchronisation
  unsigned* error_dev = nullptr;        // the same word as the device addresses it
  bool failed = false;                  // the pool could not be allocated: separate launches from now on
  std::unordered_map<hipStream_t, unsigned> streams;   // stream -> (index of its first buffer) << 1 | next buffer
};
SlotPool g_pool[MAX_DEVICES];
std::mutex g_pool_mutex;

bool bn_fused() {
  int f = g_fused.load();
  if (f < 0) {
    const char* e = getenv("LIDAL_BN_FUSED");
    f = (e == nullptr || e[0] != '0') ? 1 : 0;
    g_fused.store(f);
  }
  return f != 0;
}

// the slots of the next launch on stream `s` of the current device (false: no pool, or more than SLOT_RING /
// SLOTS_PER_STREAM streams on this device -- the caller launches merge and consumer apart)
bool next_slots(Slots* out, hipStream_t s) {
  const int d = current_device();
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  SlotPool& pool = g_pool[d];
  if (pool.failed) return false;
  if (pool.base == nullptr) {
    void *p = nullptr, *h = nullptr, *hd = nullptr;
    const size_t bytes = (size_t)SLOT_RING * 2 * SLOT_CH * sizeof(unsigned long long);
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess ||
        hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
        hipHostGetDevicePointer(&hd, h, 0) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      pool.failed = true;
      return false;
    }
    *(volatile unsigned*)h = 0;
    pool.base = (unsigned long long*)p;
    pool.error_host = (unsigned*)h;
    pool.error_dev = (unsigned*)hd;
  }
  auto it = pool.streams.find(s);
  if (it == pool.streams.end()) {
    if (pool.streams.size() >= (size_t)(SLOT_RING / SLOTS_PER_STREAM)) return false;
    it = pool.streams.emplace(s, (unsigned)(pool.streams.size() * SLOTS_PER_STREAM) << 1).first;
  }
  const unsigned first = it->second >> 1, turn = it->second & 1;
  it->second ^= 1;
  unsigned t = g_token.fetch_add(1);
  if (t == 0) t = g_token.fetch_add(1);          // (0 is what the pool was cleared to)
  out->token = t;
  out->v = pool.base + (size_t)(first + turn) * 2 * SLOT_CH;
  out->error = pool.error_dev;
  return true;
}
}  // namespace

// 0, or 1 with lidal_last_error() set: a fused BatchNorm launch on the current device gave up waiting for its merged
// values (see `Slots`).  Sticky until lidal_bn_set_fused() is called; checked by every BatchNorm entry point that can
// take the fused form and by lidal_plan_run.
extern "C" int lidal_bn_check_device(void) {
  const int d = current_device();
  const unsigned* e = g_pool[d].error_host;
  if (e == nullptr) return 0;
  const unsigned t = *(const volatile unsigned*)e;
  if (t == 0) return 0;
  set_error("BatchNorm: the launch with token %u timed out waiting for the values its first workgroups merge (in-order "
            "workgroup dispatch did not hold, or the device was stalled for ~30 s); its outputs hold NaN.  "
            "lidal_bn_set_fused(0) / LIDAL_BN_FUSED=0 selects the separate merge launches", t);
  return 1;
}

// test / A-B aid: 1 = bf16 lidal_bn_bwd takes its sums as f32 slab sums (default), 0 = the f64 partial sums of the f32 mode.
// Returns the previous setting.
extern "C" int lidal_bn_set_slab_sums(int on) {
  const int was = slab_sums_enabled() ? 1 : 0;
  g_slab_sums = on ? 1 : 0;
  return was;
}

// test / A-B aid: 1 = merge kernels inside their consumers (default), 0 = separate launches
extern "C" int lidal_bn_set_fused(int on) {
  g_fused.store(on ? 1 : 0);
  for (int d = 0; d < MAX_DEVICES; ++d)          // (also acknowledges a reported time-out)
    if (g_pool[d].error_host != nullptr) *(volatile unsigned*)g_pool[d].error_host = 0;
  return 0;
}

namespace {
// the merge of the backward sums and the dx pass as one launch; false: not taken (the caller launches them apart)
template <typename T, bool TILES>
bool bn_bwd_merge_dx(const void* x, const void* dy, int64_t ldy, int64_t n, int c, const float* gamma, const float* beta,
                     int relu, const float* mean, const float* invstd, void* dx, float* ggamma, float* gbeta,
                     const void* part, int nparts, hipStream_t s) {
  Slots slots;
  if (dx == nullptr || !bn_fused() || c > SLOT_CH || !next_slots(&slots, s)) return false;
  if (relu)
    bn_bwd_dx_merge_kernel<T, TILES, true><<<nslabs_ew(n, (int64_t)c * sizeof(T)), NT, 0, s>>>(
        (const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu, part, nparts, gbeta, ggamma, (T*)dx,
        rows_per_wg_ew(n, (int64_t)c * sizeof(T)), ldy, slots);
  else
    bn_bwd_dx_merge_kernel<T, TILES, false><<<nslabs_ew(n, (int64_t)c * sizeof(T)), NT, 0, s>>>(
        (const T*)x, (const T*)dy, n, c, mean, invstd, gamma, beta, relu, part, nparts, gbeta, ggamma, (T*)dx,
        rows_per_wg_ew(n, (int64_t)c * sizeof(T)), ldy, slots);
  return true;
}
}  // namespace

// statistics already reduced per 128-row tile by the producing convolution (conv_img.hip,
// store_tile): merge the tiles, then normalise -- no statistics pass over x
extern "C" int lidal_bn_train_fwd_tiles(const void* x, int dtype, int64_t n, int c, const float* gamma,
                                        const float* beta, float eps, float momentum,
                                        float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, int relu,
                                        const void* residual, void* y, float* save_mean,
                                        float* save_invstd, const float* tile_stats, int64_t n_tiles,
                                        void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0 && n_tiles > 0 && tile_stats != nullptr, "bn_train_fwd_tiles: needs rows and tile statistics");
  hipStream_t s = (hipStream_t)stream;
  Slots slots;
  if (bn_fused() && c <= SLOT_CH && next_slots(&slots, s)) {
    const int flags = (relu & 1) | (residual != nullptr ? 2 : 0) | ((residual != nullptr && (relu & 2)) ? 4 : 0);
#define LIDAL_APPLY_TILES(T, FV)                                                                                          \
    bn_apply_tiles_kernel<T, EW_THREADS, FV><<<nslabs_apply(n, (int64_t)c * sizeof(T)), EW_THREADS, 0, s>>>(               \
        (const T*)x, n, c, tile_stats, (int)n_tiles, eps, momentum, save_mean, save_invstd, running_mean, running_var,    \
        (long long*)num_batches_tracked, gamma, beta, (const T*)residual, (T*)y, rows_per_wg_apply(n, (int64_t)c * sizeof(T)), \
        slots)
#define LIDAL_APPLY_TILES_F(T)                                                        \
    switch (flags) {                                                                  \
      case 0: LIDAL_APPLY_TILES(T, 0); break; case 1: LIDAL_APPLY_TILES(T, 1); break; \
      case 2: LIDAL_APPLY_TILES(T, 2); break; case 3: LIDAL_APPLY_TILES(T, 3); break; \
      case 6: LIDAL_APPLY_TILES(T, 6); break; default: LIDAL_APPLY_TILES(T, 7); break; \
    }
    if (dtype == LIDAL_F32) { LIDAL_APPLY_TILES_F(float) } else { LIDAL_APPLY_TILES_F(__bf16) }
#undef LIDAL_APPLY_TILES_F
#undef LIDAL_APPLY_TILES
    LIDAL_CHECK_LAUNCH("bn_apply(tiles merged in the launch)");
    return 0;
  }
  bn_tiles_final_kernel<<<(unsigned)c, NT, 0, s>>>(tile_stats, (int)n_tiles, c, eps, momentum, save_mean,
                                                   save_invstd, running_mean, running_var,
                                                   (long long*)num_batches_tracked);
  LIDAL_CHECK_LAUNCH("bn_stats_final(tiles)");
  if (dtype == LIDAL_F32)
    bn_apply_kernel<float, false><<<nslabs_apply(n, (int64_t)c * 4), NT, 0, s>>>((const float*)x, n, c, save_mean, save_invstd,
                                                              gamma, beta, eps, relu,
                                                              (const float*)residual, (float*)y,
                                                              rows_per_wg_apply(n, (int64_t)c * 4));
  else
    bn_apply_kernel<__bf16, false><<<nslabs_apply(n, (int64_t)c * 2), NT, 0, s>>>((const __bf16*)x, n, c, save_mean,
                                                               save_invstd, gamma, beta, eps, relu,
                                                               (const __bf16*)residual, (__bf16*)y,
                                                               rows_per_wg_apply(n, (int64_t)c * 2));
  LIDAL_CHECK_LAUNCH("bn_apply");
  return 0;
}

extern "C" int lidal_bn_eval_fwd(const void* x, int dtype, int64_t n, int c, const float* gamma,
                                 const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int relu, void* y,
                                 void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32)
    bn_apply_kernel<float, true><<<nslabs_apply(n, (int64_t)c * 4), NT, 0, s>>>(
        (const float*)x, n, c, running_mean, running_var, gamma, beta, eps, relu, nullptr, (float*)y,
        rows_per_wg_apply(n, (int64_t)c * 4));
  else
    bn_apply_kernel<__bf16, true><<<nslabs_apply(n, (int64_t)c * 2), NT, 0, s>>>(
        (const __bf16*)x, n, c, running_mean, running_var, gamma, beta, eps, relu, nullptr, (__bf16*)y,
        rows_per_wg_apply(n, (int64_t)c * 2));
  LIDAL_CHECK_LAUNCH("lidal_bn_eval_fwd");
  return 0;
}

extern "C" int lidal_bn_bwd(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c,
                            const float* gamma, const float* beta, int relu,
                            const float* save_mean, const float* save_invstd, void* dx,
                            float* grad_gamma, float* grad_beta, void* ws, int64_t ws_bytes,
                            void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "bn_bwd: needs at least one row");
  LIDAL_REQUIRE(ws_bytes >= lidal_bn_workspace_bytes(n, c), "bn workspace too small");
  const int vec = dtype == LIDAL_F32 ? 4 : 8;
  LIDAL_REQUIRE(dy_stride >= c && dy_stride % vec == 0, "bn_bwd: dy row stride %lld (rows of %d, 16-byte steps)",
                (long long)dy_stride, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32)
    return bn_bwd<float>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx, grad_gamma,
                         grad_beta, (double*)ws, s);
  return bn_bwd<__bf16>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx, grad_gamma,
                        grad_beta, (double*)ws, s);
}

extern "C" int lidal_add_relu_bwd_bn_sums(const void* out, const void* g, void* gm, int dtype, int64_t n, int c,
                                          const void* x_a, const float* mean_a, const float* invstd_a, void* part_a,
                                          const void* x_b, const float* mean_b, const float* invstd_b, void* part_b,
                                          int64_t part_bytes, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "add_relu_bwd_bn_sums: needs at least one row");
  LIDAL_REQUIRE(part_bytes >= lidal_bn_workspace_bytes(n, c), "add_relu_bwd_bn_sums: partial buffers too small");
  LIDAL_REQUIRE(x_a != nullptr && part_a != nullptr && (x_b == nullptr || part_b != nullptr),
                "add_relu_bwd_bn_sums: a BatchNorm input without a buffer for its partial sums");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32)
    return bn_bwd_masked_partials<float>(out, g, gm, n, c, x_a, mean_a, invstd_a, (double*)part_a, x_b, mean_b, invstd_b,
                                         (double*)part_b, s);
  return bn_bwd_masked_partials<__bf16>(out, g, gm, n, c, x_a, mean_a, invstd_a, (double*)part_a, x_b, mean_b, invstd_b,
                                        (double*)part_b, s);
}

extern "C" int lidal_bn_bwd_from_sums(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c,
                                      const float* gamma, const float* beta, int relu, const float* save_mean,
                                      const float* save_invstd, void* dx, float* grad_gamma, float* grad_beta,
                                      const void* part, int64_t part_bytes, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "bn_bwd_from_sums: needs at least one row");
  LIDAL_REQUIRE(part_bytes >= lidal_bn_workspace_bytes(n, c), "bn_bwd_from_sums: partial buffer too small");
  const int vec = dtype == LIDAL_F32 ? 4 : 8;
  LIDAL_REQUIRE(dy_stride >= c && dy_stride % vec == 0, "bn_bwd_from_sums: dy row stride %lld (rows of %d, 16-byte steps)",
                (long long)dy_stride, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32)
    return bn_bwd_from_partials<float>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx, grad_gamma,
                                       grad_beta, (const double*)part, s);
  return bn_bwd_from_partials<__bf16>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx, grad_gamma,
                                      grad_beta, (const double*)part, s);
}

// The backward sums came with dy from the data-gradient launch that produced it (conv_img.hip, BnBwd: f32
// (sum dy', sum dy' xhat) per 128-row tile and column): merge the tiles in f64, then the dx pass.  No
// bn_bwd_partial pass over x and dy.


// The tail of a residual block backwards on the levels with many rows: gm = g * (out > 0) and, per part (a slab of
// rows: lidal_bn_tail_parts of them) and channel, the f32 pairs (sum gm, sum gm xhat) of the BatchNorm over x_a -- and
// over x_b, if given -- as [channel][part][2]: what lidal_bn_bwd_tiles takes as tile sums (n_tiles = the parts).
extern "C" int64_t lidal_bn_tail_parts(int64_t n, int c, int dtype) {
  return nslabs_ew(n, (int64_t)c * (dtype == LIDAL_F32 ? 4 : 2));
}
extern "C" int lidal_add_relu_bwd_bn_tile_sums(const void* out, const void* g, void* gm, int dtype, int64_t n, int c,
                                               const void* x_a, const float* mean_a, const float* invstd_a, float* sums_a,
                                               const void* x_b, const float* mean_b, const float* invstd_b, float* sums_b,
                                               int64_t n_parts, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "add_relu_bwd_bn_tile_sums: needs at least one row");
  LIDAL_REQUIRE(n_parts == lidal_bn_tail_parts(n, c, dtype), "add_relu_bwd_bn_tile_sums: %lld parts, lidal_bn_tail_parts says %lld",
                (long long)n_parts, (long long)lidal_bn_tail_parts(n, c, dtype));
  LIDAL_REQUIRE(x_a != nullptr && sums_a != nullptr && (x_b == nullptr || sums_b != nullptr),
                "add_relu_bwd_bn_tile_sums: a BatchNorm input without a buffer for its sums");
  hipStream_t s = (hipStream_t)stream;
  const int64_t row_bytes = (int64_t)c * (dtype == LIDAL_F32 ? 4 : 2);
  const int rpw = rows_per_wg_ew(n, row_bytes);
  const unsigned grid = (unsigned)n_parts;
#define LIDAL_TAIL(T, DUALV)                                                                                         \
  tail_tile_sums_kernel<T, DUALV><<<grid, NT, 0, s>>>((const T*)out, (const T*)g, (T*)gm, n, c, (const T*)x_a, mean_a, \
                                                      invstd_a, sums_a, (const T*)x_b, mean_b, invstd_b, sums_b, rpw)
  if (dtype == LIDAL_F32) { if (x_b) LIDAL_TAIL(float, true); else LIDAL_TAIL(float, false); }
  else { if (x_b) LIDAL_TAIL(__bf16, true); else LIDAL_TAIL(__bf16, false); }
#undef LIDAL_TAIL
  LIDAL_CHECK_LAUNCH("add_relu_bwd_bn_tile_sums");
  return 0;
}

extern "C" int lidal_bn_bwd_tiles(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c,
                                  const float* gamma, const float* beta, int relu, const float* save_mean,
                                  const float* save_invstd, void* dx, float* grad_gamma, float* grad_beta,
                                  const float* tile_sums, int64_t n_tiles, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0 && n_tiles > 0 && tile_sums != nullptr, "bn_bwd_tiles: needs rows and tile sums");
  const int vec = dtype == LIDAL_F32 ? 4 : 8;
  LIDAL_REQUIRE(dy_stride >= c && dy_stride % vec == 0, "bn_bwd_tiles: dy row stride %lld (rows of %d, 16-byte steps)",
                (long long)dy_stride, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LIDAL_F32 ? bn_bwd_merge_dx<float, true>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx,
                                                        grad_gamma, grad_beta, tile_sums, (int)n_tiles, s)
                         : bn_bwd_merge_dx<__bf16, true>(x, dy, dy_stride, n, c, gamma, beta, relu, save_mean, save_invstd, dx,
                                                         grad_gamma, grad_beta, tile_sums, (int)n_tiles, s)) {
    LIDAL_CHECK_LAUNCH("bn_bwd_dx(tile sums merged in the launch)");
    return 0;
  }
  bn_bwd_tiles_final_kernel<<<(unsigned)c, NT, 0, s>>>(tile_sums, (int)n_tiles, c, grad_beta, grad_gamma);
  LIDAL_CHECK_LAUNCH("bn_bwd_final(tiles)");
  if (dx != nullptr) {
    if (dtype == LIDAL_F32)
      bn_bwd_dx_kernel<float><<<nslabs_ew(n, (int64_t)c * 4), NT, 0, s>>>((const float*)x, (const float*)dy, n, c, save_mean,
                                                          save_invstd, gamma, beta, relu, grad_beta, grad_gamma,
                                                          (float*)dx, rows_per_wg_ew(n, (int64_t)c * 4), dy_stride);
    else
      bn_bwd_dx_kernel<__bf16><<<nslabs_ew(n, (int64_t)c * 2), NT, 0, s>>>((const __bf16*)x, (const __bf16*)dy, n, c, save_mean,
                                                           save_invstd, gamma, beta, relu, grad_beta, grad_gamma,
                                                           (__bf16*)dx, rows_per_wg_ew(n, (int64_t)c * 2), dy_stride);
    LIDAL_CHECK_LAUNCH("bn_bwd_dx");
  }
  return 0;
}

extern "C" int lidal_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, int c, float* scale,
                             float* shift, void* stream) {
  if (c <= 0) return 0;
  bn_fold_kernel<<<(unsigned)cdiv(c, 256), 256, 0, (hipStream_t)stream>>>(
      gamma, beta, running_mean, running_var, eps, c, scale, shift);
  LIDAL_CHECK_LAUNCH("lidal_bn_fold");
  return 0;
}

// Column sums of a [n, c] matrix (bias gradients of the dense layers): one pass, f64 sums per thread in row order, the
// LDS tree of the statistics kernels, pairs (sum, 0) per (workgroup, channel) for bn_bwd_final_kernel.  (Rounds 1-4 ran
// bn_bwd_partial_kernel with x = dy = the matrix, mean 0: two reads of every row and the BatchNorm arithmetic around a
// plain sum -- 72 us for the 396 662 x 256 gradient of the point branch.)  Sixteen loads in flight per thread: at one
// workgroup per CU a reducing kernel is short of bytes in flight (profiles/README.md, round 5).
constexpr int UNR_C = 16;
template <typename T>
__global__ void __launch_bounds__(NT) colsum_partial_kernel(const T* __restrict__ x, int64_t n, int c,
                                                            double* __restrict__ part, int rpw) {
  constexpr int VEC = IO<T>::VEC;
  extern __shared__ double sh[];                // [NT][VEC]
  const int cg_n = c / VEC, rpi = NT / cg_n;
  const int tid = threadIdx.x, cg = tid % cg_n, rl = tid / cg_n;
  const int64_t r_beg = (int64_t)blockIdx.x * rpw;
  const int64_t r_end = (r_beg + rpw < n) ? r_beg + rpw : n;
  double s1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = 0.;
  if (rl < rpi) {
    int64_t r = r_beg + rl;
    for (; r + (UNR_C - 1) * rpi < r_end; r += UNR_C * rpi) {
      typename IO<T>::vec v[UNR_C];
#pragma unroll
      for (int u = 0; u < UNR_C; ++u)
        v[u] = *reinterpret_cast<const typename IO<T>::vec*>(x + (r + u * rpi) * c + cg * VEC);
#pragma unroll
      for (int u = 0; u < UNR_C; ++u) {
        float f[VEC];
        IO<T>::unpack(v[u], f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s1[i] += (double)f[i];
      }
    }
    for (; r < r_end; r += rpi) {
      float f[VEC];
      IO<T>::unpack(*reinterpret_cast<const typename IO<T>::vec*>(x + r * c + cg * VEC), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s1[i] += (double)f[i];
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) sh[tid * VEC + i] = s1[i];
  tree_sum_rows<VEC, double>(sh, tid, cg_n, rpi, rl);
  if (rl == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double* dst = part + ((int64_t)blockIdx.x * c + cg * VEC + i) * 2;
      dst[0] = sh[tid * VEC + i]; dst[1] = 0.;
    }
  }
}

extern "C" int lidal_colsum(const void* x, int dtype, int64_t n, int c, float* out, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (int rc = bn_check(n, c, dtype)) return rc;
  LIDAL_REQUIRE(n > 0, "colsum: needs at least one row");
  LIDAL_REQUIRE(ws_bytes >= lidal_bn_workspace_bytes(n, c) + 3 * (int64_t)c * 4, "colsum ws too small");
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)ws;
  float* zeros = (float*)((char*)ws + (lidal_bn_workspace_bytes(n, c) / 8) * 8);
  float* ones = zeros + c;
  float* scratch = ones + c;
  (void)zeros;
  int np = nparts_for(n);
  if (dtype == LIDAL_F32)
    colsum_partial_kernel<float><<<np, NT, NT * 4 * sizeof(double), s>>>((const float*)x, n, c, part, rows_per_wg(n));
  else
    colsum_partial_kernel<__bf16><<<np, NT, NT * 8 * sizeof(double), s>>>((const __bf16*)x, n, c, part, rows_per_wg(n));
  LIDAL_CHECK_LAUNCH("colsum_partial");
  bn_bwd_final_kernel<<<(unsigned)cdiv(c, 8), NT, 0, s>>>(part, np, c, out, scratch);
  LIDAL_CHECK_LAUNCH("colsum_final");
  return 0;
}
