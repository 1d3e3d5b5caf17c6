// Weight gradient of the sparse 3D convolution for gfx950 (f32, and bf16 shapes the LDS-DMA kernel of
// wgrad_dma.hip does not serve): gw[k] = A_k^T B_k over the rule list of offset k, as a split-K MFMA GEMM
// (workgroup = (split, k, channel tile)); gathered rows are staged through LDS (converted to f32), f32
// partial slabs are reduced in a fixed order by a second kernel.
//
// (The first generation of the forward / data-gradient kernel, lidal_conv_apply with plain [k][co][ci]
// weights, lived here through round 3; conv_img.hip replaced it in round 2 and it left the product library in
// round 4: tests/native/conv_gen1.hip keeps it as a bitwise cross-check of the image kernels.)
#include <type_traits>

#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// conv_apply workgroup shape, chosen per output-column width NB (16-column blocks per workgroup):
// NW waves x G 16-row groups per wave = 128 output rows either way.  8 waves x 1 group (half the
// accumulators and A fragments per wave, twice the waves to hide the gathers) wins for the 64- and
// 96-column kernels, 4 waves x 2 groups (each weight fragment read from LDS feeds two MFMAs)
// elsewhere in bf16 (scripts/ablate_conv.py: 96->96 137.9 -> 127.3 us, 64->64 35.7 -> 34.5, but 32->32
// 45.7 -> 47.4 and 256->256 at stride 16 106 -> 112); the f32 kernels (MFMA-bound, 4x the MFMA
// issue slots per fragment) gain 6-19 % from 8 x 1 at every width.
constexpr int conv_groups(int nb, bool f32) { return (f32 || nb == 4 || nb == 6) ? 1 : 2; }
constexpr int conv_waves(int nb, bool f32) { return (f32 || nb == 4 || nb == 6) ? 8 : 4; }

// 16 raw bytes of a lane's operand fragment: the A fragments travel through the software pipeline
// in this type (as <8 x bf16> hipcc splits them into halves at every loop-carried value, which
// also drags the wait for the gather to the top of the phase)
typedef int raw4 __attribute__((ext_vector_type(4)));

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int VEC = 4;    // elements per 16-byte lane load
  static constexpr int CH = 16;    // reduction elements consumed per lane-load round (4 lane groups)
  typedef f32x4 frag;
  __device__ static frag zero() { return frag{0.f, 0.f, 0.f, 0.f}; }
  __device__ static float to_f32(float v) { return v; }
  __device__ static float from_f32(float v) { return v; }
};
template <> struct DT<__bf16> {
  static constexpr int VEC = 8;
  static constexpr int CH = 32;
  typedef bf16x8 frag;
  __device__ static frag zero() {
    frag z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
    return z;
  }
  __device__ static float to_f32(__bf16 v) { return (float)v; }
  __device__ static __bf16 from_f32(float v) { return (__bf16)v; }
};

template <typename T>
__device__ __forceinline__ typename DT<T>::frag load_frag_guarded(const T* p, int valid_elems) {
  // valid_elems: how many of the VEC elements starting at p are inside the row
  typedef typename DT<T>::frag frag;
  if (valid_elems >= DT<T>::VEC) return *reinterpret_cast<const frag*>(p);
  frag f = DT<T>::zero();
#pragma unroll
  for (int e = 0; e < DT<T>::VEC; ++e)
    if (e < valid_elems) f[e] = p[e];
  return f;
}

__device__ __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// 512 bytes of zeros in device memory: lanes without a rule (or past the channel range) load from
// here instead of branching around the load, so every gather is an unconditional 16-byte load whose
// result is first touched by the MFMA (no exec-masked control flow, no early vmcnt waits).
__device__ __attribute__((aligned(16))) unsigned char g_zero_page[512];


// ------------------------------------------------------------------------------------------
// wgrad: gw[k][ca][cb] = sum_p a[pa(p)][:]^T b[pb(p)][:]
// ------------------------------------------------------------------------------------------
constexpr int WTHREADS = 256;   // wgrad workgroups: 4 waves as 2 x 2
constexpr int BP = 32;   // rules per staging step (f32 kernel)


// workgroup tile: (2*MI*16) x (2*NI*16) of gw[k]; waves as 2 x 2.
template <typename T, int MI, int NI>
__global__ void __launch_bounds__(WTHREADS)
conv_wgrad_kernel(const T* __restrict__ a, const T* __restrict__ b, const int2* __restrict__ pairs,
                  const int64_t* __restrict__ koff, int a_col, float* __restrict__ partial,
                  int K, int ca, int cb, int tiles_b, int target_chunk) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  constexpr int SA = TA + 4, SB = TB + 4;      // LDS row strides (floats)
  __shared__ __attribute__((aligned(16))) float la[BP * SA];
  __shared__ __attribute__((aligned(16))) float lb[BP * SB];
  __shared__ int pa[BP], pb[BP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row16 = lane & 15, gsel = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = blockIdx.x, nsplit = gridDim.x;
  const int k = blockIdx.y;
  const int ta = blockIdx.z / tiles_b, tb = blockIdx.z - ta * tiles_b;
  const int ca0 = ta * TA, cb0 = tb * TB;

  const int64_t beg = koff[k], end = koff[k + 1];
  const int64_t nk = end - beg;
  const int used = splits_for(nk, nsplit, target_chunk);
  if (split >= used) return;                   // the reducer only reads `used` slabs of offset k
  int64_t chunk = (nk + used - 1) / used;
  chunk = ((chunk + BP - 1) / BP) * BP;
  const int64_t p_beg = beg + (int64_t)split * chunk;
  const int64_t p_end = (p_beg + chunk < end) ? (p_beg + chunk) : end;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t p0 = p_beg; p0 < p_end; p0 += BP) {
    __syncthreads();
    if (tid < BP) {
      int2 pr = make_int2(-1, -1);
      if (p0 + tid < p_end) pr = pairs ? pairs[p0 + tid] : make_int2((int)(p0 + tid), (int)(p0 + tid));
      pa[tid] = a_col ? pr.y : pr.x;
      pb[tid] = a_col ? pr.x : pr.y;
    }
    __syncthreads();
    // stage gathered rows as f32: 4 consecutive channels per thread per step
    for (int i = tid; i < BP * (TA / 4); i += WTHREADS) {
      int p = i / (TA / 4), c = (i - p * (TA / 4)) * 4;
      int src = pa[p];
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (src >= 0) {
        const T* ptr = a + (int64_t)src * ca + ca0 + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ca0 + c + e < ca) v[e] = DT<T>::to_f32(ptr[e]);
      }
      *reinterpret_cast<f32x4*>(la + p * SA + c) = f32x4{v[0], v[1], v[2], v[3]};
    }
    for (int i = tid; i < BP * (TB / 4); i += WTHREADS) {
      int p = i / (TB / 4), c = (i - p * (TB / 4)) * 4;
      int src = pb[p];
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (src >= 0) {
        const T* ptr = b + (int64_t)src * cb + cb0 + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb0 + c + e < cb) v[e] = DT<T>::to_f32(ptr[e]);
      }
      *reinterpret_cast<f32x4*>(lb + p * SB + c) = f32x4{v[0], v[1], v[2], v[3]};
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BP / 4; ++ks) {
      const int p = ks * 4 + gsel;
      float af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = la[p * SA + (wr * MI + mi) * 16 + row16];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = lb[p * SB + (wc * NI + ni) * 16 + row16];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
  }
  // partial[split][k][ca][cb]; D: col = lane&15, row = 4*(lane>>4) + r
  float* dst = partial + ((int64_t)split * K + k) * ca * cb;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = ca0 + (wr * MI + mi) * 16 + gsel * 4 + r;
        int j = cb0 + (wc * NI + ni) * 16 + row16;
        if (i < ca && j < cb) dst[(int64_t)i * cb + j] = acc[mi][ni][r];
      }
}

// gw[i] = sum over the slabs of i's offset, in slab order (fixed => reproducible).  V = 4: one
// float4 per thread and four slab loads in flight; V = 1 for element counts not divisible by 4.
template <int V>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const int64_t* __restrict__ koff,
                                                           float* __restrict__ gw, int K,
                                                           int64_t per_k, int splits,
                                                           int target_chunk) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
  const int64_t total = (int64_t)K * per_k;
  if (i >= total) return;
  const int k = (int)(i / per_k);
  const int used = splits_for(koff[k + 1] - koff[k], splits, target_chunk);
  if constexpr (V == 4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 3 < used; sp += 4) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[u] = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + u) * total + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
    }
    for (; sp < used; ++sp) {
      const float4 x = *reinterpret_cast<const float4*>(partial + (int64_t)sp * total + i);
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
    *reinterpret_cast<float4*>(gw + i) = s;
  } else {
    float s = 0.f;
    for (int sp = 0; sp < used; ++sp) s += partial[(int64_t)sp * total + i];
    gw[i] = s;
  }
}

// ---- bf16 wgrad: v_mfma_f32_16x16x32_bf16 with both operands read TRANSPOSED from LDS ----------
// gw[k][i][j] = sum_p A[p][i] B[p][j]: the reduction index p (rule) is the LDS tile ROW of both
// gathered operands, while the MFMA wants 8 consecutive p per lane.  ds_read_b64_tr_b16 delivers
// exactly that from the row-major tiles the 16-byte gathers produce, so no transposing store is
// needed.  BPB = 64 rules per step (two MFMA k-steps), LDS tiles double-buffered, gathered rows of
// the next step are in flight in registers while this step's MFMAs run; one barrier per step.
constexpr int BPB = 64;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

template <int MI, int NI, bool GUARD>
__global__ void __launch_bounds__(WTHREADS)
conv_wgrad_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b,
                       const int2* __restrict__ pairs, const int64_t* __restrict__ koff, int a_col,
                       float* __restrict__ partial, int K, int ca, int cb, int tiles_b,
                       int target_chunk) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  constexpr int SA = TA + 8, SB = TB + 8;              // LDS row strides (bf16), +16 B pad
  constexpr int SEG_A = TA / 8, SEG_B = TB / 8;        // 16-byte segments per gathered row
  constexpr int PT_A = (BPB * SEG_A) / WTHREADS, PT_B = (BPB * SEG_B) / WTHREADS;
  static_assert((BPB * SEG_A) % WTHREADS == 0 && (BPB * SEG_B) % WTHREADS == 0, "tile/threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* la = reinterpret_cast<__bf16*>(smem);                       // [2][BPB][SA]
  __bf16* lb = la + 2 * BPB * SA;                                     // [2][BPB][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row16 = lane & 15, gsel = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = blockIdx.x, nsplit = gridDim.x;
  const int k = blockIdx.y;
  const int ta = blockIdx.z / tiles_b, tb = blockIdx.z - ta * tiles_b;
  const int ca0 = ta * TA, cb0 = tb * TB;

  const int64_t beg = koff[k], end = koff[k + 1];
  const int64_t nk = end - beg;
  const int used = splits_for(nk, nsplit, target_chunk);
  if (split >= used) return;
  int64_t chunk = (nk + used - 1) / used;
  chunk = ((chunk + BPB - 1) / BPB) * BPB;
  const int64_t p_beg = beg + (int64_t)split * chunk;
  const int64_t p_end = (p_beg + chunk < end) ? (p_beg + chunk) : end;
  const int nsteps = (int)((p_end - p_beg + BPB - 1) / BPB);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: thread t owns segments t, t+256, ... of the [BPB][SEG] tile; row = seg / SEG
  bf16x8 ra[PT_A], rb[PT_B];
  int ia[PT_A], ib[PT_B];                   // gathered row ids of the step being loaded
  auto load_ids = [&](int step) {
    const int64_t p0 = p_beg + (int64_t)step * BPB;
#pragma unroll
    for (int t = 0; t < PT_A; ++t) {
      const int row = (tid + t * WTHREADS) / SEG_A;
      const int64_t pi = p0 + row, pc = pi < p_end ? pi : p_end - 1;     // clamped: no branch
      int2 pr = pairs ? pairs[pc] : make_int2((int)pc, (int)pc);
      if (pi >= p_end) pr = make_int2(-1, -1);
      ia[t] = a_col ? pr.y : pr.x;
    }
#pragma unroll
    for (int t = 0; t < PT_B; ++t) {
      const int row = (tid + t * WTHREADS) / SEG_B;
      const int64_t pi = p0 + row, pc = pi < p_end ? pi : p_end - 1;
      int2 pr = pairs ? pairs[pc] : make_int2((int)pc, (int)pc);
      if (pi >= p_end) pr = make_int2(-1, -1);
      ib[t] = a_col ? pr.x : pr.y;
    }
  };
  auto load_rows = [&]() {
#pragma unroll
    for (int t = 0; t < PT_A; ++t) {
      const int sg = tid + t * WTHREADS, c = (sg % SEG_A) * 8;
      const bool ok = ia[t] >= 0 && ca0 + c < ca;
      if constexpr (GUARD) {
        ra[t] = DT<__bf16>::zero();
        if (ok) ra[t] = load_frag_guarded<__bf16>(a + (int64_t)ia[t] * ca + ca0 + c, ca - ca0 - c);
      } else {      // channels are multiples of 8: whole-vector loads, misses read the zero page
        const __bf16* p = ok ? a + (int64_t)ia[t] * ca + ca0 + c
                             : reinterpret_cast<const __bf16*>(g_zero_page);
        ra[t] = *reinterpret_cast<const bf16x8*>(p);
      }
    }
#pragma unroll
    for (int t = 0; t < PT_B; ++t) {
      const int sg = tid + t * WTHREADS, c = (sg % SEG_B) * 8;
      const bool ok = ib[t] >= 0 && cb0 + c < cb;
      if constexpr (GUARD) {
        rb[t] = DT<__bf16>::zero();
        if (ok) rb[t] = load_frag_guarded<__bf16>(b + (int64_t)ib[t] * cb + cb0 + c, cb - cb0 - c);
      } else {
        const __bf16* p = ok ? b + (int64_t)ib[t] * cb + cb0 + c
                             : reinterpret_cast<const __bf16*>(g_zero_page);
        rb[t] = *reinterpret_cast<const bf16x8*>(p);
      }
    }
  };
  auto store_rows = [&](int buf) {
#pragma unroll
    for (int t = 0; t < PT_A; ++t) {
      const int sg = tid + t * WTHREADS, row = sg / SEG_A, c = (sg % SEG_A) * 8;
      *reinterpret_cast<bf16x8*>(la + (buf * BPB + row) * SA + c) = ra[t];
    }
#pragma unroll
    for (int t = 0; t < PT_B; ++t) {
      const int sg = tid + t * WTHREADS, row = sg / SEG_B, c = (sg % SEG_B) * 8;
      *reinterpret_cast<bf16x8*>(lb + (buf * BPB + row) * SB + c) = rb[t];
    }
  };

  if (nsteps > 0) {
    load_ids(0);
    load_rows();
    if (nsteps > 1) load_ids(1);
    store_rows(0);
  }
  __syncthreads();

  // transposed-read addressing (T10): lane 4q+pp of each 16-lane group supplies the address of tile
  // row (8*gsel + q [+4]) at columns 4*pp..4*pp+3 of the 16-column block; lane i receives column i.
  const int q = row16 >> 2, pp = row16 & 3;
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) {
      load_rows();                                   // rows of step+1 (ids loaded a step earlier)
      if (step + 2 < nsteps) load_ids(step + 2);
    }
    const __bf16* ta_ = la + buf * BPB * SA;
    const __bf16* tb_ = lb + buf * BPB * SB;
#pragma unroll
    for (int ks = 0; ks < BPB / 32; ++ks) {
      const int prow = ks * 32 + 8 * gsel + q;
      bf16x8 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const __bf16* base = ta_ + prow * SA + (wr * MI + mi) * 16 + 4 * pp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(base + 4 * SA));
        af[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const __bf16* base = tb_ + prow * SB + (wc * NI + ni) * 16 + 4 * pp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(base + 4 * SB));
        bf[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    if (step + 1 < nsteps) store_rows(buf ^ 1);
    __syncthreads();
  }
  float* dst = partial + ((int64_t)split * K + k) * ca * cb;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = ca0 + (wr * MI + mi) * 16 + gsel * 4 + r;
        int j = cb0 + (wc * NI + ni) * 16 + row16;
        if (i < ca && j < cb) dst[(int64_t)i * cb + j] = acc[mi][ni][r];
      }
}

template <typename T, int MI, int NI>
int launch_wgrad(const void* a, const void* b, const int* pairs, const int64_t* koff, int a_col,
                 float* partial, int splits, int target_chunk, int K, int ca, int cb,
                 hipStream_t s) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  int tiles_a = (int)cdiv(ca, TA), tiles_b = (int)cdiv(cb, TB);
  dim3 grid((unsigned)splits, (unsigned)K, (unsigned)(tiles_a * tiles_b));
  if constexpr (sizeof(T) == 2) {
    const size_t lds = 2 * BPB * ((TA + 8) + (TB + 8)) * sizeof(__bf16);
    const bool guard = (ca % 8 != 0) || (cb % 8 != 0);
    auto kern = guard ? conv_wgrad_bf16_kernel<MI, NI, true> : conv_wgrad_bf16_kernel<MI, NI, false>;
    static size_t attr_set[2][MAX_DEVICES] = {};
    const int dev = current_device();
    if (attr_set[guard][dev] < lds) {
      LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set[guard][dev] = lds;
    }
    kern<<<grid, WTHREADS, lds, s>>>((const __bf16*)a, (const __bf16*)b, (const int2*)pairs, koff,
                                     a_col, partial, K, ca, cb, tiles_b, target_chunk);
  } else {
    conv_wgrad_kernel<T, MI, NI><<<grid, WTHREADS, 0, s>>>((const T*)a, (const T*)b,
                                                           (const int2*)pairs, koff, a_col, partial,
                                                           K, ca, cb, tiles_b, target_chunk);
  }
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad");
  return 0;
}


template <typename T>
int dispatch_wgrad(const void* a, const void* b, const int* pairs, const int64_t* koff, int a_col,
                   float* partial, int splits, int target_chunk, int K, int ca, int cb,
                   hipStream_t s) {
  int mi = wgrad_blocks(ca), ni = wgrad_blocks(cb);
#define WG_CASE(M, N) \
  if (mi == M && ni == N) return launch_wgrad<T, M, N>(a, b, pairs, koff, a_col, partial, splits, target_chunk, K, ca, cb, s);
  WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
  WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
  WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4)
  WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 3) WG_CASE(4, 4)
#undef WG_CASE
  set_error("wgrad: no tile for %d x %d", ca, cb);
  return 2;
}

}  // namespace


// Split-K plan of the register kernels (f32, and bf16 shapes the DMA kernel does not serve): slabs
// of `chunk` rules, sized so that the launch is about ONE round of resident workgroups -- slabs =
// rules / chunk, workgroups = slabs x channel tiles.  The rule count is only known on the device,
// so it is estimated from the rows (~6 rules per row for a 3x3x3 map on LiDAR surfaces, exactly 1
// for the 2x2x2 maps and the dense layers).  scripts/ablate_wgrad.py (round 1): 4096 is best for
// 96->96 on 397k rows, 1024 for 128->128 on 105k rows.
static void splitk_plan(int64_t n_rows, int k, int ca, int cb, int* splits, int* chunk) {
  const int64_t tiles = cdiv(ca, wgrad_blocks(ca) * 32) * cdiv(cb, wgrad_blocks(cb) * 32);
  int64_t c = (k > 8 ? 6 : 1) * n_rows * tiles / 512;
  c = align_up(c < 1 ? 1 : c, 64);
  c = c > 4096 ? 4096 : (c < 512 ? 512 : c);
  int64_t s = cdiv(n_rows, c);          // no offset has more rules than the larger table has rows
  *splits = (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
  *chunk = (int)c;
}

extern "C" int64_t lidal_conv_wgrad_slabs(int64_t n_a, int64_t n_b, int k, int ca, int cb, int dtype) {
  const int64_t n_rows = n_a > n_b ? n_a : n_b;
  if (dtype == LIDAL_F32_SPLIT) {       // f32 operands, split form: W + k slabs, then room for the operands' bf16 pieces
    if (!wgrad_split_serves(n_a, n_b, k, ca, cb)) return -1;
    const int64_t slab = (int64_t)ca * cb * 4;
    return (int64_t)wgrad_split_workgroups(n_a, n_b, k, ca, cb) + k + cdiv(wgrad_split_scratch_bytes(n_a, n_b, ca, cb) + 256, slab);
  }
  if (dtype == LIDAL_BF16 && wgrad_dma_serves(n_a, n_b, k, ca, cb))
    return (int64_t)wgrad_dma_workgroups(n_a, n_b, k, ca, cb) + k;
  int splits, chunk;
  splitk_plan(n_rows, k, ca, cb, &splits, &chunk);
  return (int64_t)splits * k;
}

extern "C" int lidal_conv_wgrad(const void* a, const void* b, int64_t n_a, int64_t n_b,
                                const int32_t* pairs, const int64_t* koff, int a_col, float* gw,
                                float* partial, int64_t n_slabs, int k, int ca, int cb, int dtype,
                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (k == 0 || ca == 0 || cb == 0) return 0;
  LIDAL_REQUIRE(n_a >= 0 && n_b >= 0, "wgrad: negative row count");
  LIDAL_REQUIRE(dtype == LIDAL_F32 || dtype == LIDAL_BF16 || dtype == LIDAL_F32_SPLIT, "wgrad: bad dtype %d", dtype);
  LIDAL_REQUIRE(dtype != LIDAL_F32_SPLIT || wgrad_split_serves(n_a, n_b, k, ca, cb),
                "wgrad(split): channel counts must be multiples of 8 and the split operands below 4 GiB (ca=%d cb=%d)", ca, cb);
  LIDAL_REQUIRE(n_slabs >= lidal_conv_wgrad_slabs(n_a, n_b, k, ca, cb, dtype),
                "wgrad: scratch of %lld slabs, lidal_conv_wgrad_slabs asks for %lld", (long long)n_slabs,
                (long long)lidal_conv_wgrad_slabs(n_a, n_b, k, ca, cb, dtype));
  const int64_t n_rows = n_a > n_b ? n_a : n_b;
  if (dtype == LIDAL_F32_SPLIT) {       // f32 a, b: their bf16 pieces go behind the W + k slabs of `partial`
    const int W = wgrad_split_workgroups(n_a, n_b, k, ca, cb);
    char* scratch = (char*)partial + (int64_t)(W + k) * ca * cb * 4;
    scratch = (char*)align_up((int64_t)(uintptr_t)scratch, 256);
    return wgrad_split(a, b, n_a, n_b, pairs, koff, a_col, gw, partial, scratch, W, k, ca, cb, s);
  }
  if (dtype == LIDAL_BF16 && wgrad_dma_serves(n_a, n_b, k, ca, cb))
    return wgrad_dma(a, b, n_a, n_b, pairs, koff, a_col, gw, partial,
                     wgrad_dma_workgroups(n_a, n_b, k, ca, cb), k, ca, cb, s);
  int splits, target_chunk, rc;
  splitk_plan(n_rows, k, ca, cb, &splits, &target_chunk);
  if (dtype == LIDAL_F32)
    rc = dispatch_wgrad<float>(a, b, pairs, koff, a_col, partial, splits, target_chunk, k, ca, cb, s);
  else       // channel counts that are not whole 16-byte segments: the register kernel
    rc = dispatch_wgrad<__bf16>(a, b, pairs, koff, a_col, partial, splits, target_chunk, k, ca, cb, s);
  if (rc) return rc;
  int64_t n = (int64_t)k * ca * cb;
  if (((int64_t)ca * cb) % 4 == 0)
    wgrad_reduce_kernel<4><<<(unsigned)cdiv(n / 4, 256), 256, 0, s>>>(partial, koff, gw, k,
                                                                      (int64_t)ca * cb, splits,
                                                                      target_chunk);
  else
    wgrad_reduce_kernel<1><<<(unsigned)cdiv(n, 256), 256, 0, s>>>(partial, koff, gw, k,
                                                                  (int64_t)ca * cb, splits,
                                                                  target_chunk);
  LIDAL_CHECK_LAUNCH("wgrad_reduce");
  return 0;
}

extern "C" int lidal_conv_wgrad_streams_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  return wgrad_stream_serves(n_a, n_b, k, ca, cb) ? 1 : 0;
}

extern "C" int lidal_conv_wgrad_streams(const void* a, const void* b, int64_t n_a, int64_t n_b, const int32_t* spairs,
                                        const int32_t* sdesc, int n_wg, int a_col, float* gw, float* partial,
                                        int64_t n_slabs, int k, int ca, int cb, int dtype, void* stream) {
  if (k == 0 || ca == 0 || cb == 0) return 0;
  LIDAL_REQUIRE(dtype == LIDAL_BF16, "wgrad(streams): bf16 operands only (dtype %d)", dtype);
  LIDAL_REQUIRE(n_a >= 0 && n_b >= 0, "wgrad(streams): negative row count");
  LIDAL_REQUIRE(wgrad_stream_serves(n_a, n_b, k, ca, cb),
                "wgrad(streams): one channel tile of whole 16-byte segments (ca=%d cb=%d <= 128, multiples of 8)", ca, cb);
  LIDAL_REQUIRE(n_wg > 0 && n_wg % 8 == 0, "wgrad(streams): %d workgroups (a positive multiple of 8)", n_wg);
  LIDAL_REQUIRE(n_slabs >= 2ll * n_wg, "wgrad(streams): scratch of %lld slabs, 2 x %d workgroups needed", (long long)n_slabs, n_wg);
  return wgrad_stream(a, b, n_a, n_b, spairs, sdesc, n_wg, a_col, gw, partial, k, ca, cb, (hipStream_t)stream);
}
