// bf16 weight gradient of the sparse convolution, gathers by LDS-DMA.
//
//   gw[k][i][j] = sum over the rules p of offset k of  a[pa(p)][i] * b[pb(p)][j]
//
// (torchsparse convolution_backward_cuda's grad_weight loop; the rule lists are lidal_kmap_build's
// pairs/koff).  conv.hip's first bf16 kernel staged the gathered rows global -> registers -> LDS with
// ONE 64-rule stage in flight per workgroup: with 460 workgroups of 64 dependent steps on 256 CUs
// the launch moved 722 MB in 156 us and did not get faster when every gather hit L2 (144 us, the
// timing probe of round 2) -- a latency chain, 45 KB in flight per CU where HBM needs ~64 KB.
//
// This kernel keeps D stages in flight per workgroup without holding them in registers:
//   * each of the 4 waves issues its share of a stage's rows as `buffer_load_dwordx4 ... lds`
//     (16 B per lane, straight into the LDS ring of D+1 stages); rules past the slab end and channel
//     segments past the row aim out of the descriptor's range: zeros, no memory access;
//   * the row ids come from the pairs list by the same DMA, 2*D stages ahead, into a small LDS ring:
//     vector-memory operations retire in order, so an id load issued right before the rows that need
//     it would drain the whole queue; issued a full D stages earlier it is covered by the wait for
//     rows(s) that the step makes anyway;
//   * one s_waitcnt vmcnt((D-1) * loads per stage) + one barrier per step;
//   * the DMA fills LDS linearly (lane i -> byte 16*i), so the +16 B row padding of the register
//     path is not available; instead the 16-byte segments of a row are permuted as a function of
//     row bits 0, 1, 3 (swz below) so that the 16 segments one ds_read_b64_tr_b16 pass touches (rows
//     r..r+3 and r+8..r+11, two segments each) fall into 16 different bank groups.  The DMA applies
//     the inverse permutation for free: each lane aims at any global address.
// The DMA is issued from inline assembly: hipcc orders every LDS read after ALL earlier LDS-DMA
// writes (vmcnt(0)), which would serialise the ring; here the waits are the explicit ones above.
//
//
// Decomposition.  The stages of all offsets (64 rules each, offset after offset) form ONE sequence of
// T stages, cut into W equal runs, one per workgroup; W = the workgroups the chip holds at once
// (bounded so that a run keeps >= 16 stages), so the launch is a single, evenly loaded round -- the
// (split, offset) grid of conv.hip left the CUs with 1..3 slabs each and its best slab size jumped
// around with the rule count (scripts/ablate_wgrad.py).  A run that crosses an offset boundary flushes
// its accumulators and goes on; workgroup w's piece of offset k lands in slab w + k (distinct for
// every piece: w and k never both stand still), and the reducer adds the slabs w_first(k)..w_last(k)
// of each offset in that fixed order: bitwise reproducible, no atomics.  T, the cut points and the
// slab ranges are recomputed from koff by both kernels -- nothing travels through the host.
#include <stdlib.h>

#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

/* stages in flight per workgroup.  1 (the next stage travels while this one is multiplied; two LDS
   slots) measured best at every layer shape of the model -- scripts/ablate_wgrad.py, 396k..16k rows:
   96->96 135 us against 154 at depth 2, 128->128 67 against 109, 256->256 99 against 157 -- the
   deeper rings cost a resident workgroup per CU and their extra lines in flight thrash the 4 MB L2 */
constexpr int wgrad_depth(int /*stage_bytes*/) { return 1; }

constexpr int WT = 256;        // 4 waves as 2 x 2
constexpr int STREAM_HDR = 4;   // i32 words ahead of soff in a stream descriptor (lidal_wgrad_streams_build)
constexpr int RPS = 64;        // rules per stage (two MFMA k-steps)
constexpr unsigned OOB = 0xFFFFFFF0u;

// physical 16-byte segment of logical segment `seg` of tile row `row` (SEG segments per row)
template <int SEG>
__device__ __forceinline__ int swz(int row, int seg) {
  const int b3 = (row >> 3) & 1;
  if constexpr (SEG == 4) return seg ^ (2 * b3);
  if constexpr (SEG == 8) return seg ^ (4 * ((row >> 1) & 1)) ^ (2 * b3);
  if constexpr (SEG == 12) { int s = seg + 2 * b3; return s >= 12 ? s - 12 : s; }
  if constexpr (SEG == 16) return seg ^ (4 * (row & 3)) ^ (2 * b3);
  return seg;
}
template <int SEG>
__device__ __forceinline__ int unswz(int row, int phys) {
  if constexpr (SEG == 12) { int s = phys - 2 * ((row >> 3) & 1); return s < 0 ? s + 12 : s; }
  return swz<SEG>(row, phys);          // the xor maps are involutions
}

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  return u32x4{(unsigned)p, (unsigned)(p >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
// 16 B (4 B) per lane from rsrc + voff into LDS at lds_base + 16 (4) * lane; out of range -> zeros
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma4(u32x4 rsrc, unsigned lds_base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

// stages of offset k
__device__ __forceinline__ int stages_of(const int64_t* koff, int k) {
  return (int)((koff[k + 1] - koff[k] + RPS - 1) / RPS);
}

template <int MI, int NI, bool DENSE>
__global__ void __launch_bounds__(WT)
wgrad_dma_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, unsigned a_bytes,
                 unsigned b_bytes, const int2* __restrict__ pairs, const int64_t* __restrict__ koff,
                 int a_col, float* __restrict__ partial, int K, int ca, int cb, int tiles_b) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  constexpr int SEG_A = TA / 8, SEG_B = TB / 8;
  constexpr int A_BYTES = RPS * TA * 2, B_BYTES = RPS * TB * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int D = wgrad_depth(STAGE), R = D + 1;
  constexpr int IDS_R = 2 * D + 1, IDS_BYTES = RPS * 8;
  constexpr int IA = SEG_A / 4, IB = SEG_B / 4;          // DMA instructions per wave and stage
  constexpr int PER_STAGE = IA + IB + (DENSE ? 0 : 1);
  constexpr int INFLIGHT = (D - 1) * PER_STAGE;
  static_assert(INFLIGHT < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [R][STAGE] [IDS_R][IDS_BYTES]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned ids0 = lds0 + R * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row16 = lane & 15, gsel = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int w = blockIdx.x, W = gridDim.x;
  const int ta = blockIdx.y / tiles_b, tb = blockIdx.y - ta * tiles_b;
  const int ca0 = ta * TA, cb0 = tb * TB;

  // this workgroup's run [s_run, s_stop) of the T stages; lane k holds the inclusive prefix of the
  // stage counts (every wave computes the same values: no LDS, no barrier)
  int pre = lane < K ? stages_of(koff, lane) : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(pre, d);
    if (lane >= d) pre += t;
  }
  const int T = __builtin_amdgcn_readfirstlane(__shfl(pre, 63));
  const int per = (T + W - 1) / W;
  int s_run = w * per;
  const int s_stop = (s_run + per < T) ? (s_run + per) : T;
  if (s_run >= s_stop) return;

  const u32x4 rs_a = make_rsrc(a, a_bytes), rs_b = make_rsrc(b, b_bytes);

  // what this lane fetches in each of its DMA instructions: tile row and byte offset inside the
  // gathered row (OOB: the segment lies past the row's channels)
  int row_a[IA], row_b[IB];
  unsigned col_a[IA], col_b[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int sg = 64 * (wave * IA + i) + lane, row = sg / SEG_A, c = ca0 + unswz<SEG_A>(row, sg % SEG_A) * 8;
    row_a[i] = row;
    col_a[i] = c < ca ? (unsigned)c * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int sg = 64 * (wave * IB + i) + lane, row = sg / SEG_B, c = cb0 + unswz<SEG_B>(row, sg % SEG_B) * 8;
    row_b[i] = row;
    col_b[i] = c < cb ? (unsigned)c * 2u : OOB;
  }
  const unsigned rb_a = (unsigned)ca * 2u, rb_b = (unsigned)cb * 2u;
  // the piece being worked on: rules [p_beg, p_beg + n_rules) of one offset
  int64_t p_beg = 0;
  int n_rules = 0;
  u32x4 rs_p = make_rsrc(nullptr, 0u);
  const int sel_a = a_col ? 4 : 0, sel_b = a_col ? 0 : 4;      // which half of a pair feeds a / b

  // ids of stage t (64 pairs = 512 B): waves 0 / 1 bring pairs 0..31 / 32..63, one dword per lane;
  // waves 2 / 3 repeat them (same bytes to the same place) so that every wave has the same number
  // of loads in flight -- the s_waitcnt immediates below count on it
  auto issue_ids = [&](int t) __attribute__((always_inline)) {
    if constexpr (!DENSE) {
      const int pi = t * RPS + 32 * (wave & 1) + (lane >> 1);
      const unsigned off = pi < n_rules ? (unsigned)pi * 8u + (unsigned)(lane & 1) * 4u : OOB;
      dma4(rs_p, ids0 + (t % IDS_R) * IDS_BYTES + (wave & 1) * 256, off);
    }
  };
  auto issue_rows = [&](int t) __attribute__((always_inline)) {
    const unsigned stage = lds0 + (unsigned)(t % R) * STAGE;
    const int r0 = t * RPS;
    const unsigned char* ids = smem + R * STAGE + (t % IDS_R) * IDS_BYTES;
    // all id reads first, as one batch (pinned: hipcc would otherwise sink each read under its
    // `ok` branch and wait for it alone)
    int ia[IA], ib[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      if constexpr (DENSE) ia[i] = (int)(p_beg + r0 + row_a[i]);
      else ia[i] = *reinterpret_cast<const int*>(ids + row_a[i] * 8 + sel_a);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if constexpr (DENSE) ib[i] = (int)(p_beg + r0 + row_b[i]);
      else ib[i] = *reinterpret_cast<const int*>(ids + row_b[i] * 8 + sel_b);
    }
#pragma unroll
    for (int i = 0; i < IA; ++i) asm volatile("" : "+v"(ia[i]));
#pragma unroll
    for (int i = 0; i < IB; ++i) asm volatile("" : "+v"(ib[i]));
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const bool ok = r0 + row_a[i] < n_rules && col_a[i] != OOB;
      const unsigned off = ok ? (unsigned)ia[i] * rb_a + col_a[i] : OOB;
      dma16(rs_a, stage + (wave * IA + i) * 1024, off);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const bool ok = r0 + row_b[i] < n_rules && col_b[i] != OOB;
      const unsigned off = ok ? (unsigned)ib[i] * rb_b + col_b[i] : OOB;
      dma16(rs_b, stage + A_BYTES + (wave * IB + i) * 1024, off);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: lane 4q+pp of each 16-lane group supplies the address of tile row
  // (8*gsel + q [+4] [+32]) at columns 4*pp..4*pp+3 of the 16-column block; lane i receives column i.
  // The segment permutation depends on row bits 0, 1, 3 only = (q, gsel): one offset per block.
  const int q = row16 >> 2, pp = row16 & 3;
  const int trow = 8 * gsel + q;
  int fa[MI], fb[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    fa[mi] = trow * (SEG_A * 16) + swz<SEG_A>(trow, 2 * (wr * MI + mi) + (pp >> 1)) * 16 + (pp & 1) * 8;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    fb[ni] = A_BYTES + trow * (SEG_B * 16) + swz<SEG_B>(trow, 2 * (wc * NI + ni) + (pp >> 1)) * 16 + (pp & 1) * 8;

  while (s_run < s_stop) {
    // the offset this stage belongs to, and how far the run stays inside it
    const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(lane < K && pre > s_run)));
    const int k_first = k ? __builtin_amdgcn_readlane(pre, k - 1) : 0;           // first stage of offset k
    const int k_last = __builtin_amdgcn_readlane(pre, k);
    const int s_end = k_last < s_stop ? k_last : s_stop;
    const int64_t beg = koff[k], end = koff[k + 1];
    p_beg = beg + (int64_t)(s_run - k_first) * RPS;
    const int64_t p_end = (beg + (int64_t)(s_end - k_first) * RPS < end) ? beg + (int64_t)(s_end - k_first) * RPS : end;
    n_rules = (int)(p_end - p_beg);
    const int nsteps = s_end - s_run;
    rs_p = make_rsrc(DENSE ? nullptr : (const void*)(pairs + p_beg), DENSE ? 0u : (unsigned)n_rules * 8u);

    // prologue: ids of the first D stages, then (ids(j+D), rows(j)) in the loop's own order
#pragma unroll
    for (int j = 0; j < D; ++j) issue_ids(j);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) {
      issue_ids(j + D);
      issue_rows(j);
    }
    for (int step = 0; step < nsteps; ++step) {
      // rows(step) and ids(step+D) have landed: only the D-1 younger stages may be in flight
      __builtin_amdgcn_s_waitcnt(0x0F70 | (INFLIGHT & 15) | ((INFLIGHT >> 4) << 14));
      __syncthreads();                 // ... for every wave's share; and stage step-1 is free again
      issue_ids(step + 2 * D);
      issue_rows(step + D);            // into the slot of stage step-1 (past the end: zeros)
      const unsigned char* st = smem + (step % R) * STAGE;
#pragma unroll
      for (int ks = 0; ks < RPS / 32; ++ks) {
        bf16x8 af[MI], bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const unsigned char* base = st + fa[mi] + ks * 32 * (SEG_A * 16);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(base));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(base + 4 * (SEG_A * 16)));
          af[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const unsigned char* base = st + fb[ni] + ks * 32 * (SEG_B * 16);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(base));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(base + 4 * (SEG_B * 16)));
          bf[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    // the zero-fill DMAs past the end have landed and every wave is done reading before the next
    // piece's prologue (or the end of the workgroup) re-uses the ring
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    float* dst = partial + (int64_t)(w + k) * ca * cb;              // slab w + k
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = ca0 + (wr * MI + mi) * 16 + gsel * 4 + r;
          int j = cb0 + (wc * NI + ni) * 16 + row16;
          if (i < ca && j < cb) dst[(int64_t)i * cb + j] = acc[mi][ni][r];
        }
        acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    s_run = s_end;
  }
}

// gw[k] = the slabs of offset k added in workgroup order (zeros for an offset without rules).
// (Deferring the reductions of all layers to ONE launch at the end of the backward pass was tried and
// dropped: autograd's AccumulateGrad clones a gradient tensor that anything else still references --
// before it is filled -- and keeping every layer's slabs alive until then made the step slower,
// 20.2 -> 21-22 ms.)
// Eight slabs are in flight per thread and added in slab order, so the sum does not depend on the
// unrolling (10.5 us per launch with four in flight and a serial prefix loop: the 53 launches of a
// step cost 0.55 ms, more than most layers' gradient itself).
template <int V>
__global__ void __launch_bounds__(256)
wgrad_dma_reduce_kernel(const float* __restrict__ partial, const int64_t* __restrict__ koff,
                        float* __restrict__ gw, int K, int64_t per_k, int W, int rps) {
  __shared__ int sh[2];
  const int k = blockIdx.y;
  if (threadIdx.x < 64) {          // the same prefix the gradient kernel computed, lane kk = offset kk
    const int lane = threadIdx.x;
    const int mine = lane < K ? (int)((koff[lane + 1] - koff[lane] + rps - 1) / rps) : 0;    // stages of offset `lane`
    int pre = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(pre, d);
      if (lane >= d) pre += t;
    }
    const int T = __shfl(pre, 63);
    const int per = (T + W - 1) / W;
    if (lane == k) {
      const int first = pre - mine;
      sh[0] = mine > 0 ? first / per : 0;
      sh[1] = mine > 0 ? (first + mine - 1) / per - first / per + 1 : 0;
    }
  }
  __syncthreads();
  const int w0 = sh[0], n = sh[1];
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= per_k) return;
  const float* src = partial + (int64_t)(w0 + k) * per_k + i;
  constexpr int U = 8;
  if constexpr (V == 4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + U <= n; j += U) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const float4*>(src + (int64_t)(j + u) * per_k);
#pragma unroll
      for (int u = 0; u < U; ++u) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
    }
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      x[u] = j + u < n ? *reinterpret_cast<const float4*>(src + (int64_t)(j + u) * per_k)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j + u < n) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
    *reinterpret_cast<float4*>(gw + (int64_t)k * per_k + i) = s;
  } else {
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += src[(int64_t)j * per_k];
    gw[(int64_t)k * per_k + i] = s;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// f32 weight gradient in the split form (round 6): gw = a^T b over f32 operands, every product as six bf16 MFMAs
// ------------------------------------------------------------------------------------------------------------------
// The f32 training step's weight gradients ran on v_mfma_f32_16x16x4_f32 (conv.hip conv_wgrad_kernel): 25 of the step's
// ~42 ms once the forward products and data gradients had moved to the split form (bench.py --dtype f32, families).  Here
// both operands are first cut into their three bf16 pieces by a streaming pass (split_rows_kernel: row r of [n, c] f32 ->
// row r of [n, 3 c] bf16 = hi(c) | mid(c) | lo(c); v = hi + mid + lo exactly, conv_img.hip cut3), then this kernel -- the
// bf16 kernel above with THREE sub-tiles per operand and stage -- gathers the pieces of a rule's two rows by LDS-DMA and
// accumulates  lo.hi + mid.mid + hi.lo + mid.hi + hi.mid + hi.hi  (smallest first) per 32-rule step: 6 MI NI MFMAs per
// wave and step.  Stages are 32 rules (one MFMA reduction step) with the sub-tile rows padded to a power-of-two number of
// 16-byte segments (the padding segments are fetched out of range: zeros, no traffic), so that every DMA instruction
// moves whole 1-KiB runs and the bank swizzles of the bf16 kernel apply unchanged.  Same decomposition (W equal runs of
// the stage sequence, slab w + k, reduction in workgroup order): bitwise reproducible.
constexpr int SRPS = 32;       // rules per stage of the split kernel

__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ src, int c, __bf16* __restrict__ dst,
                                                         int64_t n) {
  // one thread per 8 channels of a row: 32 bytes in, 3 x 16 bytes out
  const int cv = c / 8;
  const int64_t total = n * cv;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cv;
    const int u = (int)(i - r * cv);
    const float4 v0 = *reinterpret_cast<const float4*>(src + r * c + u * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(src + r * c + u * 8 + 4);
    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned bits = __float_as_uint(f[e]);
      const unsigned hb = bits & 0xFFFF0000u;
      const float rr = f[e] - __uint_as_float(hb);
      const unsigned mb = __float_as_uint(rr) & 0xFFFF0000u;
      const unsigned lb = __float_as_uint(rr - __uint_as_float(mb));
      h[e] = (unsigned short)(hb >> 16); m[e] = (unsigned short)(mb >> 16); l[e] = (unsigned short)(lb >> 16);
    }
    unsigned short* o = reinterpret_cast<unsigned short*>(dst) + r * 3 * c + u * 8;
    *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<const u32x4*>(h);
    *reinterpret_cast<u32x4*>(o + c) = *reinterpret_cast<const u32x4*>(m);
    *reinterpret_cast<u32x4*>(o + 2 * c) = *reinterpret_cast<const u32x4*>(l);
  }
}

__device__ __forceinline__ int stages_of32(const int64_t* koff, int k) {
  return (int)((koff[k + 1] - koff[k] + SRPS - 1) / SRPS);
}
constexpr int pseg_of(int seg) { return seg <= 8 ? 8 : 16; }      // segments per LDS row of a sub-tile (padded)

template <int MI, int NI, bool DENSE>
__global__ void __launch_bounds__(WT)
wgrad_split_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, unsigned a_bytes,
                   unsigned b_bytes, const int2* __restrict__ pairs, const int64_t* __restrict__ koff,
                   int a_col, float* __restrict__ partial, int K, int ca, int cb, int tiles_b) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  constexpr int SEG_A = TA / 8, SEG_B = TB / 8;
  constexpr int PA = pseg_of(SEG_A), PB = pseg_of(SEG_B);
  constexpr int SUB_A = SRPS * PA * 16, SUB_B = SRPS * PB * 16;     // bytes of one piece's sub-tile
  constexpr int A_BYTES = 3 * SUB_A, B_BYTES = 3 * SUB_B, STAGE = A_BYTES + B_BYTES;
  constexpr int R = 2;                                              // one stage travels while one is multiplied
  constexpr int IDS_R = 3, IDS_BYTES = SRPS * 8;
  constexpr int IA = PA / 8, IB = PB / 8;                           // DMA instructions per wave, piece and stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [R][STAGE] [IDS_R][IDS_BYTES]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned ids0 = lds0 + R * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row16 = lane & 15, gsel = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int w = blockIdx.x, W = gridDim.x;
  const int ta = blockIdx.y / tiles_b, tb = blockIdx.y - ta * tiles_b;
  const int ca0 = ta * TA, cb0 = tb * TB;

  int pre = lane < K ? stages_of32(koff, lane) : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(pre, d);
    if (lane >= d) pre += t;
  }
  const int T = __builtin_amdgcn_readfirstlane(__shfl(pre, 63));
  const int per = (T + W - 1) / W;
  int s_run = w * per;
  const int s_stop = (s_run + per < T) ? (s_run + per) : T;
  if (s_run >= s_stop) return;

  const u32x4 rs_a = make_rsrc(a, a_bytes), rs_b = make_rsrc(b, b_bytes);

  // what this lane fetches in each of its DMA instructions of a sub-tile: tile row and byte offset of the segment inside
  // the piece (OOB: a padding segment, or past the row's channels)
  int row_a[IA], row_b[IB];
  unsigned col_a[IA], col_b[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int sg = 64 * (wave * IA + i) + lane, row = sg / PA, ls = unswz<PA>(row, sg % PA), c = ca0 + ls * 8;
    row_a[i] = row;
    col_a[i] = (ls < SEG_A && c < ca) ? (unsigned)c * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int sg = 64 * (wave * IB + i) + lane, row = sg / PB, ls = unswz<PB>(row, sg % PB), c = cb0 + ls * 8;
    row_b[i] = row;
    col_b[i] = (ls < SEG_B && c < cb) ? (unsigned)c * 2u : OOB;
  }
  const unsigned rb_a = (unsigned)ca * 6u, rb_b = (unsigned)cb * 6u;       // bytes of a split row: 3 pieces of c bf16
  const unsigned pc_a = (unsigned)ca * 2u, pc_b = (unsigned)cb * 2u;       // bytes of one piece inside it
  int64_t p_beg = 0;
  int n_rules = 0;
  u32x4 rs_p = make_rsrc(nullptr, 0u);
  const int sel_a = a_col ? 4 : 0, sel_b = a_col ? 0 : 4;

  // ids of stage t (32 pairs = 256 B): one dword per lane, every wave brings the same bytes to the same place (every wave
  // then has the same number of loads in flight: the counted waits below rely on it)
  auto issue_ids = [&](int t) __attribute__((always_inline)) {
    if constexpr (!DENSE) {
      const int pi = t * SRPS + (lane >> 1);
      const unsigned off = pi < n_rules ? (unsigned)pi * 8u + (unsigned)(lane & 1) * 4u : OOB;
      dma4(rs_p, ids0 + (t % IDS_R) * IDS_BYTES, off);
    }
  };
  auto issue_rows = [&](int t) __attribute__((always_inline)) {
    const unsigned stage = lds0 + (unsigned)(t % R) * STAGE;
    const int r0 = t * SRPS;
    const unsigned char* ids = smem + R * STAGE + (t % IDS_R) * IDS_BYTES;
    int ia[IA], ib[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      if constexpr (DENSE) ia[i] = (int)(p_beg + r0 + row_a[i]);
      else ia[i] = *reinterpret_cast<const int*>(ids + row_a[i] * 8 + sel_a);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if constexpr (DENSE) ib[i] = (int)(p_beg + r0 + row_b[i]);
      else ib[i] = *reinterpret_cast<const int*>(ids + row_b[i] * 8 + sel_b);
    }
#pragma unroll
    for (int i = 0; i < IA; ++i) asm volatile("" : "+v"(ia[i]));
#pragma unroll
    for (int i = 0; i < IB; ++i) asm volatile("" : "+v"(ib[i]));
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) {
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = r0 + row_a[i] < n_rules && col_a[i] != OOB;
        const unsigned off = ok ? (unsigned)ia[i] * rb_a + (unsigned)pz * pc_a + col_a[i] : OOB;
        dma16(rs_a, stage + pz * SUB_A + (wave * IA + i) * 1024, off);
      }
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const bool ok = r0 + row_b[i] < n_rules && col_b[i] != OOB;
        const unsigned off = ok ? (unsigned)ib[i] * rb_b + (unsigned)pz * pc_b + col_b[i] : OOB;
        dma16(rs_b, stage + A_BYTES + pz * SUB_B + (wave * IB + i) * 1024, off);
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int q = row16 >> 2, pp = row16 & 3;
  const int trow = 8 * gsel + q;
  int fa[MI], fb[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    fa[mi] = trow * (PA * 16) + swz<PA>(trow, 2 * (wr * MI + mi) + (pp >> 1)) * 16 + (pp & 1) * 8;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    fb[ni] = A_BYTES + trow * (PB * 16) + swz<PB>(trow, 2 * (wc * NI + ni) + (pp >> 1)) * 16 + (pp & 1) * 8;

  auto frag = [&](const unsigned char* base, int pitch16) __attribute__((always_inline)) -> bf16x8 {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 4 * pitch16));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  while (s_run < s_stop) {
    const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(lane < K && pre > s_run)));
    const int k_first = k ? __builtin_amdgcn_readlane(pre, k - 1) : 0;
    const int k_last = __builtin_amdgcn_readlane(pre, k);
    const int s_end = k_last < s_stop ? k_last : s_stop;
    const int64_t beg = koff[k], end = koff[k + 1];
    p_beg = beg + (int64_t)(s_run - k_first) * SRPS;
    const int64_t p_end = (beg + (int64_t)(s_end - k_first) * SRPS < end) ? beg + (int64_t)(s_end - k_first) * SRPS : end;
    n_rules = (int)(p_end - p_beg);
    const int nsteps = s_end - s_run;
    rs_p = make_rsrc(DENSE ? nullptr : (const void*)(pairs + p_beg), DENSE ? 0u : (unsigned)n_rules * 8u);

    issue_ids(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
    issue_ids(1);
    issue_rows(0);
    for (int step = 0; step < nsteps; ++step) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                   // rows(step) and ids(step + 1) have landed (nothing younger is in flight)
      __syncthreads();                                      // ... for every wave's share; and stage step-1 is free again
      issue_ids(step + 2);
      issue_rows(step + 1);                                 // into the slot of stage step-1 (past the end: zeros)
      const unsigned char* st = smem + (step % R) * STAGE;
      bf16x8 af[3][MI], bf[3][NI];
#pragma unroll
      for (int pz = 0; pz < 3; ++pz) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[pz][mi] = frag(st + pz * SUB_A + fa[mi], PA * 16);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[pz][ni] = frag(st + pz * SUB_B + fb[ni], PB * 16);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {       // pieces: 0 = hi, 1 = mid, 2 = lo; smallest partial products first
          f32x4 c = acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][mi], bf[0][ni], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][mi], bf[1][ni], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mi], bf[2][ni], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][mi], bf[0][ni], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mi], bf[1][ni], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mi], bf[0][ni], c, 0, 0, 0);
          acc[mi][ni] = c;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    float* dst = partial + (int64_t)(w + k) * ca * cb;              // slab w + k
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = ca0 + (wr * MI + mi) * 16 + gsel * 4 + r;
          int j = cb0 + (wc * NI + ni) * 16 + row16;
          if (i < ca && j < cb) dst[(int64_t)i * cb + j] = acc[mi][ni][r];
        }
        acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    s_run = s_end;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The streamed form (round 6): the bf16 kernel above on rule lists re-ordered into ONE STREAM PER WORKGROUP
// ------------------------------------------------------------------------------------------------------------------
// The offset-major decomposition above reads both rows of every rule from the fabric (672 MB per launch on the 96 -> 96
// layer of level 0 against 168 MB of operands: counter record of round 5): the rules that touch a row sit in 27 different
// offsets and are worked on by workgroups far apart in time.  lidal_wgrad_streams_build (below) re-orders the rule lists
// of a stride-1 map so that
//   * the rows are cut into NB blocks of consecutive SPATIAL keys (the row index where rows are numbered in coordinate
//     order, the parent's row index on a level numbered by coordinate hash); block b belongs to XCD b % 8 -- workgroup
//     w runs on XCD w % 8 (round-robin dispatch) -- which walks its blocks in order: the rows of a block are fetched once
//     into that XCD's L2 and found there by the rules of all 27 offsets;
//   * inside an XCD the W / 8 workgroups divide the OFFSETS between them in proportion to the offsets' rule counts (a
//     workgroup keeps its accumulators for the whole launch; per block it works on its share of the block's rules of
//     its offset), a workgroup at the border between two offsets serving both with two accumulator sets;
//   * a workgroup's rules are stored as one padded run of 64-rule stages (every (block, set) piece padded to whole
//     32-rule MFMA steps with out-of-range rules: zeros, no memory access; bit 31 of a rule's first index = its set).
// The kernel is then the loop above without pieces: one uninterrupted pipeline per workgroup, two slabs at the end.
// stages in flight per workgroup: one, as in the offset-major kernel (measured with two resident workgroups per CU and as
// many stages as 78 KB of LDS hold -- depth 2 for 96 -> 96, 4 for 32 -> 32 -- on 397 k / 226 k rows: 96 -> 96 88 -> 98 us,
// 64 -> 64 52 -> 53, 32 -> 32 52 -> 51.5; LIDAL_X_STREAM_DEPTH keeps the deeper ring selectable for experiments)
constexpr int stream_depth(int /*stage_bytes*/) { return 1; }

template <int MI, int NI, int DD>
__global__ void __launch_bounds__(WT, 2)
wgrad_stream_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, unsigned n_a, unsigned n_b,
                    const int2* __restrict__ spairs, const int* __restrict__ soff, const int* __restrict__ wk,
                    int a_col, float* __restrict__ partial, int ca, int cb) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  constexpr int SEG_A = TA / 8, SEG_B = TB / 8;
  constexpr int A_BYTES = RPS * TA * 2, B_BYTES = RPS * TB * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int D = DD ? DD : stream_depth(STAGE), R = D + 1, IDS_R = 2 * D + 1, IDS_BYTES = RPS * 8;
  constexpr int IA = SEG_A / 4, IB = SEG_B / 4;
  constexpr int INFLIGHT = (D - 1) * (IA + IB + 1);
  static_assert(INFLIGHT < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [R][STAGE] [IDS_R][IDS_BYTES]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned ids0 = lds0 + R * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row16 = lane & 15, gsel = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int w = blockIdx.x;
  const int s_beg = soff[w], nsteps = soff[w + 1] - s_beg;
  const int k0 = wk[2 * w], k1 = wk[2 * w + 1];
  if (k0 < 0 && k1 < 0) return;
  const int n_rules = nsteps * RPS;

  const u32x4 rs_a = make_rsrc(a, n_a * (unsigned)ca * 2u), rs_b = make_rsrc(b, n_b * (unsigned)cb * 2u);
  const u32x4 rs_p = make_rsrc((const void*)(spairs + (int64_t)s_beg * RPS), (unsigned)n_rules * 8u);

  int row_a[IA], row_b[IB];
  unsigned col_a[IA], col_b[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int sg = 64 * (wave * IA + i) + lane, row = sg / SEG_A, c = unswz<SEG_A>(row, sg % SEG_A) * 8;
    row_a[i] = row;
    col_a[i] = c < ca ? (unsigned)c * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int sg = 64 * (wave * IB + i) + lane, row = sg / SEG_B, c = unswz<SEG_B>(row, sg % SEG_B) * 8;
    row_b[i] = row;
    col_b[i] = c < cb ? (unsigned)c * 2u : OOB;
  }
  const unsigned rb_a = (unsigned)ca * 2u, rb_b = (unsigned)cb * 2u;
  const int sel_a = a_col ? 4 : 0, sel_b = a_col ? 0 : 4;

  auto issue_ids = [&](int t) __attribute__((always_inline)) {
    const int pi = t * RPS + 32 * (wave & 1) + (lane >> 1);
    const unsigned off = pi < n_rules ? (unsigned)pi * 8u + (unsigned)(lane & 1) * 4u : OOB;
    dma4(rs_p, ids0 + (t % IDS_R) * IDS_BYTES + (wave & 1) * 256, off);
  };
  auto issue_rows = [&](int t) __attribute__((always_inline)) {
    const unsigned stage = lds0 + (unsigned)(t % R) * STAGE;
    const unsigned char* ids = smem + R * STAGE + (t % IDS_R) * IDS_BYTES;
    const bool live = t < nsteps;
    unsigned ia[IA], ib[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) ia[i] = *reinterpret_cast<const unsigned*>(ids + row_a[i] * 8 + sel_a);
#pragma unroll
    for (int i = 0; i < IB; ++i) ib[i] = *reinterpret_cast<const unsigned*>(ids + row_b[i] * 8 + sel_b);
#pragma unroll
    for (int i = 0; i < IA; ++i) asm volatile("" : "+v"(ia[i]));
#pragma unroll
    for (int i = 0; i < IB; ++i) asm volatile("" : "+v"(ib[i]));
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const unsigned id = ia[i] & 0x7FFFFFFFu;
      const bool ok = live && id < n_a && col_a[i] != OOB;
      dma16(rs_a, stage + (wave * IA + i) * 1024, ok ? id * rb_a + col_a[i] : OOB);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const unsigned id = ib[i] & 0x7FFFFFFFu;
      const bool ok = live && id < n_b && col_b[i] != OOB;
      dma16(rs_b, stage + A_BYTES + (wave * IB + i) * 1024, ok ? id * rb_b + col_b[i] : OOB);
    }
  };

  f32x4 acc0[MI][NI], acc1[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc0[mi][ni] = acc1[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int q = row16 >> 2, pp = row16 & 3;
  const int trow = 8 * gsel + q;
  int fa[MI], fb[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    fa[mi] = trow * (SEG_A * 16) + swz<SEG_A>(trow, 2 * (wr * MI + mi) + (pp >> 1)) * 16 + (pp & 1) * 8;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    fb[ni] = A_BYTES + trow * (SEG_B * 16) + swz<SEG_B>(trow, 2 * (wc * NI + ni) + (pp >> 1)) * 16 + (pp & 1) * 8;

#pragma unroll
  for (int j = 0; j < D; ++j) issue_ids(j);
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue_ids(j + D);
    issue_rows(j);
  }
  for (int step = 0; step < nsteps; ++step) {
    // rows(step) and ids(step + D) have landed: only the D - 1 younger stages may be in flight
    __builtin_amdgcn_s_waitcnt(0x0F70 | (INFLIGHT & 15) | ((INFLIGHT >> 4) << 14));
    __syncthreads();
    issue_ids(step + 2 * D);
    issue_rows(step + D);
    const unsigned char* st = smem + (step % R) * STAGE;
    const unsigned char* ids = smem + R * STAGE + (step % IDS_R) * IDS_BYTES;
#pragma unroll
    for (int ks = 0; ks < RPS / 32; ++ks) {
      // the set of this 32-rule step: bit 31 of its first rule's first index (pieces are padded to whole steps)
      const unsigned flag = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(ids + ks * 256)) >> 31;
      bf16x8 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned char* base = st + fa[mi] + ks * 32 * (SEG_A * 16);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 4 * (SEG_A * 16)));
        af[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const unsigned char* base = st + fb[ni] + ks * 32 * (SEG_B * 16);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 4 * (SEG_B * 16)));
        bf[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      if (flag == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc0[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf[ni], acc0[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc1[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf[ni], acc1[mi][ni], 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int64_t per_k = (int64_t)ca * cb;
#pragma unroll
  for (int set = 0; set < 2; ++set) {
    if ((set ? k1 : k0) < 0) continue;
    float* dst = partial + (int64_t)(2 * w + set) * per_k;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = (wr * MI + mi) * 16 + gsel * 4 + r;
          const int j = (wc * NI + ni) * 16 + row16;
          if (i < ca && j < cb) dst[(int64_t)i * cb + j] = set ? acc1[mi][ni][r] : acc0[mi][ni][r];
        }
  }
}

// gw[k] = the slabs of the workgroups that served offset k, added in a fixed order: kred[k][x] = (first slot, slots, set of
// the first slot) on XCD x; slot j there = workgroup 8 j + x, slab 2 w + set (set 0 on every slot but possibly the first)
__global__ void __launch_bounds__(256)
wgrad_stream_reduce_kernel(const float* __restrict__ partial, const int* __restrict__ kred, float* __restrict__ gw,
                           int64_t per_k) {
  __shared__ int slabs[520];
  __shared__ int n_sh;
  const int k = blockIdx.y;
  if (threadIdx.x < 64) {           // lane x < 8 lists the slabs of XCD x behind those of the XCDs before it
    const int x = threadIdx.x;
    const int j0 = x < 8 ? kred[(k * 8 + x) * 3] : 0, nj = x < 8 ? kred[(k * 8 + x) * 3 + 1] : 0;
    const int set0 = x < 8 ? kred[(k * 8 + x) * 3 + 2] : 0;
    int pre = nj;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const int t = __shfl_up(pre, d);
      if (x >= d) pre += t;
    }
    if (x < 8)
      for (int jj = 0; jj < nj && pre - nj + jj < 512; ++jj) slabs[pre - nj + jj] = 2 * (8 * (j0 + jj) + x) + (jj == 0 ? set0 : 0);
    if (x == 7) n_sh = pre < 512 ? pre : 512;
  }
  __syncthreads();
  // 32 float4 columns x 8 groups: group g adds the slabs g, g + 8, .. of the list in order, the groups' sums are added in
  // group order -- a fixed association, 8 x 8 slabs in flight per column (the centre offset's list is ~110 slabs long)
  const int n = n_sh;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = ((int64_t)blockIdx.x * 32 + col) * 4;
  const bool live = i < per_k;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 8;
  if (live)
    for (int t = grp; t < n; t += 8 * U) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        x[u] = t + 8 * u < n ? *reinterpret_cast<const float4*>(partial + (int64_t)slabs[t + 8 * u] * per_k + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (t + 8 * u < n) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
    }
  __shared__ float4 gsum[8][32];
  gsum[grp][col] = s;
  __syncthreads();
  if (grp == 0 && live) {
    float4 r = gsum[0][col];
#pragma unroll
    for (int g = 1; g < 8; ++g) { const float4 v = gsum[g][col]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    *reinterpret_cast<float4*>(gw + (int64_t)k * per_k + i) = r;
  }
}

constexpr int lds_bytes(int ta, int tb) {
  const int stage = RPS * (ta + tb) * 2, d = wgrad_depth(stage);
  return (d + 1) * stage + (2 * d + 1) * RPS * 8;
}

template <int MI, int NI>
int launch(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* pairs,
           const int64_t* koff, int a_col, float* gw, float* partial, int W, int K, int ca, int cb,
           hipStream_t s) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  const size_t lds = (size_t)lds_bytes(TA, TB);
  const int tiles_a = (int)cdiv(ca, TA), tiles_b = (int)cdiv(cb, TB);
  dim3 grid((unsigned)W, (unsigned)(tiles_a * tiles_b));
  const bool dense = pairs == nullptr;
  auto kern = dense ? wgrad_dma_kernel<MI, NI, true> : wgrad_dma_kernel<MI, NI, false>;
  static size_t attr_set[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[dense][dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dense][dev] = lds;
  }
  kern<<<grid, WT, lds, s>>>((const __bf16*)a, (const __bf16*)b, (unsigned)(n_a * ca * 2),
                             (unsigned)(n_b * cb * 2), (const int2*)pairs, koff, a_col, partial, K,
                             ca, cb, tiles_b);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad (dma)");
  const int64_t per_k = (int64_t)ca * cb;
  if (per_k % 4 == 0)
    wgrad_dma_reduce_kernel<4><<<dim3((unsigned)cdiv(per_k / 4, 256), (unsigned)K), 256, 0, s>>>(
        partial, koff, gw, K, per_k, W, RPS);
  else
    wgrad_dma_reduce_kernel<1><<<dim3((unsigned)cdiv(per_k, 256), (unsigned)K), 256, 0, s>>>(
        partial, koff, gw, K, per_k, W, RPS);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad (dma reduce)");
  return 0;
}

constexpr int split_lds_bytes(int ta, int tb) {
  return 2 * 3 * SRPS * 16 * (pseg_of(ta / 8) + pseg_of(tb / 8)) + 3 * SRPS * 8;
}

template <int MI, int NI>
int launch_split(const void* a3, const void* b3, int64_t n_a, int64_t n_b, const int* pairs, const int64_t* koff, int a_col,
                 float* gw, float* partial, int W, int K, int ca, int cb, hipStream_t s) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  const size_t lds = (size_t)split_lds_bytes(TA, TB);
  const int tiles_a = (int)cdiv(ca, TA), tiles_b = (int)cdiv(cb, TB);
  dim3 grid((unsigned)W, (unsigned)(tiles_a * tiles_b));
  const bool dense = pairs == nullptr;
  auto kern = dense ? wgrad_split_kernel<MI, NI, true> : wgrad_split_kernel<MI, NI, false>;
  static size_t attr_set[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[dense][dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dense][dev] = lds;
  }
  kern<<<grid, WT, lds, s>>>((const __bf16*)a3, (const __bf16*)b3, (unsigned)(n_a * ca * 6), (unsigned)(n_b * cb * 6),
                             (const int2*)pairs, koff, a_col, partial, K, ca, cb, tiles_b);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad (split)");
  const int64_t per_k = (int64_t)ca * cb;
  if (per_k % 4 == 0)
    wgrad_dma_reduce_kernel<4><<<dim3((unsigned)cdiv(per_k / 4, 256), (unsigned)K), 256, 0, s>>>(partial, koff, gw, K, per_k, W, SRPS);
  else
    wgrad_dma_reduce_kernel<1><<<dim3((unsigned)cdiv(per_k, 256), (unsigned)K), 256, 0, s>>>(partial, koff, gw, K, per_k, W, SRPS);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad (split reduce)");
  return 0;
}

template <int MI, int NI>
int launch_stream(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* spairs, const int* sdesc, int W, int a_col,
                  float* gw, float* partial, int K, int ca, int cb, hipStream_t s) {
  constexpr int TA = 2 * MI * 16, TB = 2 * NI * 16;
  static const int depth_x = getenv("LIDAL_X_STREAM_DEPTH") ? atoi(getenv("LIDAL_X_STREAM_DEPTH")) : 0;     // (experiments: 2)
  const int D = depth_x == 2 && 3 * RPS * (TA + TB) * 2 <= 78 * 1024 ? 2 : stream_depth(RPS * (TA + TB) * 2);
  const size_t lds = (size_t)((D + 1) * RPS * (TA + TB) * 2 + (2 * D + 1) * RPS * 8);
  auto kern = D == 2 ? wgrad_stream_kernel<MI, NI, 2> : wgrad_stream_kernel<MI, NI, 0>;
  static size_t attr_set[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[D == 2][dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[D == 2][dev] = lds;
  }
  const int* soff = sdesc + STREAM_HDR;
  const int* wk = soff + (W + 1);
  const int* kred = wk + 2 * W;
  kern<<<dim3((unsigned)W), WT, lds, s>>>((const __bf16*)a, (const __bf16*)b, (unsigned)n_a, (unsigned)n_b, (const int2*)spairs,
                                          soff, wk, a_col, partial, ca, cb);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad_streams");
  const int64_t per_k = (int64_t)ca * cb;
  wgrad_stream_reduce_kernel<<<dim3((unsigned)cdiv(per_k / 4, 32), (unsigned)K), 256, 0, s>>>(partial, kred, gw, per_k);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad_streams (reduce)");
  return 0;
}

}  // namespace

namespace lidal {

bool wgrad_dma_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  if (ca % 8 != 0 || cb % 8 != 0 || k > 64) return false;      // 16-byte segments of whole rows
  return n_a * ca * 2 < (int64_t)OOB && n_b * cb * 2 < (int64_t)OOB;      // 32-bit byte offsets
}

// Workgroups of the launch (scripts/ablate_wgrad.py, 396k..16k rows): ONE evenly loaded round --
//   * two resident workgroups per CU, or one when the gathered operands exceed what the eight L2s
//     hold together and a stage is >= 16 KB: there the launch is bound by cache lines fetched, not by
//     latency (96->96 on 396k rows: 124 us with 256 workgroups, 137 with 512, 141 with 768; depth 2
//     of the ring was slower for the same reason), while the small levels want the parallelism
//     (256->256 on 43k rows: 68 us with 512, 83 with 256);
//   * shared between the channel tiles, and a multiple of 256 in total when there are that many
//     (392 workgroups for 256 CUs measured 49.6 us where 256 took 44.2 and 512 41.0);
//   * no more than leaves each run ~8 stages of the ESTIMATED rule count (the true one is only
//     known on the device; ~6 rules per row for a 3x3x3 map on LiDAR surfaces, one per row for the
//     2x2x2 maps and dense layers).
constexpr int WGRAD_RESIDENT = 0;          // 0 = the rule above
constexpr int WGRAD_MIN_STAGES = 8;
int wgrad_dma_workgroups(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  const int ta = wgrad_blocks(ca) * 32, tb = wgrad_blocks(cb) * 32;
  const int64_t tiles = cdiv(ca, ta) * cdiv(cb, tb);
  const int64_t n_rows = n_a > n_b ? n_a : n_b;
  int64_t resident = WGRAD_RESIDENT;
  if (resident == 0)
    resident = (RPS * (ta + tb) * 2 >= 16384 && (n_a * ca + n_b * cb) * 2 > (48ll << 20)) ? 1 : 2;
  const int64_t fit = (160 * 1024) / lds_bytes(ta, tb);
  if (resident > fit) resident = fit < 1 ? 1 : fit;
  int64_t w = 256 * resident / tiles;
  const int64_t stages = (k > 8 ? 6 : 1) * n_rows / RPS;
  if (w > stages / WGRAD_MIN_STAGES) w = stages / WGRAD_MIN_STAGES;
  if (w * tiles >= 256) w = (w * tiles / 256) * 256 / tiles;
  return (int)(w < 1 ? 1 : w);
}

// ---- the split form: f32 operands, pieces cut into `scratch` (bf16 [n_a, 3 ca] | [n_b, 3 cb]) first
bool wgrad_split_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  if (ca % 8 != 0 || cb % 8 != 0 || k > 64) return false;
  return n_a * ca * 6 < (int64_t)OOB && n_b * cb * 6 < (int64_t)OOB;
}
int64_t wgrad_split_scratch_bytes(int64_t n_a, int64_t n_b, int ca, int cb) {
  return align_up(n_a * ca * 6, 256) + align_up(n_b * cb * 6, 256);
}
// workgroups: one resident round, shared between the channel tiles, runs of >= 8 stages of the estimated rule count
int wgrad_split_workgroups(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  const int ta = wgrad_blocks(ca) * 32, tb = wgrad_blocks(cb) * 32;
  const int64_t tiles = cdiv(ca, ta) * cdiv(cb, tb);
  const int64_t n_rows = n_a > n_b ? n_a : n_b;
  int64_t resident = (160 * 1024) / split_lds_bytes(ta, tb);
  resident = resident < 1 ? 1 : (resident > 2 ? 2 : resident);
  int64_t w = 256 * resident / tiles;
  const int64_t stages = (k > 8 ? 6 : 1) * n_rows / SRPS;
  if (w > stages / 16) w = stages / 16;
  if (w * tiles >= 256) w = (w * tiles / 256) * 256 / tiles;
  return (int)(w < 1 ? 1 : w);
}
int wgrad_split(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* pairs, const int64_t* koff, int a_col,
                float* gw, float* partial, void* scratch, int W, int K, int ca, int cb, hipStream_t s) {
  __bf16* a3 = (__bf16*)scratch;
  __bf16* b3 = (__bf16*)((char*)scratch + align_up(n_a * ca * 6, 256));
  auto grid_of = [](int64_t items) { int64_t g = cdiv(items, 256); return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g)); };
  if (n_a > 0) split_rows_kernel<<<grid_of(n_a * (ca / 8)), 256, 0, s>>>((const float*)a, ca, a3, n_a);
  if (n_b > 0) split_rows_kernel<<<grid_of(n_b * (cb / 8)), 256, 0, s>>>((const float*)b, cb, b3, n_b);
  LIDAL_CHECK_LAUNCH("lidal_conv_wgrad (split rows)");
  const int mi = wgrad_blocks(ca), ni = wgrad_blocks(cb);
#define WS_CASE(M, N) \
  if (mi == M && ni == N) return launch_split<M, N>(a3, b3, n_a, n_b, pairs, koff, a_col, gw, partial, W, K, ca, cb, s);
  WS_CASE(1, 1) WS_CASE(1, 2) WS_CASE(1, 3) WS_CASE(1, 4)
  WS_CASE(2, 1) WS_CASE(2, 2) WS_CASE(2, 3) WS_CASE(2, 4)
  WS_CASE(3, 1) WS_CASE(3, 2) WS_CASE(3, 3) WS_CASE(3, 4)
  WS_CASE(4, 1) WS_CASE(4, 2) WS_CASE(4, 3) WS_CASE(4, 4)
#undef WS_CASE
  set_error("wgrad(split): no tile for %d x %d", ca, cb);
  return 2;
}

int wgrad_dma(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* pairs,
              const int64_t* koff, int a_col, float* gw, float* partial, int W, int K, int ca,
              int cb, hipStream_t s) {
  const int mi = wgrad_blocks(ca), ni = wgrad_blocks(cb);
#define WG_CASE(M, N) \
  if (mi == M && ni == N) return launch<M, N>(a, b, n_a, n_b, pairs, koff, a_col, gw, partial, W, K, ca, cb, s);
  WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
  WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
  WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4)
  WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 3) WG_CASE(4, 4)
#undef WG_CASE
  set_error("wgrad: no tile for %d x %d", ca, cb);
  return 2;
}

bool wgrad_stream_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb) {
  if (!wgrad_dma_serves(n_a, n_b, k, ca, cb)) return false;
  const int ta = wgrad_blocks(ca) * 32, tb = wgrad_blocks(cb) * 32;
  return ca <= ta && cb <= tb && n_a < (1ll << 31) && n_b < (1ll << 31);          // one channel tile
}
int wgrad_stream(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* spairs, const int* sdesc, int W, int a_col,
                 float* gw, float* partial, int K, int ca, int cb, hipStream_t s) {
  const int mi = wgrad_blocks(ca), ni = wgrad_blocks(cb);
#define WT_CASE(M, N) \
  if (mi == M && ni == N) return launch_stream<M, N>(a, b, n_a, n_b, spairs, sdesc, W, a_col, gw, partial, K, ca, cb, s);
  WT_CASE(1, 1) WT_CASE(1, 2) WT_CASE(1, 3) WT_CASE(1, 4)
  WT_CASE(2, 1) WT_CASE(2, 2) WT_CASE(2, 3) WT_CASE(2, 4)
  WT_CASE(3, 1) WT_CASE(3, 2) WT_CASE(3, 3) WT_CASE(3, 4)
  WT_CASE(4, 1) WT_CASE(4, 2) WT_CASE(4, 3) WT_CASE(4, 4)
#undef WT_CASE
  set_error("wgrad(streams): no tile for %d x %d", ca, cb);
  return 2;
}

}  // namespace lidal
