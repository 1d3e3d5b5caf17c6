// TEST INFRASTRUCTURE (not part of liblidal_amd.so): the FIRST GENERATION of the fused sparse-convolution kernel
// (rounds 1-3: lidal_conv_apply, weights pre-packed as [k][co][ci] and staged through registers), kept as an
// independent implementation that tests/test_ops_gpu.py compares the shipped LDS-image kernels
// (lidal_amd/csrc/conv_img.hip: lidal_conv_apply_image) against, bit for bit.  Built into
// tests/native/liblidal_gen1.so by tests/native/build.py.
//
// Sparse 3D convolution for gfx950: fused gather -> MFMA GEMM -> accumulate, output-stationary.
//
// Dataflow (lidal_conv_apply).  One workgroup (4 waves) owns BM = 128 consecutive OUTPUT rows and
// a BN-wide slice of the output channels; each wave owns 32 of those rows (two 16-row MFMA groups)
// and keeps their f32 accumulators in registers while the workgroup walks the K kernel offsets:
//
//   prologue       the wave's slice of the neighbour table nbr[k][rows] (input row or -1) is read
//                  once, coalesced, into LDS;
//   per offset k   1. W[k] (reduction dim contiguous, pre-packed) is staged into LDS once for the
//                     whole workgroup, double-buffered: global->registers before the MFMAs of the
//                     current offset, registers->LDS after them, ONE barrier per offset;
//                  2. A-fragments are gathered STRAIGHT from HBM/L2 into registers (16 B per lane,
//                     64 B contiguous per gathered row, zero for rows without a rule; no LDS round
//                     trip: every A element is used by exactly one wave), one offset ahead of use;
//                  3. B-fragments are ds_read_b128 from the staged weights and feed both row
//                     groups: v_mfma_f32_16x16x32_bf16 or v_mfma_f32_16x16x4_f32 (exact f32);
//                     a wave whose 32 rows have no rule at this offset skips its MFMAs;
//   epilogue       accumulators -> wave-private LDS tile -> whole output rows, 16-byte stores.
//
// No atomics and no LDS accumulation: each output row is produced by one wave in a fixed offset
// order, so results are bitwise reproducible and every output byte is written exactly once.
// LDS holds only the two weight slabs + the index slices (< 64 KB), so 2-3 workgroups share a CU.
//
// The same kernel serves forward, data-gradient and transposed convolution: only the neighbour
// table and the weight layout differ (see lidal_amd/nn/functional/conv.py).
//
// lidal_conv_wgrad: gw[k] = A_k^T B_k over the rule list of offset k, as a split-K MFMA GEMM
// (workgroup = (split, k, channel tile)); gathered rows are staged through LDS (converted to
// f32), f32 partial slabs are reduced in a fixed order by a second kernel.
#include <type_traits>

#include "../../lidal_amd/csrc/common.h"

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
// conv_apply
// ------------------------------------------------------------------------------------------
constexpr int MAXK = 32;         // kernel volume limit (27 and 8 on this path)

constexpr int CONV_MINWAVES = 2;       // 3 for nb <= 6 measured no better

// LDS layout (dynamic): weights T [2][BN][WSTRIDE] | nidx int [4 waves][K][G*16]
// (the weight region is re-used as the epilogue staging tile)
// GUARD = the channel count is not a multiple of the 16-byte vector (only the 4-channel bf16 stem):
// loads then fall back to element-wise guarded code.  Everywhere else every lane load is either
// wholly inside the row or wholly masked, which keeps the gathers branch-free and un-serialised
// (the guarded form made hipcc wait vmcnt(0) after every load).
template <typename T, int NB, int ROW_BYTES, bool GUARD, int G, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES, CONV_MINWAVES)
conv_apply_kernel(const T* __restrict__ in, const T* __restrict__ wk, const int* __restrict__ nbr,
                  const int* __restrict__ perm, const unsigned* __restrict__ tmasks,
                  T* __restrict__ out, int64_t n_out, int ci, int co, int K, int kflip,
                  const float* __restrict__ ep_scale, const float* __restrict__ ep_shift,
                  int ep_relu, const T* __restrict__ ep_res, unsigned in_bytes, unsigned wk_bytes) {
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int BM = NWAVES * G * 16;                   // output rows per workgroup
  constexpr int BN = 16 * NB;
  constexpr int VEC = DT<T>::VEC;
  constexpr int CH = DT<T>::CH;
  constexpr int KC = ROW_BYTES / (int)sizeof(T);        // staged reduction elements per pass
  constexpr int MAXCC = KC / CH;                         // A lane-loads per pass and row group
  constexpr int WSTRIDE = KC + VEC;                      // +16 B pad
  constexpr int SEGS = KC / VEC;                         // 16-byte segments per staged weight row
  constexpr int WPT = (BN * SEGS + NTHREADS - 1) / NTHREADS;   // staged segments per thread
  constexpr int RW = G * 16;                             // rows per wave
  constexpr int ESTRIDE = BN + VEC;                      // epilogue tile row stride (elements)
  constexpr int WREGION = (2 * BN * WSTRIDE > NWAVES * RW * ESTRIDE) ? 2 * BN * WSTRIDE
                                                                       : NWAVES * RW * ESTRIDE;
  typedef typename DT<T>::frag frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  int* nidx_all = reinterpret_cast<int*>(smem + sizeof(T) * WREGION);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int row16 = lane & 15;
  const int gsel = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * BM + wave * RW;   // first row of this wave
  const int n0 = blockIdx.y * BN;
  const int npass = (ci + KC - 1) / KC;
  int* nidx = nidx_all + wave * K * RW;                      // [K][RW], wave-private
  __shared__ unsigned tile_mask;
  __shared__ int act_k[MAXK];
  unsigned tmask;
  if (nbr == nullptr) {
    // ---- 0. no table at all: the identity rule list of a dense per-row product (K == 1)
    tmask = 1u;
    if (tid == 0) act_k[0] = 0;
    for (int r = lane; r < RW; r += 64) nidx[r] = (r0 + r < n_out) ? (int)(r0 + r) : -1;
  } else if (tmasks != nullptr) {
    // ---- 0a. occupancy mask of this 128-row tile, precomputed with the row order (one load);
    //          only the offsets it names are fetched, staged and multiplied
    unsigned m = tmasks[((int64_t)blockIdx.x * BM) >> 7];     // masks are per 128 sorted rows
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = m;
    if (tid < K && (m >> tid) & 1u) act_k[__popc(m & ((1u << tid) - 1u))] = tid;
    __syncthreads();
    const int n_act0 = __popc(m);
    for (int i = lane; i < n_act0 * RW; i += 64) {
      const int q = i / RW, r = i - q * RW;
      const int k = act_k[q];
      const int kk = kflip ? (K - 1 - k) : k;
      nidx[k * RW + r] = (r0 + r < n_out) ? nbr[(int64_t)kk * n_out + r0 + r] : -1;
    }
  } else {
    // ---- 0b. no precomputed mask: fetch every offset's slice and derive the mask here
    if (tid == 0) tile_mask = 0u;
    for (int i = lane; i < K * RW; i += 64) {
      const int k = i / RW, r = i - k * RW;
      const int kk = kflip ? (K - 1 - k) : k;
      nidx[i] = (r0 + r < n_out) ? nbr[(int64_t)kk * n_out + r0 + r] : -1;
    }
    __syncthreads();
    unsigned wmask = 0u;
    for (int k = 0; k < K; ++k) {
      const int v = (lane < RW) ? nidx[k * RW + lane] : -1;
      if (__ballot(v >= 0) != 0ull) wmask |= 1u << k;
    }
    if (lane == 0 && wmask) atomicOr(&tile_mask, wmask);
    __syncthreads();
    tmask = tile_mask;
    if (tid < K && (tmask >> tid) & 1u) act_k[__popc(tmask & ((1u << tid) - 1u))] = tid;
  }
  tmask = __builtin_amdgcn_readfirstlane(tmask);      // wave-uniform: the walk below is scalar
  const int n_act = __popc(tmask);
  const int nphase = n_act * npass;

  // The phases walk the tile's active offsets in ascending k (the set bits of tmask), each offset
  // in `npass` reduction slices.  The walk lives in scalar registers -- an LDS look-up of the
  // offset list followed by a dependent LDS read of the neighbour indices would put two LDS round
  // trips in front of every phase's loads.
  struct Walk { unsigned rem; int k; int pass; };
  auto walk_begin = [&]() {
    Walk w;
    w.k = tmask ? __builtin_ctz(tmask) : 0;
    w.rem = tmask & (tmask - 1u);
    w.pass = 0;
    return w;
  };
  auto walk_next = [&](Walk& w) {
    if (++w.pass == npass) {
      w.pass = 0;
      w.k = w.rem ? __builtin_ctz(w.rem) : 0;      // past the end: any valid slot (result unused)
      w.rem &= w.rem - 1u;
    }
  };

  // Both operand streams are addressed through buffer descriptors: a 32-bit byte offset per lane
  // (a handful of VALU ops per load instead of 64-bit pointer arithmetic -- address generation was
  // the longest segment of a phase, profiles/README.md) and a hardware range check that returns
  // zeros for any offset >= the buffer size, which is how absent rules (offset OOB_OFF) and
  // channels past the row end are served without branches or loads.
  constexpr unsigned OOB_OFF = 0x80000000u;      // >= any buffer size accepted by the launcher
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wk =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wk), 0, (int)wk_bytes, 0x00020000);

  // weight slab of offset k, slice c0 -> registers (issue early) -> LDS buffer (write late)
  // (`live` false: the same loads, all out of range -- see the phase loop)
  frag wreg[WPT];
  unsigned woff[WPT];           // per-thread byte offset of its segments inside a slab (or OOB_OFF)
  int wx[WPT];                  // first reduction element of the segment
#pragma unroll
  for (int t = 0; t < WPT; ++t) {
    const int sidx = tid + t * NTHREADS;
    const int col = sidx / SEGS, x = (sidx - col * SEGS) * VEC;
    wx[t] = x;
    woff[t] = (sidx < BN * SEGS && n0 + col < co)
                  ? (unsigned)((col * ci + x) * (int)sizeof(T)) : OOB_OFF;
  }
  // `live` false turns every load of a phase into an out-of-range one by OR-ing the top offset bit;
  // the value is laundered through an empty asm so that hipcc cannot turn the uniform flag back
  // into a branch around the loads (which is what makes its wait counts pessimistic, see below)
  auto kill_bit = [&](bool live) {
    unsigned kb = live ? 0u : OOB_OFF;
    asm volatile("" : "+s"(kb));
    return kb;
  };
  auto stage_load = [&](int k, int c0, bool live) {
    const int kc = min(KC, ci - c0);
    const unsigned kill = kill_bit(live);
    // uniform part of the address: slab k, first column n0, slice c0
    const unsigned sbase = (unsigned)(((k * co + n0) * ci + c0) * (int)sizeof(T));
#pragma unroll
    for (int t = 0; t < WPT; ++t) {
      if constexpr (GUARD) {
        const int sidx = tid + t * NTHREADS;
        const int col = sidx / SEGS;
        const bool ok = live && woff[t] != OOB_OFF && wx[t] < kc;
        const T* wsrc = wk + ((int64_t)k * co + n0) * ci + c0;
        wreg[t] = DT<T>::zero();
        if (ok) wreg[t] = load_frag_guarded<T>(wsrc + (int64_t)col * ci + wx[t], kc - wx[t]);
      } else {
        const unsigned off = ((wx[t] < kc) ? woff[t] : OOB_OFF) | kill;
        wreg[t] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs_wk, off, sbase, 0));
      }
    }
  };
  auto stage_store = [&](int buf) {
    T* dstw = wl + buf * BN * WSTRIDE;
#pragma unroll
    for (int t = 0; t < WPT; ++t) {
      const int sidx = tid + t * NTHREADS;
      const int col = sidx / SEGS, x = (sidx - col * SEGS) * VEC;
      if (sidx < BN * SEGS) *reinterpret_cast<frag*>(dstw + col * WSTRIDE + x) = wreg[t];
    }
  };
  // neighbour indices of this lane's row in each row group for offset k (wave-private LDS slice)
  auto read_idx = [&](int (&src)[G], int k) {
#pragma unroll
    for (int g = 0; g < G; ++g) src[g] = nidx[k * RW + g * 16 + row16];
  };
  // A fragments of one phase: for each row group 16 gathered input rows x kc channels, straight to
  // VGPRs (16 B per lane, 64 B contiguous per row); `present` = ballot of rows that have a rule
  auto load_a = [&](raw4 (&a)[G][MAXCC], unsigned long long (&present)[G], const int (&idx)[G],
                    int c0, bool live) {
    const int kc = min(KC, ci - c0);
    const unsigned row_bytes = (unsigned)(ci * (int)sizeof(T));
    const unsigned lane_off = (unsigned)((c0 + gsel * VEC) * (int)sizeof(T));
    const unsigned kill = kill_bit(live);
    const unsigned long long live_mask = kill ? 0ull : ~0ull;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int src = GUARD ? (live ? idx[g] : -1) : idx[g];
      present[g] = __ballot(src >= 0) & live_mask;
      if constexpr (GUARD) {
#pragma unroll
        for (int cc = 0; cc < MAXCC; ++cc) {
          const int x = cc * CH + gsel * VEC;
          const bool ok = src >= 0 && x < kc;
          frag f = DT<T>::zero();
          if (ok) f = load_frag_guarded<T>(in + (int64_t)src * ci + c0 + x, kc - x);
          a[g][cc] = __builtin_bit_cast(raw4, f);
        }
      } else {
        const unsigned base = ((src >= 0) ? (unsigned)src * row_bytes + lane_off
                                                                  : OOB_OFF) | kill;
#pragma unroll
        for (int cc = 0; cc < MAXCC; ++cc) {
          // a chunk past the row end (ci not a multiple of the pass width) is sent out of range
          const unsigned off = (cc * CH + gsel * VEC < kc) ? base + (unsigned)(cc * CH * (int)sizeof(T))
                                                           : OOB_OFF;       // (base | kill) + 128 stays OOB
          a[g][cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
        }
      }
    }
  };

  f32x4 acc[G][NB];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[g][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();            // act_k visible
  // Two register sets of A fragments used alternately (no copies): the gathers of phase p+1 are
  // issued before the MFMAs of phase p and are first waited for by the MFMAs of phase p+1, i.e. a
  // full phase later, with the weight-slab store and the barrier in between (a register copy at
  // the end of the phase would make the wave wait for its gathers right there).
  raw4 a0[G][MAXCC], a1[G][MAXCC];
  unsigned long long pres0[G], pres1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) pres0[g] = pres1[g] = 0ull;
  Walk w1 = walk_begin();       // the phase whose loads are issued next
  int idx_nxt[G];               // its neighbour indices, read from LDS one phase ahead
#pragma unroll
  for (int g = 0; g < G; ++g) idx_nxt[g] = -1;
  if (nphase > 0) {
    stage_load(w1.k, 0, true);
    stage_store(0);
    read_idx(idx_nxt, w1.k);
    load_a(a0, pres0, idx_nxt, 0, true);
    walk_next(w1);
    read_idx(idx_nxt, w1.k);
  }
  __syncthreads();            // slab 0 visible

  auto phase = [&](int p, raw4 (&a_cur)[G][MAXCC], unsigned long long (&pres_cur)[G],
                   raw4 (&a_nxt)[G][MAXCC], unsigned long long (&pres_nxt)[G]) {
    const int c0 = (p % npass) * KC;
    const int kc = min(KC, ci - c0);
    const T* wbuf = wl + (p & 1) * BN * WSTRIDE;
    const bool more = p + 1 < nphase;
    // ---- next phase's weight slab (first) and A fragments go in flight before this phase's MFMAs.
    // The last phase issues the same number of loads, aimed at the zero page: with the loads under
    // `if (more)` hipcc must pick ONE vmcnt for the MFMAs' wait that is safe on the path without
    // loads, and on the path with loads that count also waits for the 11 loads just issued --
    // every phase then sat out its own prefetch (profiles/README.md).
    stage_load(w1.k, w1.pass * KC, more);
    load_a(a_nxt, pres_nxt, idx_nxt, w1.pass * KC, more);
    walk_next(w1);
    read_idx(idx_nxt, w1.k);               // for the phase after next; first used a phase from now
    // ---- MFMAs: every B fragment read from LDS feeds the G row groups; accumulators stay in
    //      registers for all K offsets.  A wave skips the phase when none of its 32 rows has a rule
    //      for this offset; otherwise the MFMA block is branch-free (per-group skipping cost more
    //      in scalar branches than it saved: profiles/README.md)
    bool any_present = false;
#pragma unroll
    for (int g = 0; g < G; ++g) any_present |= pres_cur[g] != 0ull;
    if (any_present) {
      const T* wbase = wbuf + row16 * WSTRIDE + gsel * VEC;
#pragma unroll
      for (int cc = 0; cc < MAXCC; ++cc) {
        if (cc * CH < kc) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            frag b = *reinterpret_cast<const frag*>(wbase + nb * 16 * WSTRIDE + cc * CH);
#pragma unroll
            for (int g = 0; g < G; ++g) mma(acc[g][nb], __builtin_bit_cast(frag, a_cur[g][cc]), b);
          }
        }
      }
    }
    if (more) stage_store((p + 1) & 1);     // waits for the slab loads only: they were issued first
    __syncthreads();
  };
  for (int p = 0; p < nphase; p += 2) {
    phase(p, a0, pres0, a1, pres1);
    if (p + 1 < nphase) phase(p + 1, a1, pres1, a0, pres0);
  }

  // ---- epilogue: accumulators (D layout: col = lane&15, row = 4*(lane>>4) + r) -> wave-private
  //      LDS tile in T -> whole rows to HBM with 16-byte stores
  T* et = wl + wave * RW * ESTRIDE;
  if (ep_scale != nullptr) {       // inference: y = act(acc * scale[col] + shift[col]) (folded BN)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = n0 + nb * 16 + row16;
      const float es = col < co ? ep_scale[col] : 1.f, eh = col < co ? ep_shift[col] : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[g][nb][r] * es + eh;
          acc[g][nb][r] = ((ep_relu & 1) && v < 0.f) ? 0.f : v;
        }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        et[(g * 16 + gsel * 4 + r) * ESTRIDE + nb * 16 + row16] = DT<T>::from_f32(acc[g][nb][r]);
  __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): wave-private tile written
  constexpr int RSEGS = BN / VEC;                          // 16-byte segments per row
  for (int i = lane; i < RW * RSEGS; i += 64) {
    const int r = i / RSEGS, cseg = (i - r * RSEGS) * VEC;
    if (r0 + r >= n_out) continue;
    const int64_t row = perm ? (int64_t)perm[r0 + r] : r0 + r;
    T* dst = out + row * co + n0 + cseg;
    const T* srcp = et + r * ESTRIDE + cseg;
    if (n0 + cseg + VEC <= co) {
      frag v = *reinterpret_cast<const frag*>(srcp);
      if (ep_res != nullptr) {       // + residual row (same row, same columns), summed in f32
        const frag rr = *reinterpret_cast<const frag*>(ep_res + row * co + n0 + cseg);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = DT<T>::to_f32(v[e]) + DT<T>::to_f32(rr[e]);
          if ((ep_relu & 2) && f < 0.f) f = 0.f;          // ReLU of the residual block's sum
          v[e] = DT<T>::from_f32(f);
        }
      }
      *reinterpret_cast<frag*>(dst) = v;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (n0 + cseg + e < co) {
          float v = DT<T>::to_f32(srcp[e]);
          if (ep_res != nullptr) {
            v += DT<T>::to_f32(ep_res[row * co + n0 + cseg + e]);
            if ((ep_relu & 2) && v < 0.f) v = 0.f;
          }
          dst[e] = DT<T>::from_f32(v);
        }
    }
  }
}

struct Epi { const float* scale; const float* shift; int relu; const void* res; unsigned in_bytes, wk_bytes; };

template <typename T, int NB, int ROW_BYTES, bool GUARD>
int launch_conv_apply(const void* in, const void* wk, const int* nbr, const int* perm,
                      const unsigned* tmasks, void* out, int64_t n_out, int ci, int co, int K,
                      int kflip, Epi ep, hipStream_t s) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int G = conv_groups(NB, F32), NWAVES = conv_waves(NB, F32);
  constexpr int NTHREADS = 64 * NWAVES, BM = NWAVES * G * 16;
  static_assert(BM == 128 || BM == 64, "tile masks from lidal_kmap_order are per 128 rows");
  constexpr int BN = 16 * NB;
  constexpr int KC = ROW_BYTES / (int)sizeof(T);
  constexpr int WSTRIDE = KC + DT<T>::VEC;
  constexpr int ESTRIDE = BN + DT<T>::VEC;
  constexpr int WREGION = (2 * BN * WSTRIDE > NWAVES * G * 16 * ESTRIDE) ? 2 * BN * WSTRIDE
                                                                           : NWAVES * G * 16 * ESTRIDE;
  const size_t lds = sizeof(T) * WREGION + (size_t)NWAVES * K * G * 16 * sizeof(int);
  auto kern = conv_apply_kernel<T, NB, ROW_BYTES, GUARD, G, NWAVES>;
  // the attribute is per device: cache what was set per device id (one process may drive several)
  static size_t attr_set[MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = lds;
  }
  dim3 grid((unsigned)cdiv(n_out, BM), (unsigned)cdiv(co, BN));
  kern<<<grid, NTHREADS, lds, s>>>((const T*)in, (const T*)wk, nbr, perm, tmasks, (T*)out, n_out,
                                   ci, co, K, kflip, ep.scale, ep.shift, ep.relu, (const T*)ep.res,
                                   ep.in_bytes, ep.wk_bytes);
  LIDAL_CHECK_LAUNCH("lidal_conv_apply");
  return 0;
}

template <typename T, int ROW_BYTES, bool GUARD>
int dispatch_conv_cols(const void* in, const void* wk, const int* nbr, const int* perm,
                       const unsigned* tmasks, void* out, int64_t n_out, int ci, int co, int K,
                       int kflip, Epi ep, hipStream_t s) {
  // BN = 16*NB output channels per workgroup; grid.y covers the rest.
  if (co <= 32)
    return launch_conv_apply<T, 2, ROW_BYTES, GUARD>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  // 64-column blocks also for wide layers on the smallest levels: with fewer 128-column workgroups
  // than ~1.5 per CU the chip is under-filled and each workgroup is one long serial chain of phases
  // (256->256 on 17k rows: 106 -> 95 us; on 43k rows the extra gather passes lose, 142 -> 178)
  if (co <= 64 || (co % 64 == 0 && cdiv(n_out, 128) * cdiv(co, 128) <= 384))
    return launch_conv_apply<T, 4, ROW_BYTES, GUARD>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  if (co % 128 != 0 && (co % 96 == 0 || co < 128))
    return launch_conv_apply<T, 6, ROW_BYTES, GUARD>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  return launch_conv_apply<T, 8, ROW_BYTES, GUARD>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
}

template <typename T>
int dispatch_conv_apply(const void* in, const void* wk, const int* nbr, const int* perm,
                        const unsigned* tmasks, void* out, int64_t n_out, int ci, int co, int K,
                        int kflip, Epi ep, hipStream_t s) {
  // staged reduction bytes per pass: 128 (more workgroups per CU beat longer passes: measured in
  // profiles/README.md), except rows that are a multiple of 192 but not of 128 bytes (ci = 96
  // bf16 -> one pass of 96 instead of 64 + 32).
  const int row_bytes = ci * (int)sizeof(T);
  if (ci % DT<T>::VEC != 0)       // irregular channel count (4-channel bf16 stem): guarded loads
    return dispatch_conv_cols<T, 128, true>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  if (row_bytes % 192 == 0 && row_bytes % 128 != 0)
    return dispatch_conv_cols<T, 192, false>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  return dispatch_conv_cols<T, 128, false>(in, wk, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
}

// ------------------------------------------------------------------------------------------
// weight pack: W[k][ci][co] -> Wt[k][co][ci] (+cast); optionally also Wc[k][ci][co] = cast(W),
// the operand of the data gradient, from the same read
// ------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) weight_pack_kernel(const TI* __restrict__ w,
                                                          TO* __restrict__ wt, TO* __restrict__ wc,
                                                          int K, int ci, int co) {
  __shared__ float tile[32][33];
  const int k = blockIdx.z;
  const int i0 = blockIdx.y * 32, o0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const TI* src = w + (int64_t)k * ci * co;
  TO* dst = wt + (int64_t)k * ci * co;
  for (int r = ty; r < 32; r += 8) {
    int i = i0 + r, o = o0 + tx;
    const bool ok = i < ci && o < co;
    const float v = ok ? DT<TI>::to_f32(src[(int64_t)i * co + o]) : 0.f;
    tile[r][tx] = v;
    if (wc != nullptr && ok) wc[(int64_t)k * ci * co + (int64_t)i * co + o] = DT<TO>::from_f32(v);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int o = o0 + r, i = i0 + tx;
    if (o < co && i < ci) dst[(int64_t)o * ci + i] = DT<TO>::from_f32(tile[tx][r]);
  }
}

}  // namespace


extern "C" int lidal_conv_weight_pack(const void* w, int w_dtype, void* wt, void* wc, int wt_dtype,
                                      int k, int ci, int co, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (k == 0 || ci == 0 || co == 0) return 0;
  dim3 grid((unsigned)cdiv(co, 32), (unsigned)cdiv(ci, 32), (unsigned)k);
  if (w_dtype == LIDAL_F32 && wt_dtype == LIDAL_F32)
    weight_pack_kernel<float, float><<<grid, 256, 0, s>>>((const float*)w, (float*)wt, (float*)wc, k, ci, co);
  else if (w_dtype == LIDAL_F32 && wt_dtype == LIDAL_BF16)
    weight_pack_kernel<float, __bf16><<<grid, 256, 0, s>>>((const float*)w, (__bf16*)wt, (__bf16*)wc, k, ci, co);
  else if (w_dtype == LIDAL_BF16 && wt_dtype == LIDAL_BF16)
    weight_pack_kernel<__bf16, __bf16><<<grid, 256, 0, s>>>((const __bf16*)w, (__bf16*)wt, (__bf16*)wc, k, ci, co);
  else if (w_dtype == LIDAL_BF16 && wt_dtype == LIDAL_F32)
    weight_pack_kernel<__bf16, float><<<grid, 256, 0, s>>>((const __bf16*)w, (float*)wt, (float*)wc, k, ci, co);
  else {
    set_error("weight_pack: bad dtypes %d %d", w_dtype, wt_dtype);
    return 2;
  }
  LIDAL_CHECK_LAUNCH("lidal_conv_weight_pack");
  return 0;
}

extern "C" int lidal_conv_apply(const void* in, const void* wk, const int32_t* nbr,
                                const int32_t* perm, const uint32_t* tile_masks, void* out,
                                int64_t n_in, int64_t n_out, int ci, int co, int k, int kflip,
                                int dtype, const float* ep_scale, const float* ep_shift,
                                int ep_relu, const void* ep_residual, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out == 0 || co == 0) return 0;
  LIDAL_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv_apply: scale and shift go together");
  const int64_t esz = dtype == LIDAL_BF16 ? 2 : 4;
  LIDAL_REQUIRE(n_in >= 0 && n_in * ci * esz < 0x7FFFFFF0ll && (int64_t)k * ci * co * esz < 0x7FFFFFF0ll,
                "conv_apply: the input matrix (%lld rows x %d) and the weights must each stay below "
                "2 GiB (32-bit buffer addressing)", (long long)n_in, ci);
  Epi ep{ep_scale, ep_shift, ep_relu, ep_residual, (unsigned)(n_in * ci * esz),
         (unsigned)((int64_t)k * ci * co * esz)};
  LIDAL_REQUIRE(ci > 0 && k > 0 && k <= MAXK, "conv_apply: bad shape ci=%d k=%d", ci, k);
  LIDAL_REQUIRE(nbr != nullptr || (k == 1 && n_in == n_out), "conv_apply: a NULL table means the identity (k = 1)");
  if (dtype == LIDAL_F32) {
    LIDAL_REQUIRE(ci % 4 == 0 && co % 4 == 0, "conv_apply f32: channels must be multiples of 4");
    return dispatch_conv_apply<float>(in, wk, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
  }
  if (dtype == LIDAL_BF16) {
    LIDAL_REQUIRE((ci % 8 == 0 || ci < 8) && co % 4 == 0,
                  "conv_apply bf16: ci must be a multiple of 8 (or < 8), co a multiple of 4");
    return dispatch_conv_apply<__bf16>(in, wk, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
  }
  set_error("conv_apply: bad dtype %d", dtype);
  return 2;
}
