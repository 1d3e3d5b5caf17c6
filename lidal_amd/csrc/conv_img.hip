// Sparse 3D convolution for gfx950, second generation of lidal_conv_apply (conv.hip): same dataflow
// (output-stationary, register accumulators, A fragments gathered straight into registers one phase
// ahead, weights of the current offset shared by the workgroup through LDS) with the weight stream
// moved off the vector registers:
//
//   * weights arrive as LDS IMAGES: lidal_conv_weight_image lays every slab (offset k, column block,
//     reduction slice) out in global memory exactly as the MFMA B-fragment reads want it in LDS,
//         image[cc][gsel][nb][row16] x 16 bytes  =  Wt[n0 + 16 nb + row16][c0 + CH cc + VEC gsel ...]
//     so that (a) a slab is staged by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KiB per wave
//     instruction, no VGPRs, no ds_write) as a linear copy and (b) every ds_read_b128 of a
//     fragment is bank-conflict free without padding (the 16-byte slot of an access is row16,
//     and each of ds_read_b128's four 16-lane groups covers all sixteen row16 values once);
//   * the vector registers and LDS that frees (12 VGPRs of staging, the 16-byte row pad, the
//     index slices of absent offsets) buy a third resident workgroup per CU on the 96-column
//     kernels; the phase loop has no weight-store segment any more.
//
// Everything else -- row order by occupancy pattern, per-tile offset masks, buffer addressing with
// the hardware range check serving absent rules, scalar offset walk, copy-free two-set software
// pipeline, epilogues -- is as described in conv.hip / DESIGN.md.
#include <type_traits>

#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef int raw4 __attribute__((ext_vector_type(4)));

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int VEC = 4;
  static constexpr int CH = 16;
  typedef f32x4 frag;
  __device__ static float to_f32(float v) { return v; }
  __device__ static float from_f32(float v) { return v; }
};
template <> struct DT<__bf16> {
  static constexpr int VEC = 8;
  static constexpr int CH = 32;
  typedef bf16x8 frag;
  __device__ static float to_f32(__bf16 v) { return (float)v; }
  __device__ static __bf16 from_f32(float v) { return (__bf16)v; }
};

__device__ __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

constexpr int MAXK = 32;

constexpr int NB4_LIMIT = 384;        // tiles up to which a 64-multiple column count takes 64-column blocks
// ---- tiling policy, shared by the image packer and the launcher ------------------------------
struct Tiling { int nb; int row_bytes; };       // 16-column blocks per workgroup, staged bytes per pass

__host__ __device__ inline Tiling pick_tiling(int ci, int co, int64_t n_out, int esz) {
  Tiling t;
  const int row_bytes = ci * esz;
  // staged bytes of the reduction dim per pass: 192 for rows that are whole 192-byte but not 128-byte
  // multiples (96 bf16 channels), 64 for the other rows that 128 does not divide (32 bf16 channels)
  t.row_bytes = (row_bytes % 192 == 0 && row_bytes % 128 != 0) ? 192
                : (row_bytes % 128 != 0 && row_bytes % 64 == 0) ? 64 : 128;
  // (256-byte slices -- half the phases per offset on the layers of >= 128 channels -- measured no
  // better, profiles/README.md; the 256-byte kernels stay instantiated for the image packer's sake)
  if (co <= 32) t.nb = 2;
  else if (co <= 64 || (co % 64 == 0 && ((n_out + 127) / 128) * ((co + 127) / 128) <= NB4_LIMIT)) t.nb = 4;
  else if (co % 128 != 0 && (co % 96 == 0 || co < 128)) t.nb = 6;
  else t.nb = 8;
  return t;
}
__host__ __device__ inline int64_t image_bytes(int k, int ci, int co, Tiling t, int esz) {
  const int bn = 16 * t.nb, kc = t.row_bytes / esz;
  const int nblk = (co + bn - 1) / bn, npass = (ci + kc - 1) / kc;
  return (int64_t)k * nblk * npass * bn * t.row_bytes;
}

// image[k][nblk][pass][cc][gsel][nb][row16][VEC]; one thread per 16-byte segment.
// role 0: W[k][red][col] (forward: red = ci, col = co);  role 1: W[k][col][red] (data gradient)
template <typename TI, typename TO>
__device__ __forceinline__ void image_segment(const TI* __restrict__ w, TO* __restrict__ img, int n_red,
                                              int n_col, int role, int nb, int kc, int64_t s) {
  constexpr int VEC = DT<TO>::VEC, CH = DT<TO>::CH;
  const int bn = 16 * nb, ncc = kc / CH;
  const int nblk = (n_col + bn - 1) / bn, npass = (n_red + kc - 1) / kc;
  // (an image has < 2^31 segments: 32-bit divisions.  Measured: no change -- 136 us for the 22 M parameters of SPVCNN,
  // 1.3 TB/s; the data-gradient images read 32-byte pieces of 64 different rows per wave, that is what it waits for)
  unsigned r = (unsigned)s;
  const int row16 = (int)(r & 15u); r >>= 4;
  const int b = (int)(r % (unsigned)nb); r /= (unsigned)nb;
  const int gsel = (int)(r & 3u); r >>= 2;
  const int cc = (int)(r % (unsigned)ncc); r /= (unsigned)ncc;
  const int pass = (int)(r % (unsigned)npass); r /= (unsigned)npass;
  const int blk = (int)(r % (unsigned)nblk); r /= (unsigned)nblk;
  const int k = (int)r;
  const int col = blk * bn + b * 16 + row16;
  const int red0 = pass * kc + cc * CH + gsel * VEC;
  if (role != 0) {
    // the reduction index is the contiguous one of w (data-gradient images of a convolution, forward images of nn.Linear):
    // the thread of gsel == 0 builds the four segments of its (cc, column) -- CH consecutive elements of ONE row of w, read
    // as 16-byte pieces -- and the threads of gsel 1..3 have nothing to do.  (Rounds 2-4: every thread read its own 32 bytes
    // of a row, a wave 64 rows at a time: 136 us for the 22 M parameters of SPVCNN at the head of every step.)
    if (gsel != 0) return;
    const bool inside = col < n_col;
    const int64_t src = (int64_t)k * n_red * n_col + (int64_t)col * n_red + red0;
    float f[CH];
    if (inside && red0 + CH <= n_red && sizeof(TI) == 4 && (n_red & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
#pragma unroll
      for (int e = 0; e < CH; e += 4) {
        const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(w) + src + e);
        f[e] = q.x; f[e + 1] = q.y; f[e + 2] = q.z; f[e + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] = (inside && red0 + e < n_red) ? DT<TI>::to_f32(w[src + e]) : 0.f;
    }
    typedef typename DT<TO>::frag frag;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      TO v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = DT<TO>::from_f32(f[g * VEC + e]);
      *reinterpret_cast<frag*>(img + (s + (int64_t)g * nb * 16) * VEC) = *reinterpret_cast<frag*>(v);
    }
    return;
  }
  TO v[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int red = red0 + e;
    float f = 0.f;
    if (col < n_col && red < n_red) {
      const int64_t base = (int64_t)k * n_red * n_col;
      f = DT<TI>::to_f32(role == 0 ? w[base + (int64_t)red * n_col + col] : w[base + (int64_t)col * n_red + red]);
    }
    v[e] = DT<TO>::from_f32(f);
  }
  typedef typename DT<TO>::frag frag;
  *reinterpret_cast<frag*>(img + s * VEC) = *reinterpret_cast<frag*>(v);
}

// one launch builds up to two images of the same parameter: segments [0, segs_a) of image A (role_a,
// n_red_a x n_col_a), then segs_b segments of image B with the roles of the two channel dims swapped
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
weight_image_kernel(const TI* __restrict__ w, TO* __restrict__ img_a, int n_red, int n_col, int role,
                    int nb_a, int kc_a, int64_t segs_a, TO* __restrict__ img_b, int nb_b, int kc_b,
                    int64_t segs_b) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < segs_a) image_segment<TI, TO>(w, img_a, n_red, n_col, role, nb_a, kc_a, s);
  else if (s < segs_a + segs_b) image_segment<TI, TO>(w, img_b, n_col, n_red, role ^ 1, nb_b, kc_b, s - segs_a);
}

// every parameter of a model in ONE launch (training: the weights change every step, and 42 launches of
// 7 us each were 0.3 ms of the step): job j owns segments [first, first + segs_a + segs_b) of the grid
struct ImageJob {
  const void* w; void* img_a; void* img_b;
  long long first, segs_a, segs_b;
  int n_red, n_col, role, nb_a, kc_a, nb_b, kc_b, pad;
};
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
weight_image_batch_kernel(const ImageJob* __restrict__ jobs, int n_jobs, int64_t total) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {                       // last job whose first segment is <= s
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= s) lo = mid; else hi = mid - 1;
  }
  const ImageJob j = jobs[lo];
  const int64_t l = s - j.first;
  if (l < j.segs_a) image_segment<TI, TO>((const TI*)j.w, (TO*)j.img_a, j.n_red, j.n_col, j.role, j.nb_a, j.kc_a, l);
  else if (l < j.segs_a + j.segs_b)
    image_segment<TI, TO>((const TI*)j.w, (TO*)j.img_b, j.n_col, j.n_red, j.role ^ 1, j.nb_b, j.kc_b, l - j.segs_a);
}

// Tuning constants of the lean kernel (the measured alternatives are in profiles/README.md; the timing-only
// ablation switches of rounds 1-2 lived here and are gone: scripts/exp/README.md names the commit that has them)
// rows per workgroup tile of every kernel of this file == rows per BatchNorm statistics triple
constexpr int TILE_ROWS = 128;
// Which tile a workgroup takes: the LAST tile of the row order first (round 5).  The row order (lidal_kmap_order: rows
// sorted by their Gray-ranked offset pattern, rare offsets most significant) puts the tiles with the MOST active offsets
// at its end -- on the level-0 map of the bench batch the first twelfth of the tiles runs 1 phase each, the last 14.8
// on average and up to 27 (scripts/exp/tile_weights.py) -- and the hardware dispatches workgroups in index order: the
// longest tiles started last and the launch ended with a few of them alone on the chip.  Longest first is the classic
// list-scheduling rule; as a reversal it costs nothing: 96 -> 96 at stride 1 90.2 -> 82.2-83.6 us, 32 -> 32 39.7 -> 34.8,
// 128 -> 128 at stride 4 66.8 -> 58.7 (same results bit for bit; LIDAL_TILE_ORDER_FORWARD restores index order).
__device__ __forceinline__ int tile_of_block() {
#ifdef LIDAL_TILE_ORDER_FORWARD
  return (int)blockIdx.x;
#else
  return (int)gridDim.x - 1 - (int)blockIdx.x;
#endif
}
#ifndef LIDAL_LEAN_WAVES
#define LIDAL_LEAN_WAVES 8
#endif
// waves per workgroup = 128-row tiles.  16 (256-row tiles, scripts/build_variant.py -DLIDAL_LEAN_WAVES=16: an experiment
// build -- no BatchNorm tile statistics, no offset split) halves the slab DMAs per row
constexpr int LEAN_WAVES = LIDAL_LEAN_WAVES;
#ifndef LIDAL_LEAN_MINWAVES
#define LIDAL_LEAN_MINWAVES 4
#endif
// waves per SIMD the register budget must allow: 4 = 2 workgroups per CU (94 VGPRs).  6 (a third workgroup, 80 VGPRs)
// spills 52 registers: the 96 -> 96 layer 90 -> 210 us, the 5-scan step 14.7 -> 19.5 ms (scripts/build_variant.py mw6)
constexpr int LEAN_MINWAVES = LIDAL_LEAN_MINWAVES;
constexpr int64_t DEEP_MAX_ROWS = 150000;      // up to this many output rows the deep form of the lean kernel runs
// Optional second job of a DATA-GRADIENT launch: the tile's share of the backward sums of the BatchNorm
// whose output gradient this launch produces (out = dy of y = act(bn(x))): per column sum(dy') and
// sum(dy' * xhat), dy' = dy where the fused ReLU let the value through, xhat = (x - mean) * invstd -- x read
// row by row at the tile's own output rows.  lidal_bn_bwd_tiles merges the tiles (in f64); the separate
// pass over x and dy (bn_bwd_partial_kernel, two of BatchNorm backward's five passes) is gone.
struct BnBwd {
  const void* x; const float* mean; const float* invstd; const float* gamma; const float* beta;
  float* sums;          // f32 [tiles][co][2], or NULL: no such job
  int relu;
};

// Optional split of a tile's offsets over SEVERAL workgroups (round 4).  On the coarse levels a launch is as long as
// its heaviest tile's chain of phases (a 128-row tile that touches all 27 offsets of a 256-channel layer runs 108
// phases back to back) while most of the chip idles (29 tiles x 4 column blocks on 256 CUs).  With nsplit > 1 the
// launch has nsplit workgroups per (tile, column block): workgroup z takes the z-th share of the tile's ACTIVE
// offsets (by rank among the set bits of the tile mask), leaves its f32 accumulators in `partial` (one 16-byte
// vector per lane and 16-column block, [nsplit][tile][column block][wave][16-column block][lane]: both kernels use
// the MFMA accumulator layout, so the traffic is fully coalesced) and skips the epilogue; conv_combine_kernel adds
// the shares in split order and runs the epilogue (row permutation, affine map / ReLU / residual, BatchNorm tile
// statistics, BatchNorm backward sums) exactly as the unsplit kernel does.  Deterministic; differs from the unsplit
// result only by the association of the f32 sum over the offsets.
struct Split { float* partial; int nsplit; int co_pad; long long rows_pad; };

// ---- epilogue shared by the kernels of this file (as conv.hip): accumulators (D layout: col =
// lane&15, row = 4*(lane>>4) + r) -> wave-private LDS tile in T -> whole rows to HBM, 16-byte stores,
// with the optional affine map / ReLU / residual of the inference paths
template <typename T, int NB, int G, int NWAVES>
__device__ __forceinline__ void store_tile(f32x4 (&acc)[G][NB], unsigned char* wl, int wave, int lane,
                                           int64_t r0, int n0, int64_t n_out, int co,
                                           const int* __restrict__ perm, T* __restrict__ out,
                                           const float* __restrict__ ep_scale,
                                           const float* __restrict__ ep_shift, int ep_relu,
                                           const T* __restrict__ ep_res, bool perm_in_reg = false,
                                           int perm_v = 0, float* __restrict__ tile_stats = nullptr,
                                           int stats_tile = -1, const BnBwd* bnb = nullptr) {
  // perm_in_reg: lane l (< 16 G) of the wave holds perm[r0 + l] in perm_v, loaded when the tile began
  // (the lean kernel: no dependent load in front of the stores)
  constexpr int VEC = DT<T>::VEC;
  constexpr int BN = 16 * NB;
  constexpr int RW = G * 16;
  constexpr int ESTRIDE = BN + VEC;
  typedef typename DT<T>::frag frag;
  const int row16 = lane & 15, gsel = lane >> 4;
  T* et = reinterpret_cast<T*>(wl) + wave * RW * ESTRIDE;
  if (ep_scale != nullptr) {
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
  __builtin_amdgcn_s_waitcnt(0xC07F);
  constexpr int RSEGS = BN / VEC;
  for (int i = lane; i < RW * RSEGS; i += 64) {
    const int r = i / RSEGS, cseg = (i - r * RSEGS) * VEC;
    const int prow = perm_in_reg ? __shfl(perm_v, r & (RW - 1), 64) : 0;      // all lanes take part
    if (r0 + r >= n_out) continue;
    const int64_t row = perm ? (perm_in_reg ? (int64_t)prow : (int64_t)perm[r0 + r]) : r0 + r;
    T* dst = out + row * co + n0 + cseg;
    const T* srcp = et + r * ESTRIDE + cseg;
    if (n0 + cseg + VEC <= co) {
      frag v = *reinterpret_cast<const frag*>(srcp);
      if (ep_res != nullptr) {
        const frag rr = *reinterpret_cast<const frag*>(ep_res + row * co + n0 + cseg);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = DT<T>::to_f32(v[e]) + DT<T>::to_f32(rr[e]);
          if ((ep_relu & 2) && f < 0.f) f = 0.f;
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
  // ---- BatchNorm batch statistics of the tile (training: the layer that follows is a train-mode
  // BatchNorm): per column (count, mean, M2) of the values AS STORED, over the tile's valid rows.
  // Per wave in registers (two-pass: mean, then squared deviations), waves merged through LDS with
  // Chan's formula; lidal_bn_train_fwd_tiles merges the tiles (in f64).  Replaces the statistics
  // pass over the stored matrix.
  if (tile_stats != nullptr) {
    constexpr int EPI_BYTES = NWAVES * RW * ESTRIDE * (int)sizeof(T);
    float* st = reinterpret_cast<float*>(wl + EPI_BYTES);          // [NWAVES][BN][2]
    const int64_t left = n_out - r0;
    const int nvalid = left <= 0 ? 0 : (left < RW ? (int)left : RW);
    const float inv_n = nvalid > 0 ? 1.f / (float)nvalid : 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v[G][4];
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[g][r] = DT<T>::to_f32(DT<T>::from_f32(acc[g][nb][r]));
          if (g * 16 + gsel * 4 + r < nvalid) sum += v[g][r];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * inv_n;
      float m2 = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (g * 16 + gsel * 4 + r < nvalid) { const float d = v[g][r] - mean; m2 += d * d; }
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      if (gsel == 0) {
        st[(wave * BN + nb * 16 + row16) * 2] = mean;
        st[(wave * BN + nb * 16 + row16) * 2 + 1] = m2;
      }
    }
    __syncthreads();
    const int c = wave * 64 + lane;
    if (c < BN && n0 + c < co) {
      const int64_t rb = r0 - (int64_t)wave * RW;                  // first row of the workgroup's tile
      float na = 0.f, ma = 0.f, qa = 0.f;
      for (int w = 0; w < NWAVES; ++w) {
        const int64_t lw = n_out - (rb + (int64_t)w * RW);
        const float nw = lw <= 0 ? 0.f : (lw < RW ? (float)lw : (float)RW);
        if (nw > 0.f) {
          const float mw = st[(w * BN + c) * 2], qw = st[(w * BN + c) * 2 + 1];
          const float n = na + nw, d = mw - ma;
          ma += d * (nw / n);
          qa += qw + d * d * (na * nw / n);
          na = n;
        }
      }
      // [column][tile][3] (round 5): the merge of a channel -- on the critical path of the BatchNorm launch that follows --
      // reads ONE contiguous run instead of a 12-byte piece of every tile's row (7-9 -> 3-4 us on the 397 k-row levels)
      float* dst = tile_stats + ((int64_t)(n0 + c) * gridDim.x + (stats_tile >= 0 ? stats_tile : (int)blockIdx.x)) * 3;
      dst[0] = na; dst[1] = ma; dst[2] = qa;
    }
  }
  // ---- BatchNorm backward sums of the tile (see BnBwd): this wave's rows of x come in as whole rows (16-byte
  // lane loads, the tile's own output rows) through the wave's epilogue tile, which the write-out has left
  if (bnb != nullptr && bnb->sums != nullptr) {
    constexpr int EPI_BYTES = NWAVES * RW * ESTRIDE * (int)sizeof(T);
    float* st = reinterpret_cast<float*>(wl + EPI_BYTES);          // [NWAVES][BN][2]
    const T* bx = reinterpret_cast<const T*>(bnb->x);
    const int64_t left = n_out - r0;
    const int nvalid = left <= 0 ? 0 : (left < RW ? (int)left : RW);
    __builtin_amdgcn_s_waitcnt(0xC07F);                            // the write-out's LDS reads are done
    for (int i = lane; i < RW * RSEGS; i += 64) {
      const int r = i / RSEGS, cseg = (i - r * RSEGS) * VEC;
      const int prow = perm_in_reg ? __shfl(perm_v, r & (RW - 1), 64) : 0;
      frag v;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = DT<T>::from_f32(0.f);
      if (r < nvalid && n0 + cseg + VEC <= co) {
        const int64_t row = perm ? (perm_in_reg ? (int64_t)prow : (int64_t)perm[r0 + r]) : r0 + r;
        v = *reinterpret_cast<const frag*>(bx + row * co + n0 + cseg);
      }
      *reinterpret_cast<frag*>(et + r * ESTRIDE + cseg) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = n0 + nb * 16 + row16;
      const bool cok = col < co;
      const float mu = cok ? bnb->mean[col] : 0.f, is = cok ? bnb->invstd[col] : 0.f;
      const float ga = (cok && bnb->gamma) ? bnb->gamma[col] : 1.f, be = (cok && bnb->beta) ? bnb->beta[col] : 0.f;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = g * 16 + gsel * 4 + r;
          if (rr < nvalid) {
            const float xh = (DT<T>::to_f32(et[rr * ESTRIDE + nb * 16 + row16]) - mu) * is;
            float dy = DT<T>::to_f32(DT<T>::from_f32(acc[g][nb][r]));          // as stored
            if (bnb->relu && !(xh * ga + be > 0.f)) dy = 0.f;
            a += dy; b += dy * xh;
          }
        }
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (gsel == 0) {
        st[(wave * BN + nb * 16 + row16) * 2] = a;
        st[(wave * BN + nb * 16 + row16) * 2 + 1] = b;
      }
    }
    __syncthreads();
    const int c = wave * 64 + lane;
    if (c < BN && n0 + c < co) {
      float a = 0.f, b = 0.f;
      for (int w = 0; w < NWAVES; ++w) { a += st[(w * BN + c) * 2]; b += st[(w * BN + c) * 2 + 1]; }
      float* dst = bnb->sums + ((int64_t)(n0 + c) * gridDim.x + (stats_tile >= 0 ? stats_tile : (int)blockIdx.x)) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
constexpr int IMG_MINWAVES = 2;

// LDS (dynamic): ring of D+1 weight slabs (re-used as the epilogue tile) | dump 1 KiB.  The
// neighbour indices never touch LDS: each lane loads the index of ITS row straight from the permuted
// table (64 contiguous bytes per 16-row group) D phases ahead of the gather that uses it.
template <typename T, int NB, int ROW_BYTES, int G, int NWAVES, int MINW, bool DENSE, int D>
__global__ void __launch_bounds__(64 * NWAVES, MINW)
conv_apply_img_kernel(const T* __restrict__ in, const T* __restrict__ wimg,
                      const int* __restrict__ nbr, const int* __restrict__ perm,
                      const unsigned* __restrict__ tmasks, T* __restrict__ out, int64_t n_out,
                      int ci, int co, int K, int kflip, const float* __restrict__ ep_scale,
                      const float* __restrict__ ep_shift, int ep_relu, const T* __restrict__ ep_res,
                      unsigned in_bytes, unsigned img_bytes, unsigned nbr_bytes,
                      float* __restrict__ tile_stats, BnBwd bnb) {
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int BM = NWAVES * G * 16;
  constexpr int BN = 16 * NB;
  constexpr int VEC = DT<T>::VEC;
  constexpr int CH = DT<T>::CH;
  constexpr int KC = ROW_BYTES / (int)sizeof(T);
  constexpr int MAXCC = KC / CH;
  constexpr int SLAB = BN * ROW_BYTES;                   // bytes of one staged slab
  constexpr int PIECES = SLAB / 1024;                    // 1 KiB LDS-DMA pieces per slab
  constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;    // pieces per wave (the surplus hits the dump)
  constexpr int RW = G * 16;
  constexpr int ESTRIDE = BN + VEC;
  constexpr int EPI = NWAVES * RW * ESTRIDE * (int)sizeof(T);
  constexpr int R = D + 1;                               // ring slots: slabs, A register sets, indices
  constexpr int WREGION = (R * SLAB > EPI) ? R * SLAB : EPI;
  constexpr int GA = G * MAXCC;                          // A gathers per lane and phase
  static_assert(D >= 1 && D <= 3, "pipeline depth 1..3");
  static_assert(SLAB % 1024 == 0, "slab must be whole LDS-DMA pieces");
  typedef typename DT<T>::frag frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;
  unsigned char* dump = smem + WREGION;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row16 = lane & 15;
  const int gsel = lane >> 4;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * RW;
  const int n0 = blockIdx.y * BN;
  const int npass = (ci + KC - 1) / KC;

  // ---- tile mask: the OR of the 128-row masks this tile covers (scalar loads)
  unsigned tmask;
  if constexpr (DENSE) {          // no table at all: the identity rule list of a per-row product (K == 1)
    tmask = 1u;
  } else {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
#pragma unroll
    for (int h = 0; h < (BM >= 128 ? BM / 128 : 1); ++h)
      if ((t0 + h) * 128 < n_out) m |= tmasks[t0 + h];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = m;
  }
  tmask = __builtin_amdgcn_readfirstlane(tmask);
  const int n_act = __popc(tmask);
  const int nphase = n_act * npass;

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
      w.k = w.rem ? __builtin_ctz(w.rem) : 0;      // past the end: any valid offset (result unused)
      w.rem &= w.rem - 1u;
    }
  };

  constexpr unsigned OOB_OFF = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);
  const int nblk = gridDim.y;

  // slab (k, this column block, pass) -> LDS buffer `buf` by LDS-DMA: wave w moves pieces
  // w, w + NWAVES, ...; every wave issues exactly PPW instructions (a surplus piece, or a dead
  // phase, reads out of range -- zeros, no memory traffic -- into the dump)
  auto stage_dma = [&](int k, int pass, int slot, bool live) {
    const unsigned slab_off = (unsigned)((((int64_t)k * nblk + blockIdx.y) * npass + pass) * SLAB);
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      const int piece = wave + t * NWAVES;
      const bool ok = live && piece < PIECES;
      unsigned char* dst = ok ? wl + slot * SLAB + piece * 1024 : dump;
      const unsigned soff = ok ? slab_off + (unsigned)piece * 1024u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16,
                                               (unsigned)lane * 16u, soff, 0, 0);
    }
  };
  // neighbour index of this lane's row in each row group for offset k: one 4-byte load per group
  // (the four lanes of a row share the address).  The raw loaded value is carried to its use a
  // phase later; rows past the end of the table are masked THERE (row_ok), so nothing touches
  // the value -- and nothing waits for the load -- in the phase that issues it.
  const unsigned row_off = (unsigned)((r0 + row16) * 4);
  bool row_ok[G];
#pragma unroll
  for (int g = 0; g < G; ++g) row_ok[g] = r0 + g * 16 + row16 < n_out;
  auto load_idx = [&](int (&dst)[G], int k) {
    if constexpr (DENSE) {
#pragma unroll
      for (int g = 0; g < G; ++g) dst[g] = (int)(r0 + g * 16 + row16);
    } else {
      const int kk = kflip ? (K - 1 - k) : k;
      const unsigned koff = (unsigned)kk * (unsigned)n_out * 4u;
#pragma unroll
      for (int g = 0; g < G; ++g)
        dst[g] = __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, row_off + (unsigned)(g * 64), koff, 0);
    }
  };
  auto kill_bit = [&](bool live) {
    unsigned kb = live ? 0u : OOB_OFF;
    asm volatile("" : "+s"(kb));
    return kb;
  };
  auto load_a = [&](raw4 (&a)[G][MAXCC], unsigned long long (&present)[G], const int (&idx)[G],
                    int c0, bool live) {
    const int kc = min(KC, ci - c0);
    const unsigned row_bytes = (unsigned)(ci * (int)sizeof(T));
    const unsigned lane_off = (unsigned)((c0 + gsel * VEC) * (int)sizeof(T));
    const unsigned kill = kill_bit(live);
    const unsigned long long live_mask = kill ? 0ull : ~0ull;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int src = row_ok[g] ? idx[g] : -1;
      present[g] = __ballot(src != -1) & live_mask;
      const unsigned base = ((src >= 0) ? (unsigned)src * row_bytes + lane_off : OOB_OFF) | kill;
#pragma unroll
      for (int cc = 0; cc < MAXCC; ++cc) {
        const unsigned off = (cc * CH + gsel * VEC < kc) ? base + (unsigned)(cc * CH * (int)sizeof(T)) : OOB_OFF;
        a[g][cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
      }
    }
  };

  f32x4 acc[G][NB];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[g][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline of depth D over rings of R = D + 1 slots (weight slabs in LDS, A-fragment
  // register sets, index registers).  Vector-memory operations retire IN ORDER, so a wait for the
  // weight slab of the next phase also waits for every load issued before that slab's DMA: to keep
  // D phases of gathers in flight the slab DMA runs D phases ahead as well, and the order of issue
  // inside a phase is what keeps every wait cheap.  Phase p (slot s = p mod R) issues
  //     index(p + 2D)  ->  slab(p + D) by DMA  ->  A(p + D)
  // then runs the MFMAs of phase p on A(p) / slab(p), then waits for slab(p + 1) and joins the
  // barrier.
  //   * A(p+D) is addressed from index(p+D), issued D phases ago FIRST in its phase: waiting for it
  //     leaves D phases of slabs and gathers in flight;
  //   * the MFMAs need A(p), issued D phases ago: everything younger stays in flight (hipcc counts
  //     this wait itself);
  //   * slab(p+1) was issued D-1 phases ago: the explicit s_waitcnt leaves A(p+1) and the D-1 whole
  //     phases of loads behind it in flight (hipcc does not order a ds_read behind an LDS-DMA
  //     write on its own);
  //   * the barrier that ends phase p frees slot s for slab(p + R), which phase p + 1 issues.
  // A phase lasts about (memory latency) / D instead of one memory latency (measured on the depth-1
  // form: the loop ran at one dependent HBM round trip per phase with nothing else on its critical
  // path, profiles/README.md).
  raw4 a[R][G][MAXCC];
  unsigned long long pres[R][G];
  int idx[R][G];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int g = 0; g < G; ++g) { pres[r][g] = 0ull; idx[r][g] = -1; }
  Walk wa = walk_begin();       // the phase whose slab / A are issued next
  Walk wi = walk_begin();       // the phase whose indices are issued next
  constexpr int IL = DENSE ? 0 : G;                       // index loads per phase
  // loads that may stay in flight while slab(p+1) is awaited (steady state and prologue alike)
  constexpr int TAIL = GA + (D - 1) * (IL + PPW + GA);
  if (nphase > 0) {
#pragma unroll
    for (int q = 0; q < R; ++q) { load_idx(idx[q], wi.k); walk_next(wi); }        // index(0..D)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      stage_dma(wa.k, wa.pass, j, j < nphase);
      load_a(a[j], pres[j], idx[j], wa.pass * KC, j < nphase);
      walk_next(wa);
      if (j < D - 1) { load_idx(idx[j], wi.k); walk_next(wi); }                   // index(D+1 .. 2D-1)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | (TAIL & 15) | ((TAIL >> 4) << 14));       // slab 0 landed
  }
  __syncthreads();

  auto phase = [&](auto slot_c, int p) {
    constexpr int s = decltype(slot_c)::value;
    constexpr int sa = (s + D) % R;              // slot of phase p + D (== the slot of phase p - 1)
    constexpr int si = (s + D - 1 + R) % R;      // slot of index(p + 2D)
    const int c0 = (p % npass) * KC;
    const int kc = min(KC, ci - c0);
    const unsigned char* wbuf = wl + s * SLAB;
    const bool more = p + D < nphase;
    load_idx(idx[si], wi.k);
    walk_next(wi);
    stage_dma(wa.k, wa.pass, sa, more);
    load_a(a[sa], pres[sa], idx[sa], wa.pass * KC, more);
    walk_next(wa);
    bool any_present = false;
#pragma unroll
    for (int g = 0; g < G; ++g) any_present |= pres[s][g] != 0ull;
    if (any_present) {
      const unsigned char* wbase = wbuf + (gsel * NB) * 256 + row16 * 16;
#pragma unroll
      for (int cc = 0; cc < MAXCC; ++cc) {
        if (cc * CH < kc) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            frag b = *reinterpret_cast<const frag*>(wbase + (cc * 4 * NB + nb) * 256);
#pragma unroll
            for (int g = 0; g < G; ++g) mma(acc[g][nb], __builtin_bit_cast(frag, a[s][g][cc]), b);
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | (TAIL & 15) | ((TAIL >> 4) << 14));       // slab(p+1) landed
    __syncthreads();
  };
  for (int p = 0; p < nphase; p += R) {
    phase(std::integral_constant<int, 0>{}, p);
    if (p + 1 < nphase) phase(std::integral_constant<int, 1>{}, p + 1);
    if constexpr (R > 2) { if (p + 2 < nphase) phase(std::integral_constant<int, 2 % R>{}, p + 2); }
    if constexpr (R > 3) { if (p + 3 < nphase) phase(std::integral_constant<int, 3 % R>{}, p + 3); }
  }

  store_tile<T, NB, G, NWAVES>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu,
                               ep_res, false, 0, tile_stats, bx, &bnb);
}

// ------------------------------------------------------------------------------------------
// the lean kernel: same dataflow, a quarter of the scalar instructions
// ------------------------------------------------------------------------------------------
// PMC and ablations on the generic kernel above (profiles/README.md): its waves issue ~130 scalar
// instructions and 22 s_waitcnt per phase around 18 MFMAs, and with 16-24 waves resident the ONE
// scalar unit of a CU saturates -- a build with no gathers, no weight traffic and no MFMAs still
// took 45 % of the full run time.  This form serves the common case (whole reduction slices: ci a
// multiple of the slice; one 16-row group per wave; fewer than 2^24 input rows) with everything
// that does not change inside a tile hoisted out of the phase loop:
//   * no live/dead flags: the loop body always prefetches a phase that exists, the last phase is
//     peeled; no "channels left in this slice" tests (whole slices);
//   * a wave moves CONSECUTIVE 1-KiB pieces of a slab, so one M0 write and immediate offsets serve
//     its whole share of the DMA; waves without a share skip the DMA and the wait for it;
//   * gather addresses: one 24-bit multiply per row, the steps of a slice by immediate offsets;
//   * one walk state (offset, slice) advanced once per phase.
// NWAVES = 8 (128-row tiles) or 16 (256-row tiles: half the weight traffic per row at the same
// number of resident waves, for layers whose slab is large against their gathers).
// Tried on this kernel and dropped (each measured, profiles/README.md): gathers two phases ahead
// (three register sets / weight slots: 98 vs 94 us on the roofline layer -- the gathers run at the
// ~5.5 TB/s this access pattern gets from the Infinity Cache whatever the depth), persistent
// workgroups with the pipeline running through tile boundaries (117 us: the write-out of a tile
// then sits on every wave's critical path instead of overlapping another workgroup's phases),
// gathers with the four lanes of a quad on the 64 contiguous bytes of a row step and a ds_bpermute
// into the MFMA lane layout (the addresser merges a quad into one request: the loads alone got
// 8-18 % cheaper in a timing probe, but the twelve bpermutes per phase cost more: 99 vs 94 us).
// ---- instrument (scripts/build_variant.py ... -DLIDAL_PHASE_STAMPS; never in the product library): where a phase of
// the lean kernel spends its time.  Every wave reads the shader clock (s_memtime) at six points of every phase --
//   T0 phase start | index(p+2), slab(p+1) DMA, A(p+1) gathers issued (includes the wait for index(p+1)) T1 |
//   A(p) landed T2 | fragment reads + MFMAs issued T3 | slab(p+1) landed T4 | barrier passed T5
// -- keeps the five interval sums in scalar registers and leaves, per (workgroup, wave), 12 words:
//   phases, whole kernel (clock), prologue, sum issue, sum wait-A, sum compute, sum slab wait, sum barrier, epilogue,
//   whole kernel (s_memrealtime, 100 MHz), 0, 0.   scripts/exp/phase_stamps.py prints the histogram.
#ifdef LIDAL_PHASE_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define LIDAL_NOW() __builtin_readcyclecounter()
#endif

template <typename T, int NB, int ROW_BYTES, int NWAVES, bool DENSE>
__global__ void __launch_bounds__(64 * NWAVES, (NWAVES == 8 ? LEAN_MINWAVES : 4))
conv_lean_kernel(const T* __restrict__ in, const T* __restrict__ wimg, const int* __restrict__ nbr,
                 const int* __restrict__ perm, const unsigned* __restrict__ tmasks,
                 T* __restrict__ out, int64_t n_out, int ci, int co, int K, int kflip,
                 const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int ep_relu,
                 const T* __restrict__ ep_res, unsigned in_bytes, unsigned img_bytes,
                 unsigned nbr_bytes, float* __restrict__ tile_stats, BnBwd bnb, Split sp) {
  constexpr int BM = NWAVES * 16;
  constexpr int BN = 16 * NB;
  constexpr int CH = DT<T>::CH;
  constexpr int KC = ROW_BYTES / (int)sizeof(T);
  constexpr int MAXCC = KC / CH;                         // MFMA reduction steps (64 bytes) per slice
  constexpr int SLAB = BN * ROW_BYTES;
  constexpr int PIECES = SLAB / 1024;
  constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;    // consecutive pieces per DMA wave
  constexpr int DMA_WAVES = PIECES / PPW;                // waves that move a share
  static_assert(PIECES % PPW == 0 && DMA_WAVES <= NWAVES, "a DMA wave moves a whole share");
  static_assert(ROW_BYTES % 64 == 0 && SLAB % 1024 == 0, "slices are whole MFMA steps / DMA pieces");
  static_assert(BM % 128 == 0 || BM == 64, "tile masks are per 128 rows (a 64-row tile takes its parent's)");
  typedef typename DT<T>::frag frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                              // [2][SLAB], re-used as the epilogue tile
  constexpr int LEPI_ALL = NWAVES * 16 * (BN + DT<T>::VEC) * (int)sizeof(T) + NWAVES * BN * 2 * (int)sizeof(float);
  unsigned char* const dump = smem + ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL);     // 4 KiB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row16 = lane & 15;
  const int gsel = lane >> 4;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * 16;
  const int n0 = blockIdx.y * BN;
  const int npass = ci / KC;

  unsigned tmask = 1u;
  if constexpr (!DENSE) {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
#pragma unroll
    for (int h = 0; h < (BM >= 128 ? BM / 128 : 1); ++h)
      if ((t0 + h) * 128 < n_out) m |= tmasks[t0 + h];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = __builtin_amdgcn_readfirstlane(m);
    if (sp.nsplit > 1) {        // this workgroup's share of the tile's active offsets: ranks [lo, hi) of the set bits
      const int cnt = __popc(tmask), z = (int)blockIdx.z;
      const int lo = z * cnt / sp.nsplit, hi = (z + 1) * cnt / sp.nsplit;
      unsigned left = tmask, sub = 0u;
      for (int rk = 0; left != 0u; ++rk) {
        const unsigned bit = left & (0u - left);
        if (rk >= lo && rk < hi) sub |= bit;
        left ^= bit;
      }
      tmask = __builtin_amdgcn_readfirstlane(sub);
    }
  }
  const int nphase = __popc(tmask) * npass;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);

  // ---- invariants of the tile
  constexpr unsigned OOB_OFF = 0x80000000u;
  const unsigned row_bytes = (unsigned)ci * (unsigned)sizeof(T);
  const unsigned lane_off = (unsigned)gsel * 16u;               // this lane's 16 bytes of a 64-byte step
  const unsigned idx_voff = (unsigned)((r0 + row16) * 4);       // this row inside one offset's table row
  const bool row_in = r0 + row16 < n_out;
  const unsigned k_stride = (unsigned)n_out * 4u;               // bytes of one offset's table row
  const unsigned slab_k = (unsigned)gridDim.y * (unsigned)npass * (unsigned)SLAB;      // slabs of one offset
  const unsigned slab_base = (unsigned)blockIdx.y * (unsigned)npass * (unsigned)SLAB
                             + (unsigned)(wave * PPW) * 1024u;                          // + this wave's share
  const unsigned dma_voff = (unsigned)lane * 16u;
  const bool dma_wave = wave < DMA_WAVES;
  unsigned char* const dma_dst = wl + (wave * PPW) * 1024;
  const unsigned char* const wbase = wl + (gsel * NB) * 256 + row16 * 16;

  // this lane's entry of the row permutation, requested now and used by the write-out at the end
  // (unconditional: without a permutation, and past the last row, the range check returns zeros)
  const __amdgpu_buffer_rsrc_t rs_perm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(perm), 0, perm != nullptr ? (int)k_stride : 0, 0x00020000);
  const int perm_v = __builtin_amdgcn_raw_buffer_load_b32(rs_perm, idx_voff, 0, 0);

  // the walk over (active offset, slice): one scalar state, advanced once per phase
  unsigned rem = tmask;
  int wk = 0, wpass = npass - 1;
  auto advance = [&]() {
    if (++wpass == npass) {
      wpass = 0;
      wk = rem ? __builtin_ctz(rem) : 0;            // past the end: offset 0 (a harmless index load)
      rem &= rem - 1u;
    }
  };
  auto issue_idx = [&](int k) -> int {
    if constexpr (DENSE) {
      return (int)(r0 + row16);
    } else {
      const unsigned kk = (unsigned)(kflip ? (K - 1 - k) : k);
      return __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, idx_voff, kk * k_stride, 0);
    }
  };
  // EVERY wave issues PPW DMA instructions per phase -- a wave without a share of the slab aims out of range
  // (zeros, no memory traffic) at the dump: hipcc counts vector-memory operations statically, and behind a
  // wave-dependent branch it could not count these -- its wait for a phase's A fragments then also covered the
  // index load and the first DMA pieces issued IN that phase: one exposed memory round trip per phase
  // (0.55 us from L2 at stride 16, 1.2 us beyond it at stride 8; profiles/README.md, round 3)
  auto issue_dma = [&](int k, int pass, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const unsigned soff = dma_wave ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
    unsigned char* const base = dma_wave ? dma_dst + slot * SLAB : dump;
    static_assert(PPW <= 12, "DMA share");
    // four 1-KiB pieces per M0 value (the immediate offset field ends at 4095)
#pragma unroll
    for (int c4 = 0; c4 < (PPW + 3) / 4; ++c4) {
      auto* dst = (__attribute__((address_space(3))) void*)(base + (dma_wave ? c4 * 4096 : 0));
      const unsigned so = dma_wave ? soff + (unsigned)(c4 * 4096) : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 0, 0);
      if (c4 * 4 + 1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 1024, 0);
      if (c4 * 4 + 2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 2048, 0);
      if (c4 * 4 + 3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 3072, 0);
    }
  };
  // A fragments of (neighbour rows idx, slice pass); returns the ballot of rows that have a rule.
  // Lanes of rows without one aim out of range: zeros, no memory access.
  auto issue_a = [&](raw4 (&a)[MAXCC], int idx, int pass) __attribute__((always_inline)) -> unsigned long long {
    const bool has = idx >= 0 && row_in;
    unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
    off = has ? off : OOB_OFF;
    const unsigned soff = (unsigned)pass * (unsigned)ROW_BYTES;
#pragma unroll
    for (int cc = 0; cc < MAXCC; ++cc)
      a[cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + (unsigned)(cc * 64), soff, 0));
    return __ballot(has);
  };

  f32x4 acc[1][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[0][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const raw4 (&a)[MAXCC], unsigned long long have, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    if (have != 0ull) {       // a wave none of whose 16 rows has a rule skips
#ifdef LIDAL_LEAN_BB
      constexpr int BB = (NB % LIDAL_LEAN_BB == 0) ? LIDAL_LEAN_BB : NB;
#else
      constexpr int BB = NB;          // fragment reads issued as one batch
#endif
#pragma unroll
      for (int cc = 0; cc < MAXCC; ++cc) {
#pragma unroll
        for (int nb0 = 0; nb0 < NB; nb0 += BB) {
          frag b[BB];
#pragma unroll
          for (int j = 0; j < BB; ++j)
            b[j] = *reinterpret_cast<const frag*>(wbase + slot * SLAB + (cc * 4 * NB + nb0 + j) * 256);
#pragma unroll
          for (int j = 0; j < BB; ++j) mma(acc[0][nb0 + j], __builtin_bit_cast(frag, a[cc]), b[j]);
        }
      }
    }
  };
  // the next slab has landed: of this wave's loads only the MAXCC gathers issued behind its DMA may
  // still be in flight (waves without a DMA share have nothing to wait for)
  auto slab_wait = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (MAXCC & 15) | ((MAXCC >> 4) << 14));
    __syncthreads();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

#if defined(LIDAL_LEAN_ROLLING) && !defined(LIDAL_PHASE_STAMPS)
  // (experiment build, round 6) ONE rolling set of A fragments and one copy of the loop body, as conv_lean32_kernel: the
  // registers of reduction step cc are re-loaded with the next phase's rows right behind this phase's MFMAs of that
  // step; the tile's last phase issues its loads out of range.  Same sums in the same order as the two-set pipeline.
  {
    auto dma_rt = [&](int k, int pass, int slot, bool live_phase) {
      const bool live = dma_wave && live_phase;
      const unsigned soff = live ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
      unsigned char* const base = live ? dma_dst + slot * SLAB : dump;
#pragma unroll
      for (int c4 = 0; c4 < (PPW + 3) / 4; ++c4) {
        auto* dst = (__attribute__((address_space(3))) void*)(base + (live ? c4 * 4096 : 0));
        const unsigned so = live ? soff + (unsigned)(c4 * 4096) : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 0, 0);
        if (c4 * 4 + 1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 1024, 0);
        if (c4 * 4 + 2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 2048, 0);
        if (c4 * 4 + 3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 3072, 0);
      }
    };
    auto a_off = [&](int idx, unsigned long long& have) -> unsigned {
      const bool has = idx >= 0 && row_in;
      have = __ballot(has);
      const unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
      return has ? off : OOB_OFF;
    };
    raw4 a[MAXCC];
    if (nphase > 0) {
      advance();
      const int k0 = wk, p0 = wpass;
      const int i0 = issue_idx(k0);
      advance();
      int i_nxt = issue_idx(wk);
      int kn = wk, pn = wpass;
      unsigned long long have = 0ull;
      dma_rt(k0, p0, 0, true);
      {
        const unsigned off = a_off(i0, have);
#pragma unroll
        for (int cc = 0; cc < MAXCC; ++cc)
          a[cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + (unsigned)(cc * 64), (unsigned)p0 * (unsigned)ROW_BYTES, 0));
      }
      slab_wait();
      int slot = 0;
      for (int p = 0; p < nphase; ++p) {
        const bool more = p + 1 < nphase;
        advance();
        const int i_nn = issue_idx(wk);
        dma_rt(kn, pn, slot ^ 1, more);
        unsigned long long have_n;
        unsigned off = a_off(i_nxt, have_n);
        off = more ? off : OOB_OFF;
        have_n = more ? have_n : 0ull;
        const unsigned char* const wcur = wbase + slot * SLAB;
        const unsigned soff = (unsigned)pn * (unsigned)ROW_BYTES;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cc = 0; cc < MAXCC; ++cc) {
          if (have != 0ull) {
            frag b[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[j] = *reinterpret_cast<const frag*>(wcur + (cc * 4 * NB + j) * 256);
#pragma unroll
            for (int j = 0; j < NB; ++j) mma(acc[0][j], __builtin_bit_cast(frag, a[cc]), b[j]);
          }
          __builtin_amdgcn_sched_barrier(0);
          a[cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + (unsigned)(cc * 64), soff, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        have = have_n;
        i_nxt = i_nn; kn = wk; pn = wpass;
        slot ^= 1;
        slab_wait();
      }
    }
  }
#else
  // Pipeline: phase p issues index(p+2), slab(p+1) by DMA, A(p+1) -- in this order: vector-memory
  // operations retire in order, so the gather's wait for index(p+1) (issued a phase ago, FIRST)
  // leaves last phase's slab and gathers in flight, and slab_wait leaves this phase's gathers.
  raw4 a0[MAXCC], a1[MAXCC];
  unsigned long long h0 = 0ull, h1 = 0ull;
#ifdef LIDAL_PHASE_STAMPS
  const unsigned long long st_begin = LIDAL_NOW(), st_rbegin = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_t0 = st_begin, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_t4 = 0, st_first = st_begin, st_last = st_begin;
  unsigned long long st_issue = 0, st_waita = 0, st_comp = 0, st_slab = 0, st_bar = 0;
  constexpr int ST_AFTER = (DENSE ? 0 : 1) + PPW + MAXCC;       // loads issued behind A(p): index(p+2), slab(p+1), A(p+1)
#define ST_ISSUED() st_t1 = LIDAL_NOW(); __builtin_amdgcn_s_waitcnt(0x0F70 | (ST_AFTER & 15) | ((ST_AFTER >> 4) << 14)); st_t2 = LIDAL_NOW()
#define ST_COMPUTED() st_t3 = LIDAL_NOW(); __builtin_amdgcn_s_waitcnt(0x0F70 | (MAXCC & 15) | ((MAXCC >> 4) << 14)); st_t4 = LIDAL_NOW()
#define ST_PHASE_END() { const unsigned long long t5 = LIDAL_NOW(); st_issue += st_t1 - st_t0; st_waita += st_t2 - st_t1; \
    st_comp += st_t3 - st_t2; st_slab += st_t4 - st_t3; st_bar += t5 - st_t4; st_t0 = t5; st_last = t5; }
#else
#define ST_ISSUED()
#define ST_COMPUTED()
#define ST_PHASE_END()
#endif
  if (nphase > 0) {
    advance();
    const int k0 = wk, p0 = wpass;
    int i0 = issue_idx(k0);
    advance();
    int i1 = issue_idx(wk);
    int kn = wk, pn = wpass;                        // (offset, slice) of the phase issued next
    issue_dma(k0, p0, S0{});
    h0 = issue_a(a0, i0, p0);
    slab_wait();
#ifdef LIDAL_PHASE_STAMPS
    st_t0 = st_first = st_last = LIDAL_NOW();
#endif
    int p = 0;
    while (true) {
      if (p + 1 >= nphase) { compute(a0, h0, S0{}); break; }
      advance();
      i0 = issue_idx(wk);
      issue_dma(kn, pn, S1{});
      h1 = issue_a(a1, i1, pn);
      kn = wk; pn = wpass;
      ST_ISSUED();
      compute(a0, h0, S0{});
      ST_COMPUTED();
      slab_wait();
      ST_PHASE_END();
      ++p;
      if (p + 1 >= nphase) { compute(a1, h1, S1{}); break; }
      advance();
      i1 = issue_idx(wk);
      issue_dma(kn, pn, S0{});
      h0 = issue_a(a0, i0, pn);
      kn = wk; pn = wpass;
      ST_ISSUED();
      compute(a1, h1, S1{});
      ST_COMPUTED();
      slab_wait();
      ST_PHASE_END();
      ++p;
    }
    __syncthreads();                                // every wave is done with the last slab
  }
#endif
#undef ST_ISSUED
#undef ST_COMPUTED
#undef ST_PHASE_END
  if (sp.nsplit > 1) {          // a share of the tile's offsets: the f32 accumulators go to the combining kernel
    f32x4* dst = reinterpret_cast<f32x4*>(sp.partial) +
                 ((((int64_t)blockIdx.z * gridDim.x + bx) * gridDim.y + blockIdx.y) * NWAVES + wave) * (NB * 64) + lane;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dst[nb * 64] = acc[0][nb];
    return;
  }
  store_tile<T, NB, 1, NWAVES>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu,
                               ep_res, true, perm_v, tile_stats, bx, &bnb);
#ifdef LIDAL_PHASE_STAMPS
  if (g_stamp_buf != nullptr) {
    __builtin_amdgcn_s_waitcnt(0);                  // the tile's stores have left
    const unsigned long long st_end = LIDAL_NOW(), st_rend = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* d = g_stamp_buf + ((((size_t)blockIdx.y * gridDim.x + bx) * NWAVES) + wave) * 12;
      d[0] = (unsigned long long)nphase; d[1] = st_end - st_begin; d[2] = st_first - st_begin; d[3] = st_issue; d[4] = st_waita;
      d[5] = st_comp; d[6] = st_slab; d[7] = st_bar; d[8] = st_end - st_last; d[9] = st_rend - st_rbegin;
      d[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                // HW_ID (wave, simd, cu, sh, se ...)
              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);       // XCC_ID
      d[11] = st_rbegin;
    }
  }
#endif
}

// The shares of a split launch (struct Split) added in split order, then the epilogue of the unsplit kernel.
template <typename T, int NB, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES)
conv_combine_kernel(Split sp, const int* __restrict__ perm, T* __restrict__ out, int64_t n_out, int co,
                    const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int ep_relu,
                    const T* __restrict__ ep_res, float* __restrict__ tile_stats, BnBwd bnb) {
  constexpr int BM = NWAVES * 16, BN = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row16 = lane & 15, gsel = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * BM + wave * 16;
  const int n0 = blockIdx.y * BN;
  f32x4 acc[1][NB];
  const f32x4* src = reinterpret_cast<const f32x4*>(sp.partial) +
                     (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NWAVES + wave) * (NB * 64) + lane;
  const int64_t share = (int64_t)gridDim.x * gridDim.y * NWAVES * (NB * 64);      // vectors per split
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f32x4 v = src[nb * 64];
    for (int z = 1; z < sp.nsplit; ++z) {
      const f32x4 t = src[(int64_t)z * share + nb * 64];
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    acc[0][nb] = v;
  }
  const int perm_v = (perm != nullptr && r0 + row16 < n_out) ? perm[r0 + row16] : 0;
  store_tile<T, NB, 1, NWAVES>(acc, smem, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu,
                               ep_res, true, perm_v, tile_stats, -1, &bnb);
}



// ------------------------------------------------------------------------------------------
// the deep form of the lean kernel: slabs and gathers TWO phases ahead (rings of three)
// ------------------------------------------------------------------------------------------
// On the coarse levels a launch is as long as its heaviest tile: a 128-row tile that touches all 27
// offsets of a 256-channel layer runs 27 x 4 slices = 108 phases back to back, ~1.2 us each
// (profiles/README.md, round 3) -- and a phase computes for only ~0.4 us: the rest is the latency of
// the next slab (the 3.5 MB weight image of such a layer does not stay in the 4 MB L2 next to the
// gathered rows) that ONE phase of look-ahead does not cover, with too few workgroups per CU to
// hide it (1-2.6).  Here phase p issues index(p+3), slab(p+2) and A(p+2); same accumulation order
// (offsets ascending, slices ascending), hence bitwise the lean kernel's results.  Used where the
// rows are few (launch_img); on the fine levels -- gather-throughput bound, 3 resident workgroups --
// two phases of look-ahead measured slower (round 2).
template <typename T, int NB, int ROW_BYTES, int NWAVES, bool DENSE>
__global__ void __launch_bounds__(64 * NWAVES, LEAN_MINWAVES)
conv_lean_deep_kernel(const T* __restrict__ in, const T* __restrict__ wimg, const int* __restrict__ nbr,
                      const int* __restrict__ perm, const unsigned* __restrict__ tmasks,
                      T* __restrict__ out, int64_t n_out, int ci, int co, int K, int kflip,
                      const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int ep_relu,
                      const T* __restrict__ ep_res, unsigned in_bytes, unsigned img_bytes,
                      unsigned nbr_bytes, float* __restrict__ tile_stats, BnBwd bnb) {
  constexpr int BM = NWAVES * 16;
  constexpr int BN = 16 * NB;
  constexpr int CH = DT<T>::CH;
  constexpr int KC = ROW_BYTES / (int)sizeof(T);
  constexpr int MAXCC = KC / CH;
  constexpr int SLAB = BN * ROW_BYTES;
  constexpr int PIECES = SLAB / 1024;
  constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;
  constexpr int DMA_WAVES = PIECES / PPW;
  constexpr int IL = DENSE ? 0 : 1;
  static_assert(PIECES % PPW == 0 && DMA_WAVES <= NWAVES, "a DMA wave moves a whole share");
  static_assert(ROW_BYTES % 64 == 0 && SLAB % 1024 == 0, "slices are whole MFMA steps / DMA pieces");
  static_assert(BM == TILE_ROWS, "tile masks are per 128 rows");
  typedef typename DT<T>::frag frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                              // [3][SLAB], re-used as the epilogue tile
  constexpr int EPI = NWAVES * 16 * (BN + DT<T>::VEC) * (int)sizeof(T) + NWAVES * BN * 2 * (int)sizeof(float);
  constexpr int WREGION = (3 * SLAB > EPI) ? 3 * SLAB : EPI;
  unsigned char* dump = smem + WREGION;                  // 4 KiB: where the DMA of a phase past the end lands

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row16 = lane & 15;
  const int gsel = lane >> 4;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * 16;
  const int n0 = blockIdx.y * BN;
  const int npass = ci / KC;

  unsigned tmask = 1u;
  if constexpr (!DENSE) {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
    if (t0 * 128 < n_out) m = tmasks[t0];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = __builtin_amdgcn_readfirstlane(m);
  }
  const int nphase = __popc(tmask) * npass;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);

  constexpr unsigned OOB_OFF = 0x80000000u;
  const unsigned row_bytes = (unsigned)ci * (unsigned)sizeof(T);
  const unsigned lane_off = (unsigned)gsel * 16u;
  const unsigned idx_voff = (unsigned)((r0 + row16) * 4);
  const bool row_in = r0 + row16 < n_out;
  const unsigned k_stride = (unsigned)n_out * 4u;
  const unsigned slab_k = (unsigned)gridDim.y * (unsigned)npass * (unsigned)SLAB;
  const unsigned slab_base = (unsigned)blockIdx.y * (unsigned)npass * (unsigned)SLAB + (unsigned)(wave * PPW) * 1024u;
  const unsigned dma_voff = (unsigned)lane * 16u;
  const bool dma_wave = wave < DMA_WAVES;
  unsigned char* const dma_dst = wl + (wave * PPW) * 1024;
  const unsigned char* const wbase = wl + (gsel * NB) * 256 + row16 * 16;

  const __amdgpu_buffer_rsrc_t rs_perm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(perm), 0, perm != nullptr ? (int)k_stride : 0, 0x00020000);
  const int perm_v = __builtin_amdgcn_raw_buffer_load_b32(rs_perm, idx_voff, 0, 0);

  unsigned rem = tmask;
  int wk = 0, wpass = npass - 1;
  auto advance = [&]() {
    if (++wpass == npass) {
      wpass = 0;
      wk = rem ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
    }
  };
  auto issue_idx = [&](int k) -> int {
    if constexpr (DENSE) {
      return (int)(r0 + row16);
    } else {
      const unsigned kk = (unsigned)(kflip ? (K - 1 - k) : k);
      return __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, idx_voff, kk * k_stride, 0);
    }
  };
  // every DMA wave issues its PPW pieces in EVERY phase (a phase past the end reads out of range -- zeros,
  // no memory traffic -- into the dump): the counted waits below rely on it
  auto issue_dma = [&](int k, int pass, auto slot_c, bool live_phase) {
    constexpr int slot = decltype(slot_c)::value;
    const bool live = live_phase && dma_wave;           // (every wave issues, see the lean kernel)
    const unsigned soff = live ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
    unsigned char* base = live ? dma_dst + slot * SLAB : dump;
    static_assert(PPW <= 12, "DMA share");
#pragma unroll
    for (int c4 = 0; c4 < (PPW + 3) / 4; ++c4) {
      auto* dst = (__attribute__((address_space(3))) void*)(base + (live ? c4 * 4096 : 0));
      const unsigned so = live ? soff + (unsigned)(c4 * 4096) : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 0, 0);
      if (c4 * 4 + 1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 1024, 0);
      if (c4 * 4 + 2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 2048, 0);
      if (c4 * 4 + 3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 3072, 0);
    }
  };
  auto issue_a = [&](raw4 (&a)[MAXCC], int idx, int pass, bool live) __attribute__((always_inline)) -> unsigned long long {
    const bool has = idx >= 0 && row_in && live;
    unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
    off = has ? off : OOB_OFF;
    const unsigned soff = (unsigned)pass * (unsigned)ROW_BYTES;
#pragma unroll
    for (int cc = 0; cc < MAXCC; ++cc)
      a[cc] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + (unsigned)(cc * 64), soff, 0));
    return __ballot(has);
  };

  f32x4 acc[1][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[0][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const raw4 (&a)[MAXCC], unsigned long long have, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    if (have != 0ull) {
#pragma unroll
      for (int cc = 0; cc < MAXCC; ++cc) {
        frag b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
          b[j] = *reinterpret_cast<const frag*>(wbase + slot * SLAB + (cc * 4 * NB + j) * 256);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma(acc[0][j], __builtin_bit_cast(frag, a[cc]), b[j]);
      }
    }
  };
  // slab(p+1) has landed: behind it only A(p+1), and this phase's index load, slab(p+2) share and A(p+2)
  constexpr int TAIL = 2 * MAXCC + IL + PPW;
  auto slab_wait = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (TAIL & 15) | ((TAIL >> 4) << 14));
    __syncthreads();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;

  raw4 a0[MAXCC], a1[MAXCC], a2[MAXCC];
  unsigned long long h0 = 0ull, h1 = 0ull, h2 = 0ull;
  if (nphase > 0) {
    advance();
    int k0 = wk, p0 = wpass, i0 = issue_idx(k0);
    advance();
    int k1 = wk, p1 = wpass, i1 = issue_idx(k1);
    advance();
    int k2 = wk, p2 = wpass, i2 = issue_idx(k2);
    issue_dma(k0, p0, S0{}, true);
    h0 = issue_a(a0, i0, p0, true);
    issue_dma(k1, p1, S1{}, 1 < nphase);
    h1 = issue_a(a1, i1, p1, 1 < nphase);
    // slab(0) has landed: behind it A(0), slab(1), A(1)
    {
      constexpr int T0 = 2 * MAXCC + PPW;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (T0 & 15) | ((T0 >> 4) << 14));
    }
    __syncthreads();
    int p = 0;
    while (true) {
      // phase p in slot 0: index(p+3) -> (k0, i0), slab(p+2) / A(p+2) -> slot 2
      advance(); k0 = wk; p0 = wpass; i0 = issue_idx(k0);
      issue_dma(k2, p2, S2{}, p + 2 < nphase);
      h2 = issue_a(a2, i2, p2, p + 2 < nphase);
      compute(a0, h0, S0{});
      slab_wait();
      if (++p >= nphase) break;
      // phase p in slot 1: index(p+3) -> (k1, i1), slab(p+2) / A(p+2) -> slot 0
      advance(); k1 = wk; p1 = wpass; i1 = issue_idx(k1);
      issue_dma(k0, p0, S0{}, p + 2 < nphase);
      h0 = issue_a(a0, i0, p0, p + 2 < nphase);
      compute(a1, h1, S1{});
      slab_wait();
      if (++p >= nphase) break;
      // phase p in slot 2: index(p+3) -> (k2, i2), slab(p+2) / A(p+2) -> slot 1
      advance(); k2 = wk; p2 = wpass; i2 = issue_idx(k2);
      issue_dma(k1, p1, S1{}, p + 2 < nphase);
      h1 = issue_a(a1, i1, p1, p + 2 < nphase);
      compute(a2, h2, S2{});
      slab_wait();
      if (++p >= nphase) break;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);               // the dump writes of the phases past the end are done
  __syncthreads();
  store_tile<T, NB, 1, NWAVES>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu,
                               ep_res, true, perm_v, tile_stats, bx, &bnb);
}


// ------------------------------------------------------------------------------------------
// f32 features, f32 weights, products on the bf16 MFMA: the split form (round 5)
// ------------------------------------------------------------------------------------------
// The f32 mode (the reference's precision: score/prob_inference.py infers in fp32) ran on v_mfma_f32_16x16x4_f32 --
// 157 TFLOP/s dense against 2.5 PFLOP/s for bf16 -- and its convolutions were MFMA-bound (66 % of that peak on the
// 96 -> 96 layer, 80 % of a scored frame's kernel time).  Here an f32 operand v is cut into THREE bf16 pieces by
// truncation, v = hi + mid + lo EXACTLY (8 + 8 + 8 significand bits; hi = top 16 bits of v, mid = top 16 bits of
// v - hi, lo = v - hi - mid, each difference exact in f32), and a product a.b is the sum of the six partial
// products of weight <= 2 (a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_mid b_mid + a_lo b_hi; every one exact
// in the MFMA's f32 accumulator: 8 x 8 bits), added smallest first.  Dropped: a_mid b_lo + a_lo b_mid + a_lo b_lo
// <= 3 * 2^-24 |a b| -- the size of ONE f32 rounding of the product -- so the result carries the error of an f32
// dot product whose products were each rounded once more (tests/test_ops_gpu.py bounds it against f64: 1e-6 of the
// output scale, where the exact-f32 kernel sits at 2e-7).  Six bf16 MFMAs (96 cycles) replace the eight f32 MFMAs
// (256 cycles) of a 16 x 16 x 32 block.
//   * weights: lidal_conv_weight_image with dtype LIDAL_F32_SPLIT lays the three pieces of a 32-channel slice side by
//     side as ONE 96-"channel" bf16 slice (hi | mid | lo) in the bf16 image layout with 192-byte rows, so slabs, the
//     LDS-DMA and the conflict-free fragment reads are the bf16 96-channel kernel's;
//   * features: a lane gathers the 8 consecutive f32 channels its A fragment covers (two 16-byte loads), cuts them in
//     registers (4 VALU operations per element + 3 packs per pair) and feeds three bf16x8 fragments;
//   * the rest -- pattern-sorted rows, tile masks, scalar offset walk, one phase of look-ahead, epilogue -- is the lean
//     kernel's.  One phase = (active offset, 32-channel slice): 6 NB MFMAs per wave.
// Used for inference (no_grad) and wherever the caller passes LIDAL_F32_SPLIT; the training step's f32 parity mode
// keeps the exact f32 MFMA (its golden gradients are pinned to that association).
struct Pieces { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Pieces cut3(const raw4& lo4, const raw4& hi4) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned u = (unsigned)(e < 4 ? lo4[e] : hi4[e - 4]);
    const float v = __uint_as_float(u);
    h[e] = u & 0xFFFF0000u;
    const float r = v - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r) & 0xFFFF0000u;
    l[e] = __float_as_uint(r - __uint_as_float(m[e]));
  }
  raw4 ph, pm, pl;
#pragma unroll
  for (int j = 0; j < 4; ++j) {       // bf16 element 2j in the low half: bytes {3,2} of the odd element | {3,2} of the even one
    ph[j] = (int)__builtin_amdgcn_perm(h[2 * j + 1], h[2 * j], 0x07060302u);
    pm[j] = (int)__builtin_amdgcn_perm(m[2 * j + 1], m[2 * j], 0x07060302u);
    pl[j] = (int)__builtin_amdgcn_perm(l[2 * j + 1], l[2 * j], 0x07060302u);
  }
  Pieces p;
  p.hi = __builtin_bit_cast(bf16x8, ph); p.mid = __builtin_bit_cast(bf16x8, pm); p.lo = __builtin_bit_cast(bf16x8, pl);
  return p;
}

constexpr int SPLIT_KCH = 32;                  // f32 channels per slice
constexpr int SPLIT_ROW = 3 * SPLIT_KCH * 2;   // bytes of a slice's image row: hi | mid | lo in bf16 = 192

template <int NB, int NWAVES, bool DENSE>
__global__ void __launch_bounds__(64 * NWAVES, LEAN_MINWAVES)
conv_split_kernel(const float* __restrict__ in, const __bf16* __restrict__ wimg, const int* __restrict__ nbr,
                  const int* __restrict__ perm, const unsigned* __restrict__ tmasks, float* __restrict__ out,
                  int64_t n_out, int ci, int co, int K, int kflip, const float* __restrict__ ep_scale,
                  const float* __restrict__ ep_shift, int ep_relu, const float* __restrict__ ep_res, unsigned in_bytes,
                  unsigned img_bytes, unsigned nbr_bytes, Split sp) {
  constexpr int BM = NWAVES * 16;
  constexpr int BN = 16 * NB;
  constexpr int SLAB = BN * SPLIT_ROW;
  constexpr int PIECES = SLAB / 1024;
  constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;
  constexpr int DMA_WAVES = PIECES / PPW;
  static_assert(PIECES % PPW == 0 && DMA_WAVES <= NWAVES && SLAB % 1024 == 0, "a DMA wave moves a whole share");
  static_assert(BM == TILE_ROWS, "tile masks are per 128 rows");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                              // [2][SLAB], re-used as the epilogue tile
  constexpr int LEPI_ALL = NWAVES * 16 * (BN + 4) * 4 + NWAVES * BN * 2 * (int)sizeof(float);
  unsigned char* const dump = smem + ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL);     // 4 KiB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row16 = lane & 15;
  const int gsel = lane >> 4;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * 16;
  const int n0 = blockIdx.y * BN;
  const int npass = ci / SPLIT_KCH;

  unsigned tmask = 1u;
  if constexpr (!DENSE) {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
    if (t0 * 128 < n_out) m = tmasks[t0];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = __builtin_amdgcn_readfirstlane(m);
    if (sp.nsplit > 1) {        // this workgroup's share of the tile's active offsets (struct Split): ranks [lo, hi) of the set bits
      const int cnt = __popc(tmask), z = (int)blockIdx.z;
      const int lo = z * cnt / sp.nsplit, hi = (z + 1) * cnt / sp.nsplit;
      unsigned left = tmask, sub = 0u;
      for (int rk = 0; left != 0u; ++rk) {
        const unsigned bit = left & (0u - left);
        if (rk >= lo && rk < hi) sub |= bit;
        left ^= bit;
      }
      tmask = __builtin_amdgcn_readfirstlane(sub);
    }
  }
  const int nphase = __popc(tmask) * npass;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);

  constexpr unsigned OOB_OFF = 0x80000000u;
  const unsigned row_bytes = (unsigned)ci * 4u;
  const unsigned lane_off = (unsigned)gsel * 32u;               // this lane's 8 f32 channels of a 32-channel slice
  const unsigned idx_voff = (unsigned)((r0 + row16) * 4);
  const bool row_in = r0 + row16 < n_out;
  const unsigned k_stride = (unsigned)n_out * 4u;
  const unsigned slab_k = (unsigned)gridDim.y * (unsigned)npass * (unsigned)SLAB;
  const unsigned slab_base = (unsigned)blockIdx.y * (unsigned)npass * (unsigned)SLAB + (unsigned)(wave * PPW) * 1024u;
  const unsigned dma_voff = (unsigned)lane * 16u;
  const bool dma_wave = wave < DMA_WAVES;
  unsigned char* const dma_dst = wl + (wave * PPW) * 1024;
  const unsigned char* const wbase = wl + (gsel * NB) * 256 + row16 * 16;

  const __amdgpu_buffer_rsrc_t rs_perm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(perm), 0, perm != nullptr ? (int)k_stride : 0, 0x00020000);
  const int perm_v = __builtin_amdgcn_raw_buffer_load_b32(rs_perm, idx_voff, 0, 0);

  unsigned rem = tmask;
  int wk = 0, wpass = npass - 1;
  auto advance = [&]() {
    if (++wpass == npass) {
      wpass = 0;
      wk = rem ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
    }
  };
  auto issue_idx = [&](int k) -> int {
    if constexpr (DENSE) {
      return (int)(r0 + row16);
    } else {
      const unsigned kk = (unsigned)(kflip ? (K - 1 - k) : k);
      return __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, idx_voff, kk * k_stride, 0);
    }
  };
  auto issue_dma = [&](int k, int pass, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const unsigned soff = dma_wave ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
    unsigned char* const base = dma_wave ? dma_dst + slot * SLAB : dump;
    static_assert(PPW <= 4, "DMA share: one M0 value");
    auto* dst = (__attribute__((address_space(3))) void*)base;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, soff, 0, 0);
    if (1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, soff, 1024, 0);
    if (2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, soff, 2048, 0);
    if (3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, soff, 3072, 0);
  };
  auto issue_a = [&](raw4 (&a)[2], int idx, int pass) __attribute__((always_inline)) -> unsigned long long {
    const bool has = idx >= 0 && row_in;
    unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
    off = has ? off : OOB_OFF;
    const unsigned soff = (unsigned)pass * (unsigned)(SPLIT_KCH * 4);
    a[0] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, soff, 0));
    a[1] = __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 16u, soff, 0));
    return __ballot(has);
  };

  f32x4 acc[1][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[0][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const raw4 (&a)[2], unsigned long long have, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    if (have != 0ull) {
      const Pieces p = cut3(a[0], a[1]);
#ifdef LIDAL_SPLIT_BB
      constexpr int BB = (NB % LIDAL_SPLIT_BB == 0) ? LIDAL_SPLIT_BB : 2;
#else
      constexpr int BB = 2;             // column blocks whose three fragments are read as one batch
#endif
#pragma unroll
      for (int nb0 = 0; nb0 < NB; nb0 += BB) {
        bf16x8 bh[BB], bm[BB], bl[BB];
#pragma unroll
        for (int j = 0; j < BB; ++j) {
          const unsigned char* q = wbase + slot * SLAB + (nb0 + j) * 256;
          bh[j] = *reinterpret_cast<const bf16x8*>(q);
          bm[j] = *reinterpret_cast<const bf16x8*>(q + 4 * NB * 256);
          bl[j] = *reinterpret_cast<const bf16x8*>(q + 8 * NB * 256);
        }
#pragma unroll
        for (int j = 0; j < BB; ++j) {      // smallest partial products first
          f32x4 c = acc[0][nb0 + j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.lo, bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.mid, bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.hi, bl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.mid, bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.hi, bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.hi, bh[j], c, 0, 0, 0);
          acc[0][nb0 + j] = c;
        }
      }
    }
  };
  auto slab_wait = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | 2);             // only the two gathers issued behind the DMA may be in flight
    __syncthreads();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  raw4 a0[2], a1[2];
  unsigned long long h0 = 0ull, h1 = 0ull;
  if (nphase > 0) {
    advance();
    const int k0 = wk, p0 = wpass;
    int i0 = issue_idx(k0);
    advance();
    int i1 = issue_idx(wk);
    int kn = wk, pn = wpass;
    issue_dma(k0, p0, S0{});
    h0 = issue_a(a0, i0, p0);
    slab_wait();
    int p = 0;
    while (true) {
      if (p + 1 >= nphase) { compute(a0, h0, S0{}); break; }
      advance();
      i0 = issue_idx(wk);
      issue_dma(kn, pn, S1{});
      h1 = issue_a(a1, i1, pn);
      kn = wk; pn = wpass;
      compute(a0, h0, S0{});
      slab_wait();
      ++p;
      if (p + 1 >= nphase) { compute(a1, h1, S1{}); break; }
      advance();
      i1 = issue_idx(wk);
      issue_dma(kn, pn, S0{});
      h0 = issue_a(a0, i0, pn);
      kn = wk; pn = wpass;
      compute(a1, h1, S1{});
      slab_wait();
      ++p;
    }
    __syncthreads();
  }
  if (sp.nsplit > 1) {          // a share of the tile's offsets: the f32 accumulators go to conv_combine_kernel
    f32x4* dst = reinterpret_cast<f32x4*>(sp.partial) +
                 ((((int64_t)blockIdx.z * gridDim.x + bx) * gridDim.y + blockIdx.y) * NWAVES + wave) * (NB * 64) + lane;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dst[nb * 64] = acc[0][nb];
    return;
  }
  store_tile<float, NB, 1, NWAVES>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu,
                                   ep_res, true, perm_v, nullptr, bx, nullptr);
}

// ------------------------------------------------------------------------------------------
// the split form on v_mfma_f32_32x32x16_bf16: 32 rows per wave (round 6)
// ------------------------------------------------------------------------------------------
// conv_split_kernel reads one 1-KiB B fragment from LDS per 16 x 16 x 32 MFMA (16 cycles): 3 NB fragment reads for 6 NB
// MFMAs per wave and slice, the CU's LDS array and its four MFMA pipes loaded about equally -- it ran at 53 % of its MFMA
// bound.  Here a wave owns 32 rows and multiplies 32 x 32 x 16 blocks (32 cycles each): a B fragment (16 channels x 32
// columns, still one ds_read_b128 = 1 KiB per wave) now serves twice the rows, so a 128-row tile reads half the fragment
// bytes for the same MFMA cycles, and has 4 waves instead of 8 (half the DMA and barrier participants per row).
//   * the weight image is conv_split_kernel's, byte for byte: lane l = (r = l & 31, h = l >> 5) of a 32 x 32 x 16 B operand
//     holds B[k = 8 h + j][column r]; with k-step ks of a 32-channel slice that is the image's 8-channel group
//     gsel = 2 ks + h and its column (2 cb + (r >> 4)) * 16 + (r & 15): the lane's address is base + r * 16 inside a
//     contiguous 512-byte run -- conflict free for ds_read_b128's four 16-lane groups as before;
//   * features: lane (r, h) gathers channels 16 ks + 8 h .. + 7 of row r for both k-steps (four 16-byte loads per phase
//     -- the bytes per row are the same 128), cuts them into the three bf16 pieces in registers;
//   * accumulators: NB / 2 blocks of 16 registers (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h); the
//     write-out goes through a wave-private LDS tile 16 rows at a time (store_rows32).
// Same products as conv_split_kernel (six partial products per f32 product, smallest first within a k-step); the sums run
// in another order (16 channels per MFMA instead of 32), so results agree with it to f32 rounding of the accumulation,
// not bit for bit (tests/test_ops_gpu.py bounds both against f64).
typedef __attribute__((ext_vector_type(16))) float f32x16;
#ifndef LIDAL_SPLIT32_MINWAVES
#define LIDAL_SPLIT32_MINWAVES 2
#endif
__host__ __device__ constexpr int split32_ppw(int pieces, int nwaves) {        // consecutive 1-KiB pieces per DMA wave: equal whole shares
  int p = (pieces + nwaves - 1) / nwaves;
  while (pieces % p != 0) ++p;
  return p;
}

// accumulators of the 32-row form -> HBM, 16 rows at a time through a wave-private LDS tile [16][BN + VEC] of T, whole
// rows with 16-byte stores, with the affine map / ReLU / residual of the inference epilogue.  perm_v: lane l < 32 holds
// perm[r0 + l].
template <typename T, int NB2>
__device__ __forceinline__ void store_rows32(f32x16 (&acc)[NB2], unsigned char* wl, int wave, int lane, int64_t r0,
                                             int n0, int64_t n_out, int co, const int* __restrict__ perm,
                                             T* __restrict__ out, const float* __restrict__ ep_scale,
                                             const float* __restrict__ ep_shift, int ep_relu,
                                             const T* __restrict__ ep_res, int perm_v) {
  constexpr int VEC = DT<T>::VEC;
  constexpr int BN = 32 * NB2;
  constexpr int ESTRIDE = BN + VEC;
  typedef typename DT<T>::frag frag;
  const int r32 = lane & 31, h = lane >> 5;
  T* et = reinterpret_cast<T*>(wl) + wave * 16 * ESTRIDE;
  if (ep_scale != nullptr) {
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb) {
      const int col = n0 + cb * 32 + r32;
      const float es = col < co ? ep_scale[col] : 1.f, eh = col < co ? ep_shift[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[cb][r] * es + eh;
        acc[cb][r] = ((ep_relu & 1) && v < 0.f) ? 0.f : v;
      }
    }
  }
  constexpr int RSEGS = BN / VEC;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // rows 16 half .. + 15 of the wave: register quads 2 half and 2 half + 1
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          et[(8 * q + 4 * h + i) * ESTRIDE + cb * 32 + r32] = DT<T>::from_f32(acc[cb][(2 * half + q) * 4 + i]);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = lane; i < 16 * RSEGS; i += 64) {
      const int r = i / RSEGS, cseg = (i - r * RSEGS) * VEC;
      const int prow = __shfl(perm_v, 16 * half + r, 64);           // all lanes take part
      const int64_t trow = r0 + 16 * half + r;
      if (trow >= n_out) continue;
      const int64_t row = perm ? (int64_t)prow : trow;
      T* dst = out + row * co + n0 + cseg;
      const T* srcp = et + r * ESTRIDE + cseg;
      if (n0 + cseg + VEC <= co) {
        frag v = *reinterpret_cast<const frag*>(srcp);
        if (ep_res != nullptr) {
          const frag rr = *reinterpret_cast<const frag*>(ep_res + row * co + n0 + cseg);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float f = DT<T>::to_f32(v[e]) + DT<T>::to_f32(rr[e]);
            if ((ep_relu & 2) && f < 0.f) f = 0.f;
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
    __builtin_amdgcn_s_waitcnt(0xC07F);           // the tile is read before the second half overwrites it
  }
}

template <int NB, int NWAVES, bool DENSE>
__global__ void __launch_bounds__(64 * NWAVES, (NWAVES == 8 ? 4 : LIDAL_SPLIT32_MINWAVES))
conv_split32_kernel(const float* __restrict__ in, const __bf16* __restrict__ wimg, const int* __restrict__ nbr,
                    const int* __restrict__ perm, const unsigned* __restrict__ tmasks, float* __restrict__ out,
                    int64_t n_out, int ci, int co, int K, int kflip, const float* __restrict__ ep_scale,
                    const float* __restrict__ ep_shift, int ep_relu, const float* __restrict__ ep_res, unsigned in_bytes,
                    unsigned img_bytes, unsigned nbr_bytes, Split sp) {
  constexpr int BM = NWAVES * 32;                        // 128- or 256-row tiles
  constexpr int BN = 16 * NB;
  constexpr int NB2 = NB / 2;
  constexpr int SLAB = BN * SPLIT_ROW;
  constexpr int PIECES = SLAB / 1024;
  constexpr int PPW = split32_ppw(PIECES, NWAVES);
  constexpr int DMA_WAVES = PIECES / PPW;
  static_assert(NB % 2 == 0, "32-column blocks");
  static_assert(PIECES % PPW == 0 && DMA_WAVES <= NWAVES && SLAB % 1024 == 0 && PPW <= 12, "a DMA wave moves a whole share");
  static_assert(BM % TILE_ROWS == 0, "tile masks are per 128 rows");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                              // [2][SLAB], re-used as the epilogue tile
  constexpr int LEPI_ALL = NWAVES * 16 * (BN + 4) * 4;
  unsigned char* const dump = smem + ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL);     // 4 KiB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31;
  const int h = lane >> 5;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * 32;
  const int n0 = blockIdx.y * BN;
  const int npass = ci / SPLIT_KCH;

  unsigned tmask = 1u;
  if constexpr (!DENSE) {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
#pragma unroll
    for (int q = 0; q < BM / 128; ++q)
      if ((t0 + q) * 128 < n_out) m |= tmasks[t0 + q];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = __builtin_amdgcn_readfirstlane(m);
    if (sp.nsplit > 1) {        // this workgroup's share of the tile's active offsets (struct Split)
      const int cnt = __popc(tmask), z = (int)blockIdx.z;
      const int lo = z * cnt / sp.nsplit, hi = (z + 1) * cnt / sp.nsplit;
      unsigned left = tmask, sub = 0u;
      for (int rk = 0; left != 0u; ++rk) {
        const unsigned bit = left & (0u - left);
        if (rk >= lo && rk < hi) sub |= bit;
        left ^= bit;
      }
      tmask = __builtin_amdgcn_readfirstlane(sub);
    }
  }
  const int nphase = __popc(tmask) * npass;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);

  constexpr unsigned OOB_OFF = 0x80000000u;
  const unsigned row_bytes = (unsigned)ci * 4u;
  const unsigned lane_off = (unsigned)h * 32u;                  // this lane's 8 f32 channels of a 16-channel k-step
  const unsigned idx_voff = (unsigned)((r0 + r32) * 4);
  const bool row_in = r0 + r32 < n_out;
  const unsigned k_stride = (unsigned)n_out * 4u;
  const unsigned slab_k = (unsigned)gridDim.y * (unsigned)npass * (unsigned)SLAB;
  const unsigned slab_base = (unsigned)blockIdx.y * (unsigned)npass * (unsigned)SLAB + (unsigned)(wave * PPW) * 1024u;
  const unsigned dma_voff = (unsigned)lane * 16u;
  const bool dma_wave = wave < DMA_WAVES;
  unsigned char* const dma_dst = wl + (wave * PPW) * 1024;
  const unsigned char* const wbase = wl + (h * NB) * 256 + r32 * 16;

  const __amdgpu_buffer_rsrc_t rs_perm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(perm), 0, perm != nullptr ? (int)k_stride : 0, 0x00020000);
  const int perm_v = __builtin_amdgcn_raw_buffer_load_b32(rs_perm, idx_voff, 0, 0);

  unsigned rem = tmask;
  int wk = 0, wpass = npass - 1;
  auto advance = [&]() {
    if (++wpass == npass) {
      wpass = 0;
      wk = rem ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
    }
  };
  auto issue_idx = [&](int k) -> int {
    if constexpr (DENSE) {
      return (int)(r0 + r32);
    } else {
      const unsigned kk = (unsigned)(kflip ? (K - 1 - k) : k);
      return __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, idx_voff, kk * k_stride, 0);
    }
  };
  // every wave issues PPW DMA instructions per phase (a wave without a share aims out of range at the dump): the
  // counted wait below relies on it, as in the lean kernel
  auto issue_dma = [&](int k, int pass, int slot, bool live_phase) {
    const bool live = dma_wave && live_phase;
    const unsigned soff = live ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
    unsigned char* const base = live ? dma_dst + slot * SLAB : dump;
#pragma unroll
    for (int c4 = 0; c4 < (PPW + 3) / 4; ++c4) {
      auto* dst = (__attribute__((address_space(3))) void*)(base + (live ? c4 * 4096 : 0));
      const unsigned so = live ? soff + (unsigned)(c4 * 4096) : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 0, 0);
      if (c4 * 4 + 1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 1024, 0);
      if (c4 * 4 + 2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 2048, 0);
      if (c4 * 4 + 3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 3072, 0);
    }
  };
  // a[2 ks + t]: channels 16 ks + 8 h + 4 t .. + 3 of the lane's row.  ONE rolling set (as conv_lean32_kernel): the two
  // registers of a k-step are cut into their bf16 pieces, re-loaded at once with the NEXT phase's row, and only then
  // multiplied -- the phase's gathers sit between its MFMAs, not in a burst behind the barrier.
  auto a_off = [&](int idx, unsigned long long& have) -> unsigned {
    const bool has = idx >= 0 && row_in;
    have = __ballot(has);
    const unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
    return has ? off : OOB_OFF;
  };
  auto load_a = [&](unsigned off, int pass, int j) -> raw4 {
    return __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(
        rs_in, off + (unsigned)((j >> 1) * 64 + (j & 1) * 16), (unsigned)pass * (unsigned)(SPLIT_KCH * 4), 0));
  };

  f32x16 acc[NB2];
#pragma unroll
  for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

  raw4 a[4];
  auto mfma_step = [&](const Pieces& p, int ks, const unsigned char* wcur) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb) {
      const unsigned char* q = wcur + ks * (2 * NB * 256) + cb * 512;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(q + 4 * NB * 256);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(q + 8 * NB * 256);
      f32x16 c = acc[cb];                 // smallest partial products first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.lo, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.mid, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.hi, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.mid, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.hi, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.hi, bh, c, 0, 0, 0);
      acc[cb] = c;
    }
  };
  auto slab_wait = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);             // only the four gathers issued behind the DMA may be in flight
    __syncthreads();
  };

  if (nphase > 0) {
    advance();
    const int k0 = wk, p0 = wpass;
    const int i0 = issue_idx(k0);
    advance();
    int i_nxt = issue_idx(wk);                      // rows of phase 1
    int kn = wk, pn = wpass;                        // (offset, slice) of phase 1
    unsigned long long have = 0ull;
    issue_dma(k0, p0, 0, true);
    {
      const unsigned off = a_off(i0, have);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = load_a(off, p0, j);
    }
    slab_wait();
    int slot = 0;
    for (int p = 0; p < nphase; ++p) {              // (the tile's last phase issues its loads out of range: one loop body)
      const bool more = p + 1 < nphase;
      advance();
      const int i_nn = issue_idx(wk);
      issue_dma(kn, pn, slot ^ 1, more);
      unsigned long long have_n;
      unsigned off = a_off(i_nxt, have_n);
      off = more ? off : OOB_OFF;
      have_n = more ? have_n : 0ull;
      const unsigned char* const wcur = wbase + slot * SLAB;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Pieces pc;
        if (have != 0ull) pc = cut3(a[2 * ks], a[2 * ks + 1]);
        __builtin_amdgcn_sched_barrier(0);
        a[2 * ks] = load_a(off, pn, 2 * ks);
        a[2 * ks + 1] = load_a(off, pn, 2 * ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (have != 0ull) mfma_step(pc, ks, wcur);
        __builtin_amdgcn_sched_barrier(0);
      }
      have = have_n;
      i_nxt = i_nn; kn = wk; pn = wpass;
      slot ^= 1;
      slab_wait();
    }
  }
  if (sp.nsplit > 1) {          // a share of the tile's offsets: the f32 accumulators go to conv_combine32_kernel
    f32x4* dst = reinterpret_cast<f32x4*>(sp.partial) +
                 ((((int64_t)blockIdx.z * gridDim.x + bx) * gridDim.y + blockIdx.y) * NWAVES + wave) * (NB2 * 4 * 64) + lane;
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[(cb * 4 + q) * 64] = f32x4{acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
    return;
  }
  store_rows32<float, NB2>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu, ep_res, perm_v);
}

// the shares of a split launch of the 32-row form added in split order, then its epilogue
template <typename T, int NB2, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES)
conv_combine32_kernel(Split sp, const int* __restrict__ perm, T* __restrict__ out, int64_t n_out, int co,
                      const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int ep_relu,
                      const T* __restrict__ ep_res) {
  constexpr int BM = NWAVES * 32, BN = 32 * NB2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * BM + wave * 32;
  const int n0 = blockIdx.y * BN;
  f32x16 acc[NB2];
  const f32x4* src = reinterpret_cast<const f32x4*>(sp.partial) +
                     (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NWAVES + wave) * (NB2 * 4 * 64) + lane;
  const int64_t share = (int64_t)gridDim.x * gridDim.y * NWAVES * (NB2 * 4 * 64);      // vectors per split
#pragma unroll
  for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = src[(cb * 4 + q) * 64];
      for (int z = 1; z < sp.nsplit; ++z) {
        const f32x4 t = src[(int64_t)z * share + (cb * 4 + q) * 64];
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
      }
      acc[cb][4 * q] = v[0]; acc[cb][4 * q + 1] = v[1]; acc[cb][4 * q + 2] = v[2]; acc[cb][4 * q + 3] = v[3];
    }
  const int64_t prow_i = r0 + (lane & 31);
  const int perm_v = (perm != nullptr && prow_i < n_out) ? perm[prow_i] : 0;
  store_rows32<T, NB2>(acc, smem, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu, ep_res, perm_v);
}

// ------------------------------------------------------------------------------------------
// the lean kernel on v_mfma_f32_32x32x16_bf16 with 256-row tiles (round 6)
// ------------------------------------------------------------------------------------------
// What the lean kernel moves through a CU's vector-memory path per launch of the roofline layer (96 -> 96, 396 662 rows,
// 1.88 M rules): 361 MB of gathered rows -- and 1.29 GB of WEIGHT SLABS (23.9 k (tile, offset) pairs x 3 slices x 18 KB,
// L2 -> LDS by DMA): the slab of a phase serves only 128 rows.  256-row tiles of 16 waves x 16 rows halved that but left one
// workgroup per CU (94 registers x 16 waves) and lost (round 5).  With 32 rows per wave a 256-row tile is 8 waves: the same
// waves and rows in flight per CU as today (two workgroups, <= 128 registers), half the slab bytes, half the DMA
// instructions, half the LDS fragment reads and half the barrier participants per row.
//   * A operand: lane (r = l & 31, h = l >> 5) holds channels 16 ks + 8 h .. + 7 of row r for k-step ks: one 16-byte load
//     per k-step (ROW_BYTES / 32 per phase), the same bytes per row as before;
//   * B operand: the bf16 image as it is -- k-step ks is the image's 32-channel step cc = ks / 2, 8-channel group
//     gsel = 2 (ks & 1) + h; a lane reads base + r * 16 inside a contiguous 512-byte run: conflict free;
//   * a tile's offset mask is the OR of its two 128-row masks (the row order keeps neighbouring tiles' patterns close).
// Same sums in another order (16 channels per MFMA instead of 32): equal to the lean kernel to f32 rounding of the
// accumulation, bitwise reproducible run to run.
#ifndef LIDAL_LEAN32_MINWAVES
#define LIDAL_LEAN32_MINWAVES 4
#endif
template <int NB, int ROW_BYTES, bool DENSE>
__global__ void __launch_bounds__(512, LIDAL_LEAN32_MINWAVES)
conv_lean32_kernel(const __bf16* __restrict__ in, const __bf16* __restrict__ wimg, const int* __restrict__ nbr,
                   const int* __restrict__ perm, const unsigned* __restrict__ tmasks, __bf16* __restrict__ out,
                   int64_t n_out, int ci, int co, int K, int kflip, const float* __restrict__ ep_scale,
                   const float* __restrict__ ep_shift, int ep_relu, const __bf16* __restrict__ ep_res, unsigned in_bytes,
                   unsigned img_bytes, unsigned nbr_bytes) {
  typedef __bf16 T;
  constexpr int NWAVES = 8;
  constexpr int BM = NWAVES * 32;                        // 256 rows
  constexpr int BN = 16 * NB;
  constexpr int NB2 = NB / 2;
  constexpr int KS = ROW_BYTES / 32;                     // 16-channel k-steps per slice
  constexpr int SLAB = BN * ROW_BYTES;
  constexpr int PIECES = SLAB / 1024;
  constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;
  constexpr int DMA_WAVES = PIECES / PPW;
  static_assert(NB % 2 == 0 && ROW_BYTES % 64 == 0, "32-column blocks, whole 32-channel image steps");
  static_assert(PIECES % PPW == 0 && DMA_WAVES <= NWAVES && SLAB % 1024 == 0 && PPW <= 12, "a DMA wave moves a whole share");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                              // [2][SLAB], re-used as the epilogue tile
  constexpr int LEPI_ALL = NWAVES * 16 * (BN + 8) * 2;
  unsigned char* const dump = smem + ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL);     // 4 KiB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31;
  const int h = lane >> 5;
  const int bx = tile_of_block();
  const int64_t r0 = (int64_t)bx * BM + wave * 32;
  const int n0 = blockIdx.y * BN;
  const int npass = ci / (ROW_BYTES / 2);

  unsigned tmask = 1u;
  if constexpr (!DENSE) {
    unsigned m = 0u;
    const int64_t t0 = ((int64_t)bx * BM) >> 7;
#pragma unroll
    for (int q = 0; q < BM / 128; ++q)
      if ((t0 + q) * 128 < n_out) m |= tmasks[t0 + q];
    if (kflip) m = __brev(m) >> (32 - K);
    tmask = __builtin_amdgcn_readfirstlane(m);
  }
  const int nphase = __popc(tmask) * npass;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wimg), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nbr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(nbr), 0, (int)nbr_bytes, 0x00020000);

  constexpr unsigned OOB_OFF = 0x80000000u;
  const unsigned row_bytes = (unsigned)ci * 2u;
  const unsigned lane_off = (unsigned)h * 16u;                  // this lane's 8 channels of a 16-channel k-step
  const unsigned idx_voff = (unsigned)((r0 + r32) * 4);
  const bool row_in = r0 + r32 < n_out;
  const unsigned k_stride = (unsigned)n_out * 4u;
  const unsigned slab_k = (unsigned)gridDim.y * (unsigned)npass * (unsigned)SLAB;
  const unsigned slab_base = (unsigned)blockIdx.y * (unsigned)npass * (unsigned)SLAB + (unsigned)(wave * PPW) * 1024u;
  const unsigned dma_voff = (unsigned)lane * 16u;
  const bool dma_wave = wave < DMA_WAVES;
  unsigned char* const dma_dst = wl + (wave * PPW) * 1024;
  const unsigned char* const wbase = wl + (h * NB) * 256 + r32 * 16;

  const __amdgpu_buffer_rsrc_t rs_perm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(perm), 0, perm != nullptr ? (int)k_stride : 0, 0x00020000);
  const int perm_v = __builtin_amdgcn_raw_buffer_load_b32(rs_perm, idx_voff, 0, 0);

  unsigned rem = tmask;
  int wk = 0, wpass = npass - 1;
  auto advance = [&]() {
    if (++wpass == npass) {
      wpass = 0;
      wk = rem ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
    }
  };
  auto issue_idx = [&](int k) -> int {
    if constexpr (DENSE) {
      return (int)(r0 + r32);
    } else {
      const unsigned kk = (unsigned)(kflip ? (K - 1 - k) : k);
      return __builtin_amdgcn_raw_buffer_load_b32(rs_nbr, idx_voff, kk * k_stride, 0);
    }
  };
  // every wave issues PPW DMA instructions in EVERY phase (the counted wait relies on it): a wave without a share, and
  // every wave in the tile's last phase (nothing follows), aims out of range (zeros, no memory traffic) at the dump
  auto issue_dma = [&](int k, int pass, int slot, bool live_phase) {
    const bool live = dma_wave && live_phase;
    const unsigned soff = live ? (unsigned)k * slab_k + (unsigned)pass * (unsigned)SLAB + slab_base : OOB_OFF;
    unsigned char* const base = live ? dma_dst + slot * SLAB : dump;
#pragma unroll
    for (int c4 = 0; c4 < (PPW + 3) / 4; ++c4) {
      auto* dst = (__attribute__((address_space(3))) void*)(base + (live ? c4 * 4096 : 0));
      const unsigned so = live ? soff + (unsigned)(c4 * 4096) : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 0, 0);
      if (c4 * 4 + 1 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 1024, 0);
      if (c4 * 4 + 2 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 2048, 0);
      if (c4 * 4 + 3 < PPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, dma_voff, so, 3072, 0);
    }
  };
  // ONE set of A fragments, rolling: the registers of k-step ks are re-loaded with the NEXT phase's rows as soon as this
  // phase's MFMAs of that k-step have been issued -- half the fragment registers of the two-set pipeline (which spilled
  // here: 48 accumulators + 2 x 24), and the phase's vector-memory instructions are spread between its MFMAs instead of
  // being issued as one burst by all waves right behind the barrier.
  auto a_off = [&](int idx, unsigned long long& have) -> unsigned {
    const bool has = idx >= 0 && row_in;
    have = __ballot(has);
    const unsigned off = __umul24((unsigned)idx, row_bytes) + lane_off;
    return has ? off : OOB_OFF;
  };
  auto load_a = [&](unsigned off, int pass, int ks) -> raw4 {
    return __builtin_bit_cast(raw4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + (unsigned)(ks * 32),
                                                                          (unsigned)pass * (unsigned)ROW_BYTES, 0));
  };

  f32x16 acc[NB2];
#pragma unroll
  for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

  raw4 a[KS];
  auto mfma_step = [&](int ks, const unsigned char* wcur) __attribute__((always_inline)) {
    const unsigned char* q = wcur + (ks >> 1) * (4 * NB * 256) + (ks & 1) * (2 * NB * 256);
    bf16x8 b[NB2];
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb) b[cb] = *reinterpret_cast<const bf16x8*>(q + cb * 512);
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb)
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks]), b[cb], acc[cb], 0, 0, 0);
  };
  auto slab_wait = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (KS & 15) | ((KS >> 4) << 14));    // only the KS gathers issued behind the DMA
    __syncthreads();
  };

  if (nphase > 0) {
    advance();
    const int k0 = wk, p0 = wpass;
    const int i0 = issue_idx(k0);
    advance();
    int i_nxt = issue_idx(wk);                      // rows of phase 1
    int kn = wk, pn = wpass;                        // (offset, slice) of phase 1
    unsigned long long have = 0ull;
    issue_dma(k0, p0, 0, true);
    {
      const unsigned off = a_off(i0, have);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = load_a(off, p0, ks);
    }
    slab_wait();
    // one phase on slab `slot`: index(p+2), slab(p+1) by DMA into the other slot, then per k-step this phase's MFMAs and
    // the gather of the next phase's fragment into the registers just consumed.  The tile's last phase issues the same
    // instructions out of range (no peeled copy of the loop body: one set of accumulators, one copy of the code).
    int slot = 0;
    for (int p = 0; p < nphase; ++p) {
      const bool more = p + 1 < nphase;
      advance();
      const int i_nn = issue_idx(wk);
      issue_dma(kn, pn, slot ^ 1, more);
      unsigned long long have_n;
      unsigned off = a_off(i_nxt, have_n);
      off = more ? off : OOB_OFF;
      have_n = more ? have_n : 0ull;
      const unsigned char* const wcur = wbase + slot * SLAB;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (have != 0ull) mfma_step(ks, wcur);
        __builtin_amdgcn_sched_barrier(0);
        a[ks] = load_a(off, pn, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
      have = have_n;
      i_nxt = i_nn; kn = wk; pn = wpass;
      slot ^= 1;
      slab_wait();
    }
  }
  store_rows32<T, NB2>(acc, wl, wave, lane, r0, n0, n_out, co, perm, out, ep_scale, ep_shift, ep_relu, ep_res, perm_v);
}

// one 16-byte segment of a split image: image[k][nblk][slice][part][gsel][nb][row16] x 8 bf16, part = hi | mid | lo of
// the f32 weight W[k][slice * 32 + gsel * 8 + e][col] (role 0) / W[k][col][...] (role 1)
template <typename TI>
__device__ __forceinline__ void split_segment(const TI* __restrict__ w, __bf16* __restrict__ img, int n_red, int n_col,
                                              int role, int nb, int64_t s) {
  const int bn = 16 * nb;
  const int nblk = (n_col + bn - 1) / bn, npass = n_red / SPLIT_KCH;
  unsigned r = (unsigned)s;
  const int row16 = (int)(r & 15u); r >>= 4;
  const int b = (int)(r % (unsigned)nb); r /= (unsigned)nb;
  const int gsel = (int)(r & 3u); r >>= 2;
  const int part = (int)(r % 3u); r /= 3u;
  const int pass = (int)(r % (unsigned)npass); r /= (unsigned)npass;
  const int blk = (int)(r % (unsigned)nblk); r /= (unsigned)nblk;
  const int k = (int)r;
  const int col = blk * bn + b * 16 + row16;
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int red = pass * SPLIT_KCH + gsel * 8 + e;
    float f = 0.f;
    if (col < n_col) {
      const int64_t base = (int64_t)k * n_red * n_col;
      f = DT<TI>::to_f32(role == 0 ? w[base + (int64_t)red * n_col + col] : w[base + (int64_t)col * n_red + red]);
    }
    const unsigned u = __float_as_uint(f);
    const unsigned hb = u & 0xFFFF0000u;
    const float rr = f - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(rr) & 0xFFFF0000u;
    const unsigned lb = __float_as_uint(rr - __uint_as_float(mb));
    v[e] = (unsigned short)((part == 0 ? hb : (part == 1 ? mb : lb)) >> 16);
  }
  *reinterpret_cast<raw4*>(reinterpret_cast<unsigned short*>(img) + s * 8) = *reinterpret_cast<raw4*>(v);
}
template <typename TI>
__global__ void __launch_bounds__(256)
weight_image_split_kernel(const TI* __restrict__ w, __bf16* __restrict__ img, int n_red, int n_col, int role, int nb,
                          int64_t segs) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < segs) split_segment<TI>(w, img, n_red, n_col, role, nb, s);
}
// the forward images of several parameters in one launch (the jobs of lidal_conv_weight_image_job with LIDAL_F32_SPLIT)
template <typename TI>
__global__ void __launch_bounds__(256)
weight_image_batch_split_kernel(const ImageJob* __restrict__ jobs, int n_jobs, int64_t total) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= s) lo = mid; else hi = mid - 1;
  }
  const ImageJob j = jobs[lo];
  const int64_t l = s - j.first;
  if (l < j.segs_a) split_segment<TI>((const TI*)j.w, (__bf16*)j.img_a, j.n_red, j.n_col, j.role, j.nb_a, l);
  else if (l < j.segs_a + j.segs_b)       // (training, round 6: the data-gradient image of the same parameter)
    split_segment<TI>((const TI*)j.w, (__bf16*)j.img_b, j.n_col, j.n_red, j.role ^ 1, j.nb_b, l - j.segs_a);
}

struct Epi { const float* scale; const float* shift; int relu; const void* res; unsigned in_bytes, img_bytes, nbr_bytes; float* tile_stats; BnBwd bnb; void* ws; long long ws_bytes; };

// Split policy (struct Split): only launches that leave the chip mostly idle -- at most SPLIT_MAX_WGS workgroups -- and
// whose tiles have a long chain (a 27-offset map, or an 8-offset map over several reduction slices)
// (measured with the coalesced partial tiles, 5-scan step: <= 256 workgroups 14.68 ms, <= 700 -- the 43 k-row level split
// in two instead of the deep kernel -- 15.06, <= 1700 15.55; one scan 6.35 / 6.54 / 6.56)
#ifndef LIDAL_SPLIT_MAX_WGS
#define LIDAL_SPLIT_MAX_WGS 256
#endif
constexpr int64_t SPLIT_MAX_WGS = LIDAL_SPLIT_MAX_WGS;
#ifndef LIDAL_SPLIT_F32_MAX_WGS
#define LIDAL_SPLIT_F32_MAX_WGS 512
#endif
// the same for the split form of the f32 product (conv_split_kernel).  Measured on an 8-view frame and the 5-scan batch
// (scripts/exp/split_check.py, limits 0 / 600 / 1536): 256 -> 256 at stride 16 (408 / 262 workgroups) 334.9 -> 261-265 us /
// 265 -> 183-185 us with the offsets split three ways; at stride 8 (1038-1350 workgroups) neutral to -4 %, at stride 4
// (1265 workgroups) 136 -> 185 us: a loss -- the f32 partial tiles cost more than the tail they remove
constexpr int64_t SPLIT_F32_MAX_WGS = LIDAL_SPLIT_F32_MAX_WGS;
__host__ inline int pick_split(int64_t wgs, int K, int npass) {
  if (wgs > SPLIT_MAX_WGS || K * npass < 27) return 1;
  return wgs <= 128 ? 4 : (wgs <= 192 ? 3 : 2);
}

constexpr int IMG_G = 1, IMG_NWAVES = 8, IMG_DEPTH = 1;       // generic kernel: row groups per wave, waves, pipeline depth

template <typename T, int NB, int ROW_BYTES>
int launch_img(const void* in, const void* wimg, const int* nbr, const int* perm, const unsigned* tmasks,
               void* out, int64_t n_out, int ci, int co, int K, int kflip, Epi ep, hipStream_t s) {
  constexpr int G = IMG_G, NWAVES = IMG_NWAVES;
  constexpr int NTHREADS = 64 * NWAVES, BM = NWAVES * G * 16, BN = 16 * NB;
  constexpr int SLAB = BN * ROW_BYTES;
  constexpr int EPI = NWAVES * G * 16 * (BN + DT<T>::VEC) * (int)sizeof(T);
  constexpr int D = IMG_DEPTH;
  constexpr int WREGION = ((D + 1) * SLAB > EPI) ? (D + 1) * SLAB : EPI;
  static_assert(BM == TILE_ROWS, "tile masks and BatchNorm statistics triples are per 128-row tile");
  if constexpr (G == 1 && NWAVES == 8) {
    if (ci % (ROW_BYTES / (int)sizeof(T)) == 0 && ep.in_bytes / ((unsigned)ci * sizeof(T)) < (1u << 24)) {
      // 256-row tiles (16 waves) where the weight slab outweighs the gathers of a 128-row tile
      constexpr int LW = LEAN_WAVES;
      constexpr int LBM = LW * 16;
#if LIDAL_LEAN_WAVES == 8
      static_assert(LBM == TILE_ROWS, "tile masks and BatchNorm statistics triples are per 128-row tile");
#else
      LIDAL_REQUIRE(ep.tile_stats == nullptr && ep.bnb.sums == nullptr, "experiment build (LIDAL_LEAN_WAVES): no tile statistics");
#endif
      // (experiment, LIDAL_LEAN32=1: the 32-row form with 256-row tiles where no BatchNorm statistics are asked for)
      if constexpr (sizeof(T) == 2 && NB % 2 == 0 && ROW_BYTES <= 192) {
        static const bool lean32 = [] { const char* e = getenv("LIDAL_LEAN32"); return e != nullptr && atoi(e) != 0; }();
        if (lean32 && ep.tile_stats == nullptr && ep.bnb.sums == nullptr) {
          constexpr int EPI32 = 8 * 16 * (BN + 8) * 2;
          constexpr int LDS32 = ((2 * SLAB > EPI32) ? 2 * SLAB : EPI32) + 4096;
          auto k32 = nbr ? conv_lean32_kernel<NB, ROW_BYTES, false> : conv_lean32_kernel<NB, ROW_BYTES, true>;
          static size_t attr32[2][MAX_DEVICES] = {};
          const int d32 = current_device();
          if (attr32[nbr ? 0 : 1][d32] < (size_t)LDS32) {
            LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, LDS32));
            attr32[nbr ? 0 : 1][d32] = LDS32;
          }
          dim3 g32((unsigned)cdiv(n_out, 256), (unsigned)cdiv(co, BN));
          k32<<<g32, 512, LDS32, s>>>((const __bf16*)in, (const __bf16*)wimg, nbr, perm, tmasks, (__bf16*)out, n_out, ci, co,
                                      K, kflip, ep.scale, ep.shift, ep.relu, (const __bf16*)ep.res, ep.in_bytes,
                                      ep.img_bytes, ep.nbr_bytes);
          LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(lean32)");
          return 0;
        }
      }
      constexpr int LEPI = LW * 16 * (BN + DT<T>::VEC) * (int)sizeof(T);
      constexpr int LSTATS = LW * BN * 2 * (int)sizeof(float);        // per-wave column statistics
      constexpr int LEAN_LDS = ((2 * SLAB > LEPI + LSTATS) ? 2 * SLAB : LEPI + LSTATS) + 4096;      // + the DMA dump
      // the 256-wide layers of the coarse levels (4+ slices per offset, few rows): two phases of look-ahead,
      // conv_lean_deep_kernel -- measured on every layer shape of the model (scripts/exp/deep_rows.py, bit-equal
      // everywhere): 256->256 on 43k rows 131 -> 112 us, 384->256 194 -> 163; neutral to 6 % slower on the
      // others (and on every shape of a single scan), which therefore keep the lean kernel
      if constexpr (sizeof(T) == 2 && NB == 8 && ROW_BYTES == 128 && LW == 8) {
#ifdef LIDAL_PHASE_STAMPS
       if (false) {           // (the instrumented build times the lean kernel on every level)
#else
       if (nbr != nullptr && n_out <= DEEP_MAX_ROWS && n_out >= 30000 && ci >= 256) {
#endif
        constexpr int DEPI = LEPI + LSTATS;
        constexpr int DEEP_LDS = ((3 * SLAB > DEPI) ? 3 * SLAB : DEPI) + 4096;
        auto dk = conv_lean_deep_kernel<T, NB, ROW_BYTES, LW, false>;
        static size_t deep_attr[MAX_DEVICES] = {};
        const int ddev = current_device();
        if (deep_attr[ddev] < (size_t)DEEP_LDS) {
          LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dk),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS));
          deep_attr[ddev] = DEEP_LDS;
        }
        dim3 dgrid((unsigned)cdiv(n_out, LBM), (unsigned)cdiv(co, BN));
        dk<<<dgrid, 64 * LW, DEEP_LDS, s>>>((const T*)in, (const T*)wimg, nbr, perm, tmasks, (T*)out, n_out,
                                            ci, co, K, kflip, ep.scale, ep.shift, ep.relu, (const T*)ep.res,
                                            ep.in_bytes, ep.img_bytes, ep.nbr_bytes, ep.tile_stats, ep.bnb);
        LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(deep)");
        return 0;
       }
      }
      auto lk = nbr ? conv_lean_kernel<T, NB, ROW_BYTES, LW, false>
                    : conv_lean_kernel<T, NB, ROW_BYTES, LW, true>;
      static size_t lean_attr[2][MAX_DEVICES] = {};
      const int ldev = current_device();
      if (lean_attr[nbr ? 0 : 1][ldev] < (size_t)LEAN_LDS) {
        LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lk),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_LDS));
        lean_attr[nbr ? 0 : 1][ldev] = LEAN_LDS;
      }
      dim3 lgrid((unsigned)cdiv(n_out, LBM), (unsigned)cdiv(co, BN));
      Split sp{nullptr, 1, 0, 0};
      // (bf16 only: the f32 parity mode keeps ONE f32 sum over the offsets per output element, the association its
      // golden gradients were taken with -- through 49 train-mode BatchNorm layers a last-bit change of a sum moves
      // end-to-end gradients by up to 1e-3)
      if (nbr != nullptr && ep.ws != nullptr && sizeof(T) == 2) {
        const int ns = pick_split((int64_t)lgrid.x * lgrid.y, K, ci / (ROW_BYTES / (int)sizeof(T)));
        const long long rows_pad = (long long)lgrid.x * LBM;
        const int co_pad = (int)lgrid.y * BN;
        if (ns > 1 && ep.ws_bytes >= (long long)ns * rows_pad * co_pad * 4)
          sp = Split{(float*)ep.ws, ns, co_pad, rows_pad};
      }
      lgrid.z = (unsigned)sp.nsplit;
      lk<<<lgrid, 64 * LW, LEAN_LDS, s>>>((const T*)in, (const T*)wimg, nbr, perm, tmasks, (T*)out, n_out,
                                          ci, co, K, kflip, ep.scale, ep.shift, ep.relu, (const T*)ep.res,
                                          ep.in_bytes, ep.img_bytes, ep.nbr_bytes, ep.tile_stats, ep.bnb, sp);
      LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(lean)");
      if (sp.nsplit > 1) {
        constexpr int COMB_LDS = LEPI + LSTATS;
        auto ck = conv_combine_kernel<T, NB, LW>;
        static size_t comb_attr[MAX_DEVICES] = {};
        if (comb_attr[ldev] < (size_t)COMB_LDS) {
          LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ck),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, COMB_LDS));
          comb_attr[ldev] = COMB_LDS;
        }
        lgrid.z = 1;
        ck<<<lgrid, 64 * LW, COMB_LDS, s>>>(sp, perm, (T*)out, n_out, co, ep.scale, ep.shift, ep.relu,
                                            (const T*)ep.res, ep.tile_stats, ep.bnb);
        LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(combine)");
      }
      return 0;
    }
  }
  const size_t lds = WREGION + 1024 + NWAVES * BN * 2 * sizeof(float);
  auto kern = nbr ? conv_apply_img_kernel<T, NB, ROW_BYTES, G, NWAVES, IMG_MINWAVES, false, D>
                  : conv_apply_img_kernel<T, NB, ROW_BYTES, G, NWAVES, IMG_MINWAVES, true, D>;
  static size_t attr_set[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[nbr ? 0 : 1][dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[nbr ? 0 : 1][dev] = lds;
  }
  dim3 grid((unsigned)cdiv(n_out, BM), (unsigned)cdiv(co, BN));
  kern<<<grid, NTHREADS, lds, s>>>((const T*)in, (const T*)wimg, nbr, perm, tmasks, (T*)out, n_out, ci,
                                   co, K, kflip, ep.scale, ep.shift, ep.relu, (const T*)ep.res,
                                   ep.in_bytes, ep.img_bytes, ep.nbr_bytes, ep.tile_stats, ep.bnb);
  LIDAL_CHECK_LAUNCH("lidal_conv_apply_image");
  return 0;
}

template <typename T>
int dispatch_img(Tiling t, const void* in, const void* wimg, const int* nbr, const int* perm,
                 const unsigned* tmasks, void* out, int64_t n_out, int ci, int co, int K, int kflip,
                 Epi ep, hipStream_t s) {
#define IMG_CASE(NBV, RB) \
  if (t.nb == NBV && t.row_bytes == RB) \
    return launch_img<T, NBV, RB>(in, wimg, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  IMG_CASE(2, 64) IMG_CASE(4, 64) IMG_CASE(6, 64) IMG_CASE(8, 64)
  IMG_CASE(2, 128) IMG_CASE(4, 128) IMG_CASE(6, 128) IMG_CASE(8, 128)
  IMG_CASE(2, 192) IMG_CASE(4, 192) IMG_CASE(6, 192) IMG_CASE(8, 192)
  IMG_CASE(2, 256) IMG_CASE(4, 256) IMG_CASE(6, 256) IMG_CASE(8, 256)
#undef IMG_CASE
  set_error("conv_apply_image: no kernel for tiling nb=%d row_bytes=%d", t.nb, t.row_bytes);
  return 2;
}


// ---- the split form: tiling and launch
__host__ inline int split_nb(int co) { return co <= 32 ? 2 : (co <= 64 ? 4 : ((co % 128 != 0 && (co % 96 == 0 || co < 128)) ? 6 : 8)); }
__host__ inline int64_t split_image_bytes(int k, int n_red, int n_col) {
  const int bn = 16 * split_nb(n_col);
  return (int64_t)k * ((n_col + bn - 1) / bn) * (n_red / SPLIT_KCH) * bn * SPLIT_ROW;
}
// which MFMA shape the split form multiplies with: 16 (conv_split_kernel) unless LIDAL_SPLIT_MFMA=32 asks for the 32-row form
// (conv_split32_kernel, round 6: built to halve the LDS fragment reads per MFMA -- measured equal on the 96-column layers and
// 8-30 % slower on the 128-column ones, profiles/r06_mfma32_ab.txt; it stays selectable and tested)
static int split_mfma_rows() {
  static const int rows = [] {
    const char* e = getenv("LIDAL_SPLIT_MFMA");
    return (e != nullptr && atoi(e) == 32) ? 32 : 16;
  }();
  return rows;
}

// rows per tile of the 32-row form: 256 (8 waves) or 128 (4 waves); LIDAL_SPLIT32_WAVES=4 / 8 forces one
// (128-column kernels: 138 registers -- three 4-wave workgroups per CU; the narrower ones fit the 128 of two 8-wave workgroups)
static int split32_waves(int nb) {
  static const int w = [] {
    const char* e = getenv("LIDAL_SPLIT32_WAVES");
    return e != nullptr ? atoi(e) : 0;
  }();
  return (w == 4 || w == 8) ? w : (nb >= 8 ? 4 : 8);
}

template <int NB, int LW>
int launch_split32(const void* in, const void* wimg, const int* nbr, const int* perm, const unsigned* tmasks, void* out,
                   int64_t n_out, int ci, int co, int K, int kflip, Epi ep, hipStream_t s) {
  constexpr int BM = LW * 32, BN = 16 * NB, SLAB = BN * SPLIT_ROW;
  constexpr int LEPI_ALL = LW * 16 * (BN + 4) * 4;
  constexpr int LDS = ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL) + 4096;
  auto kern = nbr ? conv_split32_kernel<NB, LW, false> : conv_split32_kernel<NB, LW, true>;
  static size_t attr[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr[nbr ? 0 : 1][dev] < (size_t)LDS) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[nbr ? 0 : 1][dev] = LDS;
  }
  dim3 grid((unsigned)cdiv(n_out, BM), (unsigned)cdiv(co, BN));
  Split sp{nullptr, 1, 0, 0};         // (the policy of launch_split, in workgroups of 128 rows)
  if (nbr != nullptr && ep.ws != nullptr) {
    const int64_t wgs = cdiv(n_out, 128) * grid.y;
    const int ns = (wgs > SPLIT_F32_MAX_WGS || K * (ci / SPLIT_KCH) < 54) ? 1 : (wgs <= 256 ? 4 : 3);
    const long long rows_pad = (long long)grid.x * BM;
    const int co_pad = (int)grid.y * BN;
    if (ns > 1 && ep.ws_bytes >= (long long)ns * rows_pad * co_pad * 4) sp = Split{(float*)ep.ws, ns, co_pad, rows_pad};
  }
  grid.z = (unsigned)sp.nsplit;
  kern<<<grid, 64 * LW, LDS, s>>>((const float*)in, (const __bf16*)wimg, nbr, perm, tmasks, (float*)out, n_out, ci, co, K,
                                  kflip, ep.scale, ep.shift, ep.relu, (const float*)ep.res, ep.in_bytes, ep.img_bytes,
                                  ep.nbr_bytes, sp);
  LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(split32)");
  if (sp.nsplit > 1) {
    auto ck = conv_combine32_kernel<float, NB / 2, LW>;
    static size_t comb_attr[MAX_DEVICES] = {};
    if (comb_attr[dev] < (size_t)LEPI_ALL) {
      LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ck), hipFuncAttributeMaxDynamicSharedMemorySize, LEPI_ALL));
      comb_attr[dev] = LEPI_ALL;
    }
    grid.z = 1;
    ck<<<grid, 64 * LW, LEPI_ALL, s>>>(sp, perm, (float*)out, n_out, co, ep.scale, ep.shift, ep.relu, (const float*)ep.res);
    LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(split32, combine)");
  }
  return 0;
}

template <int NB>
int launch_split(const void* in, const void* wimg, const int* nbr, const int* perm, const unsigned* tmasks, void* out,
                 int64_t n_out, int ci, int co, int K, int kflip, Epi ep, hipStream_t s) {
  if (split_mfma_rows() == 32) {
    if constexpr (NB < 8) {
      if (split32_waves(NB) == 8)
        return launch_split32<NB, 8>(in, wimg, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
    }
    return launch_split32<NB, 4>(in, wimg, nbr, perm, tmasks, out, n_out, ci, co, K, kflip, ep, s);
  }
  constexpr int LW = 8, BN = 16 * NB, SLAB = BN * SPLIT_ROW;
  constexpr int LEPI_ALL = LW * 16 * (BN + 4) * 4 + LW * BN * 2 * (int)sizeof(float);
  constexpr int LDS = ((2 * SLAB > LEPI_ALL) ? 2 * SLAB : LEPI_ALL) + 4096;
  auto kern = nbr ? conv_split_kernel<NB, LW, false> : conv_split_kernel<NB, LW, true>;
  static size_t attr[2][MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr[nbr ? 0 : 1][dev] < (size_t)LDS) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[nbr ? 0 : 1][dev] = LDS;
  }
  dim3 grid((unsigned)cdiv(n_out, LW * 16), (unsigned)cdiv(co, BN));
  // The offsets of a tile over several workgroups (struct Split) where the launch does not fill the chip once: a
  // tile of a coarse level runs up to 27 offsets x Cin/32 slices back to back (216 phases of ~1 us at 256 channels)
  // while the launch has one to two rounds of workgroups -- as long as its heaviest tile and a quantised round count.
  Split sp{nullptr, 1, 0, 0};
  if (nbr != nullptr && ep.ws != nullptr) {
    const int64_t wgs = (int64_t)grid.x * grid.y;
    const int ns = (wgs > SPLIT_F32_MAX_WGS || K * (ci / SPLIT_KCH) < 54) ? 1 : (wgs <= 256 ? 4 : 3);
    const long long rows_pad = (long long)grid.x * LW * 16;
    const int co_pad = (int)grid.y * BN;
    if (ns > 1 && ep.ws_bytes >= (long long)ns * rows_pad * co_pad * 4) sp = Split{(float*)ep.ws, ns, co_pad, rows_pad};
  }
  grid.z = (unsigned)sp.nsplit;
  kern<<<grid, 64 * LW, LDS, s>>>((const float*)in, (const __bf16*)wimg, nbr, perm, tmasks, (float*)out, n_out, ci, co, K,
                                  kflip, ep.scale, ep.shift, ep.relu, (const float*)ep.res, ep.in_bytes, ep.img_bytes,
                                  ep.nbr_bytes, sp);
  LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(split)");
  if (sp.nsplit > 1) {
    constexpr int COMB_LDS = LEPI_ALL;
    auto ck = conv_combine_kernel<float, NB, LW>;
    static size_t comb_attr[MAX_DEVICES] = {};
    if (comb_attr[dev] < (size_t)COMB_LDS) {
      LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ck), hipFuncAttributeMaxDynamicSharedMemorySize, COMB_LDS));
      comb_attr[dev] = COMB_LDS;
    }
    grid.z = 1;
    BnBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    ck<<<grid, 64 * LW, COMB_LDS, s>>>(sp, perm, (float*)out, n_out, co, ep.scale, ep.shift, ep.relu, (const float*)ep.res,
                                       nullptr, none);
    LIDAL_CHECK_LAUNCH("lidal_conv_apply_image(split, combine)");
  }
  return 0;
}

}  // namespace

extern "C" int lidal_conv_stats_tile_rows(void) { return TILE_ROWS; }

extern "C" int lidal_conv_weight_image_tiling(int ci, int co, int dtype, int64_t n_out) {
  if (dtype == LIDAL_F32_SPLIT) return split_nb(co) * 1000 + SPLIT_ROW + 1;     // (+1: never equal to a bf16 tiling's key)
  const Tiling t = pick_tiling(ci, co, n_out, dtype == LIDAL_BF16 ? 2 : 4);
  return t.nb * 1000 + t.row_bytes;
}

extern "C" int64_t lidal_conv_weight_image_bytes(int k, int ci, int co, int dtype, int64_t n_out) {
  if (dtype == LIDAL_F32_SPLIT) return (ci > 0 && ci % SPLIT_KCH == 0) ? split_image_bytes(k, ci, co) : -1;
  const int esz = dtype == LIDAL_BF16 ? 2 : 4;
  return image_bytes(k, ci, co, pick_tiling(ci, co, n_out, esz), esz);
}

static int weight_images(const void* w, int w_dtype, int role, void* img_a, int64_t n_out_a, void* img_b,
                         int64_t n_out_b, int dtype, int k, int n_red, int n_col, hipStream_t s) {
  if (k == 0 || n_red == 0 || n_col == 0) return 0;
  LIDAL_REQUIRE(role == 0 || role == 1, "weight_image: role must be 0 (forward) or 1 (data gradient)");
  if (dtype == LIDAL_F32_SPLIT) {       // hi | mid | lo bf16 pieces of an f32 (or bf16) weight, see conv_split_kernel
    LIDAL_REQUIRE(img_b == nullptr && n_red % SPLIT_KCH == 0, "weight_image(split): one image per call, reduction a "
                  "multiple of %d channels (got %d)", SPLIT_KCH, n_red);
    const int64_t segs = split_image_bytes(k, n_red, n_col) / 16;
    LIDAL_REQUIRE(segs < (1ll << 31), "weight_image: an image of more than 2^31 segments");
    const unsigned grid = (unsigned)cdiv(segs, 256);
    if (w_dtype == LIDAL_F32)
      weight_image_split_kernel<float><<<grid, 256, 0, s>>>((const float*)w, (__bf16*)img_a, n_red, n_col, role,
                                                            split_nb(n_col), segs);
    else
      weight_image_split_kernel<__bf16><<<grid, 256, 0, s>>>((const __bf16*)w, (__bf16*)img_a, n_red, n_col, role,
                                                             split_nb(n_col), segs);
    LIDAL_CHECK_LAUNCH("lidal_conv_weight_image(split)");
    return 0;
  }
  const int esz = dtype == LIDAL_BF16 ? 2 : 4;
  const Tiling ta = pick_tiling(n_red, n_col, n_out_a, esz);
  const int64_t segs_a = image_bytes(k, n_red, n_col, ta, esz) / 16;
  Tiling tb = ta;
  int64_t segs_b = 0;
  if (img_b != nullptr) {
    tb = pick_tiling(n_col, n_red, n_out_b, esz);
    segs_b = image_bytes(k, n_col, n_red, tb, esz) / 16;
  }
  LIDAL_REQUIRE(segs_a < (1ll << 31) && segs_b < (1ll << 31), "weight_image: an image of more than 2^31 segments");
  const unsigned grid = (unsigned)cdiv(segs_a + segs_b, 256);
#define IMG_LAUNCH(TI, TO) \
  weight_image_kernel<TI, TO><<<grid, 256, 0, s>>>((const TI*)w, (TO*)img_a, n_red, n_col, role, ta.nb, \
                                                   ta.row_bytes / esz, segs_a, (TO*)img_b, tb.nb,       \
                                                   tb.row_bytes / esz, segs_b)
  if (w_dtype == LIDAL_F32 && dtype == LIDAL_F32) IMG_LAUNCH(float, float);
  else if (w_dtype == LIDAL_F32 && dtype == LIDAL_BF16) IMG_LAUNCH(float, __bf16);
  else if (w_dtype == LIDAL_BF16 && dtype == LIDAL_BF16) IMG_LAUNCH(__bf16, __bf16);
  else if (w_dtype == LIDAL_BF16 && dtype == LIDAL_F32) IMG_LAUNCH(__bf16, float);
  else {
    set_error("weight_image: bad dtypes %d %d", w_dtype, dtype);
    return 2;
  }
#undef IMG_LAUNCH
  LIDAL_CHECK_LAUNCH("lidal_conv_weight_image");
  return 0;
}

extern "C" int lidal_conv_weight_image(const void* w, int w_dtype, int role, void* img, int dtype,
                                       int k, int n_red, int n_col, int64_t n_out, void* stream) {
  return weight_images(w, w_dtype, role, img, n_out, nullptr, 0, dtype, k, n_red, n_col, (hipStream_t)stream);
}

extern "C" int lidal_conv_weight_image_pair(const void* w, int w_dtype, void* img_fwd, int64_t n_out_fwd,
                                            void* img_bwd, int64_t n_out_bwd, int dtype, int k, int ci,
                                            int co, void* stream) {
  return weight_images(w, w_dtype, 0, img_fwd, n_out_fwd, img_bwd, n_out_bwd, dtype, k, ci, co,
                       (hipStream_t)stream);
}

extern "C" int lidal_conv_weight_image_job_bytes(void) { return (int)sizeof(ImageJob); }

// fills one ImageJob (host memory) for the pair of images lidal_conv_weight_image_pair would build
// (role 0: w is [k][ci][co]; role 1: w is [k][co][ci], nn.Linear's layout -- the forward image still
// reduces over ci); `first` = segments of the jobs before it; returns this job's segment count (< 0: error)
extern "C" int64_t lidal_conv_weight_image_job(void* job, const void* w, int role, void* img_fwd,
                                               int64_t n_out_fwd, void* img_bwd, int64_t n_out_bwd,
                                               int dtype, int k, int ci, int co, int64_t first) {
  if (job == nullptr || (dtype != LIDAL_F32 && dtype != LIDAL_BF16 && dtype != LIDAL_F32_SPLIT) || (role != 0 && role != 1)) {
    set_error("weight_image_job: bad arguments");
    return -1;
  }
  if (dtype == LIDAL_F32_SPLIT) {       // reduction over ci; with img_bwd (training) also the image that reduces over co
    if (ci % SPLIT_KCH != 0 || (img_bwd != nullptr && co % SPLIT_KCH != 0)) {
      set_error("weight_image_job(split): the reduction must be a multiple of %d channels (got ci=%d%s co=%d)", SPLIT_KCH, ci,
                img_bwd != nullptr ? ", data-gradient image:" : ",", co);
      return -1;
    }
    ImageJob j;
    j.w = w; j.img_a = img_fwd; j.img_b = img_bwd; j.first = first;
    j.segs_a = split_image_bytes(k, ci, co) / 16;
    j.segs_b = img_bwd != nullptr ? split_image_bytes(k, co, ci) / 16 : 0;
    j.n_red = ci; j.n_col = co; j.role = role; j.nb_a = split_nb(co); j.kc_a = 0; j.nb_b = split_nb(ci); j.kc_b = 0; j.pad = 0;
    if (j.segs_a >= (1ll << 31) || j.segs_b >= (1ll << 31)) {
      set_error("weight_image_job: an image of %lld segments", (long long)(j.segs_a > j.segs_b ? j.segs_a : j.segs_b));
      return -1;
    }
    *reinterpret_cast<ImageJob*>(job) = j;
    return j.segs_a + j.segs_b;
  }
  const int esz = dtype == LIDAL_BF16 ? 2 : 4;
  const Tiling ta = pick_tiling(ci, co, n_out_fwd, esz);
  ImageJob j;
  j.w = w; j.img_a = img_fwd; j.img_b = img_bwd; j.first = first;
  j.segs_a = image_bytes(k, ci, co, ta, esz) / 16;
  j.n_red = ci; j.n_col = co; j.role = role; j.nb_a = ta.nb; j.kc_a = ta.row_bytes / esz;
  j.segs_b = 0; j.nb_b = ta.nb; j.kc_b = j.kc_a; j.pad = 0;
  if (img_bwd != nullptr) {
    const Tiling tb = pick_tiling(co, ci, n_out_bwd, esz);
    j.segs_b = image_bytes(k, co, ci, tb, esz) / 16;
    j.nb_b = tb.nb; j.kc_b = tb.row_bytes / esz;
  }
  if (j.segs_a >= (1ll << 31) || j.segs_b >= (1ll << 31)) {       // (image_segment indexes a segment with 32 bits)
    set_error("weight_image_job: an image of %lld segments", (long long)(j.segs_a > j.segs_b ? j.segs_a : j.segs_b));
    return -1;
  }
  *reinterpret_cast<ImageJob*>(job) = j;
  return j.segs_a + j.segs_b;
}

// jobs: DEVICE array of n_jobs ImageJob records (as lidal_conv_weight_image_job filled them, in order)
extern "C" int lidal_conv_weight_image_batch(const void* jobs, int n_jobs, int64_t total_segments,
                                             int w_dtype, int dtype, void* stream) {
  if (n_jobs == 0 || total_segments == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)cdiv(total_segments, 256);
#define IMG_BATCH(TI, TO) \
  weight_image_batch_kernel<TI, TO><<<grid, 256, 0, s>>>((const ImageJob*)jobs, n_jobs, total_segments)
  if (dtype == LIDAL_F32_SPLIT && w_dtype == LIDAL_F32)
    weight_image_batch_split_kernel<float><<<grid, 256, 0, s>>>((const ImageJob*)jobs, n_jobs, total_segments);
  else if (dtype == LIDAL_F32_SPLIT && w_dtype == LIDAL_BF16)
    weight_image_batch_split_kernel<__bf16><<<grid, 256, 0, s>>>((const ImageJob*)jobs, n_jobs, total_segments);
  else if (w_dtype == LIDAL_F32 && dtype == LIDAL_F32) IMG_BATCH(float, float);
  else if (w_dtype == LIDAL_F32 && dtype == LIDAL_BF16) IMG_BATCH(float, __bf16);
  else if (w_dtype == LIDAL_BF16 && dtype == LIDAL_BF16) IMG_BATCH(__bf16, __bf16);
  else if (w_dtype == LIDAL_BF16 && dtype == LIDAL_F32) IMG_BATCH(__bf16, float);
  else {
    set_error("weight_image_batch: bad dtypes %d %d", w_dtype, dtype);
    return 2;
  }
#undef IMG_BATCH
  LIDAL_CHECK_LAUNCH("lidal_conv_weight_image_batch");
  return 0;
}

static int conv_apply_image(const void* in, const void* wimg, const int32_t* nbr, const int32_t* perm,
                            const uint32_t* tile_masks, void* out, int64_t n_in, int64_t n_out, int ci, int co,
                            int k, int kflip, int dtype, const float* ep_scale, const float* ep_shift,
                            int ep_relu, const void* ep_residual, float* tile_stats, BnBwd bnb, void* ws,
                            int64_t ws_bytes, hipStream_t s) {
  if (n_out == 0 || co == 0) return 0;
  LIDAL_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv_apply_image: scale and shift go together");
  LIDAL_REQUIRE(dtype == LIDAL_F32 || dtype == LIDAL_BF16 || dtype == LIDAL_F32_SPLIT, "conv_apply_image: bad dtype %d", dtype);
  if (dtype == LIDAL_F32_SPLIT) {       // f32 in / out, weights as a split image: conv_split_kernel
    LIDAL_REQUIRE(ci > 0 && ci % SPLIT_KCH == 0 && co % 4 == 0 && k > 0 && k <= MAXK,
                  "conv_apply_image(split): ci must be a multiple of %d, co of 4, k <= %d (got ci=%d co=%d k=%d)",
                  SPLIT_KCH, MAXK, ci, co, k);
    LIDAL_REQUIRE(nbr == nullptr ? (k == 1 && n_in == n_out) : (tile_masks != nullptr),
                  "conv_apply_image: needs lidal_kmap_order's tile masks (or NULL table = identity, k = 1)");
    LIDAL_REQUIRE(tile_stats == nullptr && bnb.sums == nullptr, "conv_apply_image(split): no BatchNorm statistics (inference form)");
    const int64_t ib = split_image_bytes(k, ci, co);
    LIDAL_REQUIRE(n_in >= 0 && n_in * ci * 4 < 0x7FFFFFF0ll && ib < 0x7FFFFFF0ll && n_in < (1 << 24),
                  "conv_apply_image: the input matrix and the weight image must each stay below 2 GiB (2^24 rows)");
    LIDAL_REQUIRE((int64_t)k * n_out * 4 < 0x7FFFFFF0ll, "conv_apply_image: neighbour table above 2 GiB");
    Epi ep{ep_scale, ep_shift, ep_relu, ep_residual, (unsigned)(n_in * ci * 4), (unsigned)ib,
           (unsigned)((int64_t)k * n_out * 4), nullptr, bnb, ws, (long long)ws_bytes};
    switch (split_nb(co)) {
      case 2: return launch_split<2>(in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
      case 4: return launch_split<4>(in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
      case 6: return launch_split<6>(in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
      default: return launch_split<8>(in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
    }
  }
  const int esz = dtype == LIDAL_BF16 ? 2 : 4;
  const int vec = 16 / esz;
  LIDAL_REQUIRE(ci > 0 && ci % vec == 0 && co % 4 == 0 && k > 0 && k <= MAXK,
                "conv_apply_image: ci must be a multiple of %d, co of 4, k <= %d (got ci=%d co=%d k=%d)",
                vec, MAXK, ci, co, k);
  LIDAL_REQUIRE(nbr == nullptr ? (k == 1 && n_in == n_out) : (tile_masks != nullptr),
                "conv_apply_image: needs lidal_kmap_order's tile masks (or NULL table = identity, k = 1)");
  LIDAL_REQUIRE(bnb.sums == nullptr || (co % vec == 0 && ep_residual == nullptr && ep_scale == nullptr &&
                                        tile_stats == nullptr && bnb.x && bnb.mean && bnb.invstd),
                "conv_apply_image: BatchNorm backward sums need whole 16-byte column vectors, no other epilogue");
  const Tiling t = pick_tiling(ci, co, n_out, esz);
  const int64_t ib = image_bytes(k, ci, co, t, esz);
  LIDAL_REQUIRE(n_in >= 0 && n_in * ci * esz < 0x7FFFFFF0ll && ib < 0x7FFFFFF0ll,
                "conv_apply_image: the input matrix and the weight image must each stay below 2 GiB");
  LIDAL_REQUIRE((int64_t)k * n_out * 4 < 0x7FFFFFF0ll, "conv_apply_image: neighbour table above 2 GiB");
  Epi ep{ep_scale, ep_shift, ep_relu, ep_residual, (unsigned)(n_in * ci * esz), (unsigned)ib,
         (unsigned)((int64_t)k * n_out * 4), tile_stats, bnb, ws, (long long)ws_bytes};
  if (dtype == LIDAL_F32)
    return dispatch_img<float>(t, in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
  return dispatch_img<__bf16>(t, in, wimg, nbr, perm, tile_masks, out, n_out, ci, co, k, kflip, ep, s);
}

#ifdef LIDAL_PHASE_STAMPS
// (instrumented builds only) u64 [workgroups][waves][12] for the lean kernel's phase stamps, or NULL
extern "C" int lidal_debug_phase_stamps(void* buf) {
  LIDAL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)));
  return 0;
}
#endif

extern "C" int lidal_conv_apply_image(const void* in, const void* wimg, const int32_t* nbr,
                                      const int32_t* perm, const uint32_t* tile_masks, void* out,
                                      int64_t n_in, int64_t n_out, int ci, int co, int k, int kflip,
                                      int dtype, const float* ep_scale, const float* ep_shift,
                                      int ep_relu, const void* ep_residual, float* tile_stats,
                                      void* stream) {
  BnBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return conv_apply_image(in, wimg, nbr, perm, tile_masks, out, n_in, n_out, ci, co, k, kflip, dtype, ep_scale,
                          ep_shift, ep_relu, ep_residual, tile_stats, none, nullptr, 0, (hipStream_t)stream);
}

extern "C" int64_t lidal_conv_apply_workspace_bytes(int64_t n_out, int co) {
  const int64_t tiles = cdiv(n_out > 0 ? n_out : 1, TILE_ROWS);
  if (tiles * cdiv(co, 128) > SPLIT_F32_MAX_WGS) return 0;    // (no tiling makes few enough workgroups to be split)
  return 4 * tiles * TILE_ROWS * align_up(co, 128) * 4;
}

extern "C" int lidal_conv_apply_image_ws(const void* in, const void* wimg, const int32_t* nbr,
                                         const int32_t* perm, const uint32_t* tile_masks, void* out,
                                         int64_t n_in, int64_t n_out, int ci, int co, int k, int kflip,
                                         int dtype, const float* ep_scale, const float* ep_shift,
                                         int ep_relu, const void* ep_residual, float* tile_stats, void* ws,
                                         int64_t ws_bytes, void* stream) {
  BnBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return conv_apply_image(in, wimg, nbr, perm, tile_masks, out, n_in, n_out, ci, co, k, kflip, dtype, ep_scale,
                          ep_shift, ep_relu, ep_residual, tile_stats, none, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int lidal_conv_dgrad_bn_sums(const void* gout, const void* wimg, const int32_t* nbr,
                                        const int32_t* perm, const uint32_t* tile_masks, void* gin,
                                        int64_t n_gout, int64_t n_gin, int c_gout, int c_gin, int k, int kflip,
                                        int dtype, const void* bn_x, const float* bn_mean,
                                        const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                        int bn_relu, float* bn_sums, void* stream) {
  BnBwd b{bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_sums, bn_relu};
  return conv_apply_image(gout, wimg, nbr, perm, tile_masks, gin, n_gout, n_gin, c_gout, c_gin, k, kflip, dtype,
                          nullptr, nullptr, 0, nullptr, nullptr, b, nullptr, 0, (hipStream_t)stream);
}

extern "C" int lidal_conv_dgrad_bn_sums_ws(const void* gout, const void* wimg, const int32_t* nbr,
                                           const int32_t* perm, const uint32_t* tile_masks, void* gin,
                                           int64_t n_gout, int64_t n_gin, int c_gout, int c_gin, int k, int kflip,
                                           int dtype, const void* bn_x, const float* bn_mean,
                                           const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                           int bn_relu, float* bn_sums, void* ws, int64_t ws_bytes, void* stream) {
  BnBwd b{bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_sums, bn_relu};
  return conv_apply_image(gout, wimg, nbr, perm, tile_masks, gin, n_gout, n_gin, c_gout, c_gin, k, kflip, dtype,
                          nullptr, nullptr, 0, nullptr, nullptr, b, ws, ws_bytes, (hipStream_t)stream);
}
