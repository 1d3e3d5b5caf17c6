// Sorted-unique, coordinate downsample and kernel-map (rule) construction for gfx950.
//
// Everything here is order-preserving: torchsparse's rule order is (k, out_idx) ascending and
// the output-coordinate order is the sorted order of torch.unique, so compaction is done with
// wave ballots + prefix sums (never atomic append).  Pattern used three times below:
//   pass 1  per-block count of kept elements           (ballot popcount)
//   pass 2  exclusive scan of the block counts         (one workgroup)
//   pass 3  per-block ordered compaction               (ballot rank + wave offsets in LDS)
#include <cstring>

#include "common.h"

// The reference computes these quantities with numpy (separately rounded products and sums); hipcc
// contracts a*b+c into fma even through __dmul_rn/__dadd_rn, so this unit is built with
// -ffp-contract=off (lidal_amd/build.py).

using namespace lidal;

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 4;                    // elements per thread
constexpr int kTile = kBlock * kItems;       // elements per workgroup

// Exclusive scan of `counts[n]` (int) -> offsets[n] (int64), total -> offsets[n].
__device__ __forceinline__ void scan_counts_body(const int* __restrict__ counts, int64_t n,
                                                 int64_t* __restrict__ offsets) {
  __shared__ int64_t wave_sum[16];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024) {
    int64_t i = base + threadIdx.x;
    int64_t v = (i < n) ? counts[i] : 0;
    int64_t incl = v;                         // inclusive scan inside the wave
    for (int d = 1; d < 64; d <<= 1) {
      int64_t t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_sum[w];
    int64_t carry = carry_s;
    if (i < n) offsets[i] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[n] = carry_s;
}
__global__ void __launch_bounds__(1024) scan_counts_kernel(const int* __restrict__ counts, int64_t n,
                                                           int64_t* __restrict__ offsets) {
  scan_counts_body(counts, n, offsets);
}

// block-level ordered rank: returns the exclusive rank of this thread's kept element among the
// kept elements of the whole block iteration, and the iteration total through *total.
__device__ __forceinline__ int block_rank(bool keep, int* wave_cnt /*[4] LDS*/, int* total) {
  unsigned long long m = __ballot(keep);
  int r = ballot_rank(m);
  int wave = threadIdx.x >> 6;
  if (lane_id() == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    int c = wave_cnt[w];
    if (w < wave) off += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return off + r;
}

// ---------------- sorted unique of u64 keys ----------------
__global__ void __launch_bounds__(kBlock) head_count_kernel(const uint64_t* __restrict__ s,
                                                            int64_t n, int* __restrict__ counts) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * kTile;
  int c = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    int64_t i = base + it * kBlock + threadIdx.x;
    bool head = (i < n) && (i == 0 || s[i] != s[i - 1]);
    c += __popcll(__ballot(head));
  }
  if (lane_id() == 0) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = cnt;
}

__global__ void __launch_bounds__(kBlock) head_compact_kernel(const uint64_t* __restrict__ s,
                                                              int64_t n,
                                                              const int64_t* __restrict__ offsets,
                                                              int64_t nblocks,
                                                              int64_t* __restrict__ out,
                                                              int64_t* __restrict__ n_out) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t base = (int64_t)blockIdx.x * kTile;
  int64_t pos = offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    int64_t i = base + it * kBlock + threadIdx.x;
    uint64_t v = (i < n) ? s[i] : 0;
    bool head = (i < n) && (i == 0 || v != s[i - 1]);
    int tot;
    int r = block_rank(head, wave_cnt, &tot);
    if (head) out[pos + r] = (int64_t)v;
    pos += tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = offsets[nblocks];
}

struct UniqueWs {
  uint64_t* sorted;
  int* counts;
  int64_t* offsets;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t total;
};

UniqueWs carve_unique_ws(void* ws, int64_t n) {
  UniqueWs u;
  int64_t nblocks = cdiv(n > 0 ? n : 1, kTile);
  const size_t tmp = (size_t)radix_sort_ws_bytes(n > 0 ? n : 1, 8, false);
  char* p = (char*)ws;
  int64_t o = 0;
  u.sorted = (uint64_t*)(p + o); o += align_up(8 * (n > 0 ? n : 1), 256);
  u.counts = (int*)(p + o);      o += align_up(4 * nblocks, 256);
  u.offsets = (int64_t*)(p + o); o += align_up(8 * (nblocks + 1), 256);
  u.sort_tmp = (void*)(p + o);   o += align_up((int64_t)tmp, 256);
  u.sort_tmp_bytes = tmp;
  u.total = o;
  return u;
}

int unique_sorted_u64(const uint64_t* keys, int64_t n, int64_t* out, int64_t* n_out_dev, void* ws,
                      int64_t ws_bytes, int end_bit, hipStream_t s) {
  if (n == 0) {
    LIDAL_HIP(hipMemsetAsync(n_out_dev, 0, 8, s));
    return 0;
  }
  UniqueWs u = carve_unique_ws(ws, n);
  LIDAL_REQUIRE(ws_bytes >= u.total, "unique workspace too small: %lld < %lld",
                (long long)ws_bytes, (long long)u.total);
  if (int rc = radix_sort(keys, nullptr, u.sorted, nullptr, n, 8, end_bit, u.sort_tmp, (int64_t)u.sort_tmp_bytes, s))
    return rc;
  int64_t nblocks = cdiv(n, kTile);
  head_count_kernel<<<(int)nblocks, kBlock, 0, s>>>(u.sorted, n, u.counts);
  LIDAL_CHECK_LAUNCH("head_count");
  scan_counts_kernel<<<1, 1024, 0, s>>>(u.counts, nblocks, u.offsets);
  LIDAL_CHECK_LAUNCH("scan_counts");
  head_compact_kernel<<<(int)nblocks, kBlock, 0, s>>>(u.sorted, n, u.offsets, nblocks, out,
                                                       n_out_dev);
  LIDAL_CHECK_LAUNCH("head_compact");
  return 0;
}

// ---------------- downsample: pack (b,x,y,z) -> u64, unique, unpack ----------------
__global__ void __launch_bounds__(256) pack_coords_kernel(const int4* __restrict__ coords,
                                                          int64_t n, int sx, int sy, int sz,
                                                          uint64_t* __restrict__ keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  // floor division: coordinates are non-negative on this path (checked by the caller)
  uint64_t x = (uint64_t)((c.x / sx) * sx), y = (uint64_t)((c.y / sy) * sy),
           z = (uint64_t)((c.z / sz) * sz), b = (uint64_t)c.w;
  keys[i] = (b << 48) | (x << 32) | (y << 16) | z;
}

__global__ void __launch_bounds__(256) unpack_coords_kernel(const int64_t* __restrict__ keys,
                                                            const int64_t* __restrict__ n_dev,
                                                            int4* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_dev) return;
  uint64_t k = (uint64_t)keys[i];
  int4 c;
  c.w = (int)(k >> 48);
  c.x = (int)((k >> 32) & 0xFFFF);
  c.y = (int)((k >> 16) & 0xFFFF);
  c.z = (int)(k & 0xFFFF);
  out[i] = c;
}

// ---------------- kernel map ----------------
// kItems table look-ups of one thread with their first probes all in flight before any is
// resolved (a look-up is a dependent chain key -> value; one at a time leaves the memory system idle).
__device__ __forceinline__ void lookup_batch(const TableView& t, const uint64_t (&key)[kItems],
                                             const bool (&ok)[kItems], int (&r)[kItems]) {
  uint64_t slot[kItems];
  unsigned long long first[kItems];
  bool maybe[kItems];
  // the occupancy bitmap first (common.h): most of the probed neighbours do not exist, and a clear bit says so
  // from the L2s -- only the keys that pass go on to their slot
  if (t.bits != nullptr) {
    unsigned word[kItems];
    uint64_t mixed[kItems];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      mixed[it] = mix_key(key[it]);
      slot[it] = mixed[it] & t.mask;
      const uint64_t b = bit_of(mixed[it], t.mask);
      word[it] = ok[it] ? t.bits[b >> 5] : 0u;
    }
#pragma unroll
    for (int it = 0; it < kItems; ++it) maybe[it] = ok[it] && ((word[it] >> (bit_of(mixed[it], t.mask) & 31)) & 1u);
  } else {
#pragma unroll
    for (int it = 0; it < kItems; ++it) { slot[it] = slot_of(key[it], t.mask); maybe[it] = ok[it]; }
  }
#pragma unroll
  for (int it = 0; it < kItems; ++it) first[it] = maybe[it] ? t.keys[slot[it]] : kEmptyKey;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    r[it] = -1;
    if (!maybe[it] || first[it] == kEmptyKey) continue;
    if (first[it] == key[it]) { r[it] = t.vals[slot[it]]; continue; }
    uint64_t sl = (slot[it] + 1) & t.mask;                 // collision: walk on
    while (true) {
      unsigned long long k = t.keys[sl];
      if (k == key[it]) { r[it] = t.vals[sl]; break; }
      if (k == kEmptyKey) break;
      sl = (sl + 1) & t.mask;
    }
  }
}

// pass 1: probe.  Workgroup (b, k) looks up offset k for the kTile output rows of block b (one
// thread per (row, offset) pair x kItems rows: the K look-ups of a row are independent, so they run
// as separate threads rather than one serial chain per row); nbr_out[k][j] stores are coalesced
// along j; counts[k][b] = hits of the block.
__device__ __forceinline__ void kmap_probe_body(const TableView& t, const int4* __restrict__ coords, int64_t n_out,
                                                const int* __restrict__ offsets, int* __restrict__ nbr_out,
                                                int* __restrict__ counts, int64_t nblocks, int64_t bx, int k) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int ox = offsets[k * 3 + 0], oy = offsets[k * 3 + 1], oz = offsets[k * 3 + 2];
  const int64_t base = bx * kTile;
  uint64_t key[kItems];
  bool ok[kItems];
  int r[kItems];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    ok[it] = j < n_out;
    const int4 c = ok[it] ? coords[j] : make_int4(0, 0, 0, 0);
    key[it] = (uint64_t)fnv60(c.x + ox, c.y + oy, c.z + oz, c.w);
  }
  lookup_batch(t, key, ok, r);
  int found = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    if (ok[it]) nbr_out[(int64_t)k * n_out + j] = r[it];
    found += __popcll(__ballot(r[it] >= 0));
  }
  if (lane_id() == 0) atomicAdd(&cnt, found);
  __syncthreads();
  if (threadIdx.x == 0) counts[(int64_t)k * nblocks + bx] = cnt;
}
__global__ void __launch_bounds__(kBlock) kmap_probe_kernel(TableView t, const int4* __restrict__ coords,
                                                            int64_t n_out, const int* __restrict__ offsets, int K,
                                                            int* __restrict__ nbr_out, int* __restrict__ counts,
                                                            int64_t nblocks) {
  kmap_probe_body(t, coords, n_out, offsets, nbr_out, counts, nblocks, blockIdx.x, blockIdx.y);
}

// pass 1 (symmetric form): when the output coordinates ARE the input coordinates and the kernel is
// odd and centred (every k3 stride-1 conv), rule (i, j, k) implies rule (j, i, K-1-k) and the centre
// offset is the identity.  Only the first K/2 offsets are probed (grid.y = K/2); each hit also
// fills its mirror entry (a unique (offset, row) slot, so no write conflicts).  Halves the probes.
__device__ __forceinline__ void kmap_probe_sym_body(const TableView& t, const int4* __restrict__ coords, int64_t n,
                                                    const int* __restrict__ offsets, int K,
                                                    int* __restrict__ nbr_out, int64_t bx, int k) {
  const int64_t base = bx * kTile;
  const int half = K / 2;
  const int ox = offsets[k * 3 + 0], oy = offsets[k * 3 + 1], oz = offsets[k * 3 + 2];
  uint64_t key[kItems];
  bool ok[kItems];
  int r[kItems];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    ok[it] = j < n;
    const int4 c = ok[it] ? coords[j] : make_int4(0, 0, 0, 0);
    key[it] = (uint64_t)fnv60(c.x + ox, c.y + oy, c.z + oz, c.w);
  }
  lookup_batch(t, key, ok, r);
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    if (!ok[it]) continue;
    if (k == 0) nbr_out[(int64_t)half * n + j] = (int)j;            // centre offset: identity
    nbr_out[(int64_t)k * n + j] = r[it];
    if (r[it] >= 0) nbr_out[(int64_t)(K - 1 - k) * n + r[it]] = (int)j;
  }
}
// The symmetric probe through the SPATIAL bitmap (common.h; tables built by lidal_hash_table_build_coords): one thread
// per (row, group), group g = the up to three offsets that share (dy, dz) -- k = 3g .. 3g + 2 for g < K/6, the single
// offset K/2 - 1 ... for the last: with x-fastest offsets (nn/utils.py get_kernel_offsets, odd kernels) they are x - s, x,
// x + s, one 32-bit word of the bitmap (two at a word boundary).  A clear bit is the answer; a set bit goes on to its slot.
// 5 word reads per voxel instead of 13 hashed-bit reads for a 3x3x3 map; same table entries bit for bit.
__device__ __forceinline__ int slot_lookup(const TableView& t, uint64_t key) {
  uint64_t sl = slot_of(key, t.mask);
  while (true) {
    const unsigned long long k2 = t.keys[sl];
    if (k2 == key) return t.vals[sl];
    if (k2 == kEmptyKey) return -1;
    sl = (sl + 1) & t.mask;
  }
}
__device__ __forceinline__ void kmap_probe_sym_spatial_body(const TableView& t, const int4* __restrict__ coords, int64_t n,
                                                            const int* __restrict__ offsets, int K,
                                                            int* __restrict__ nbr_out, int64_t bx, int g) {
  const int half = K / 2;
  const int k0 = 3 * g;
  if (k0 >= half) return;                            // (the launch has K/2 block rows: the groups use the first of them)
  const int nk = (half - k0 < 3) ? half - k0 : 3;    // offsets of this group
  const int shift = t.hdr[1], xb = t.hdr[2], yb = t.hdr[3];
  int ox[3], oy[3], oz[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int k = k0 + (u < nk ? u : 0);
    ox[u] = offsets[k * 3 + 0]; oy[u] = offsets[k * 3 + 1]; oz[u] = offsets[k * 3 + 2];
  }
  const int64_t base = bx * kTile;
  int4 c[kItems];
  bool ok[kItems];
  unsigned sb[kItems][3], word[kItems][3];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    ok[it] = j < n;
    c[it] = ok[it] ? coords[j] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 3; ++u) sb[it][u] = sbit_of(c[it].x + ox[u], c[it].y + oy[u], c[it].z + oz[u], c[it].w, shift, xb, yb);
  }
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    // (three reads of mostly ONE word: the second and third hit the line the first brought in)
#pragma unroll
    for (int u = 0; u < 3; ++u) word[it][u] = (ok[it] && u < nk) ? t.sbits[sb[it][u] >> 5] : 0u;
  }
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t j = base + it * kBlock + threadIdx.x;
    if (!ok[it]) continue;
    if (g == 0) nbr_out[(int64_t)half * n + j] = (int)j;            // centre offset: identity
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (u >= nk) continue;
      const int k = k0 + u;
      int r = -1;
      if ((word[it][u] >> (sb[it][u] & 31u)) & 1u)
        r = slot_lookup(t, (uint64_t)fnv60(c[it].x + ox[u], c[it].y + oy[u], c[it].z + oz[u], c[it].w));
      nbr_out[(int64_t)k * n + j] = r;
      if (r >= 0) nbr_out[(int64_t)(K - 1 - k) * n + r] = (int)j;
    }
  }
}
__device__ __forceinline__ bool table_is_spatial(const TableView& t) {
  return t.hdr != nullptr && t.sbits != nullptr && t.hdr[0] == kSpatialMagic;
}
__global__ void __launch_bounds__(kBlock) kmap_probe_sym_kernel(TableView t, const int4* __restrict__ coords,
                                                                int64_t n, const int* __restrict__ offsets, int K,
                                                                int* __restrict__ nbr_out) {
  if (table_is_spatial(t)) kmap_probe_sym_spatial_body(t, coords, n, offsets, K, nbr_out, blockIdx.x, blockIdx.y);
  else kmap_probe_sym_body(t, coords, n, offsets, K, nbr_out, blockIdx.x, blockIdx.y);
}

// per-(offset, block) counts of an already filled table (feeds the same scan + compaction)
__device__ __forceinline__ void kmap_count_body(const int* __restrict__ nbr_out, int64_t n_out,
                                                int* __restrict__ counts, int64_t nblocks, int64_t bx, int k) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int64_t base = bx * kTile;
  int found = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    int64_t j = base + it * kBlock + threadIdx.x;
    int r = (j < n_out) ? nbr_out[(int64_t)k * n_out + j] : -1;
    found += __popcll(__ballot(r >= 0));
  }
  if (lane_id() == 0) atomicAdd(&cnt, found);
  __syncthreads();
  if (threadIdx.x == 0) counts[(int64_t)k * nblocks + bx] = cnt;
}
__global__ void __launch_bounds__(kBlock) kmap_count_kernel(const int* __restrict__ nbr_out, int64_t n_out, int K,
                                                            int* __restrict__ counts, int64_t nblocks) {
  kmap_count_body(nbr_out, n_out, counts, nblocks, blockIdx.x, blockIdx.y);
}

// pass 3: ordered compaction of (in_idx, out_idx) pairs, block (b, k).
__device__ __forceinline__ void kmap_compact_body(const int* __restrict__ nbr_out, int64_t n_out,
                                                  const int64_t* __restrict__ offsets, int64_t nblocks,
                                                  int2* __restrict__ nbmaps, int64_t bx, int k) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t base = bx * kTile;
  int64_t pos = offsets[(int64_t)k * nblocks + bx];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    int64_t j = base + it * kBlock + threadIdx.x;
    int r = (j < n_out) ? nbr_out[(int64_t)k * n_out + j] : -1;
    int tot;
    int rank = block_rank(r >= 0, wave_cnt, &tot);
    if (r >= 0) nbmaps[pos + rank] = make_int2(r, (int)j);
    pos += tot;
  }
}
__global__ void __launch_bounds__(kBlock) kmap_compact_kernel(const int* __restrict__ nbr_out, int64_t n_out,
                                                              const int64_t* __restrict__ offsets, int64_t nblocks,
                                                              int2* __restrict__ nbmaps) {
  kmap_compact_body(nbr_out, n_out, offsets, nblocks, nbmaps, blockIdx.x, blockIdx.y);
}

__global__ void kmap_sizes_kernel(const int64_t* __restrict__ offsets, int64_t nblocks, int K,
                                  int* __restrict__ nbsizes, int64_t* __restrict__ koff) {
  int k = threadIdx.x;
  if (k < K) {
    int64_t a = offsets[(int64_t)k * nblocks], b = offsets[(int64_t)(k + 1) * nblocks];
    nbsizes[k] = (int)(b - a);
    koff[k] = a;
  }
  if (k == K) koff[K] = offsets[(int64_t)K * nblocks];
}

__global__ void __launch_bounds__(256) kmap_invert_kernel(const int* __restrict__ nbr_out,
                                                          int64_t n_out, int K,
                                                          int* __restrict__ nbr_in, int64_t n_in) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (j >= n_out) return;
  int i = nbr_out[(int64_t)k * n_out + j];
  if (i >= 0) nbr_in[(int64_t)k * n_in + i] = (int)j;
}



// ---------------- occupancy-pattern row order for the output-stationary conv kernel -------------
// mask[j] = bit k set iff nbr[k][j] >= 0.  Sorting the rows by this K-bit pattern puts rows with
// the same set of occupied offsets next to each other, so a 16-row MFMA group (and a 128-row tile)
// needs only the offsets of ITS pattern: on LiDAR surfaces the non-empty (group, offset) fraction
// drops from 0.5-0.8 to ~0.23 (waste 1.3x instead of 3-4.5x).  Output: perm (sorted position ->
// row) and the table permuted into that order.
// Sort key of a row = its mask with the bits re-ranked so that the RAREST offsets are the most
// significant: for a 3x3x3 kernel the eight corner offsets, then the twelve edges, the six faces and
// the centre (rank by the L1 norm of the offset, a property of the index k alone: its three base-3
// digits).  Rows that differ only in common offsets then sit next to each other and a 128-row tile
// spans fewer distinct rare offsets: on the bench batch the active offsets per tile drop from 7.8
// (plain mask value) to 7.0.  Other volumes keep the plain mask.
struct BitRank { unsigned char to_key[32]; unsigned char to_mask[32]; };
__host__ __device__ inline BitRank bit_rank(int K) {
  BitRank r;
  for (int i = 0; i < 32; ++i) r.to_key[i] = r.to_mask[i] = (unsigned char)i;
  if (K != 27) return r;
  int pos = 26;
  for (int want = 3; want >= 0; --want)
    for (int k = 0; k < 27; ++k) {
      const int a = k % 3, b = (k / 3) % 3, c = k / 9;
      const int l1 = (a != 1) + (b != 1) + (c != 1);
      if (l1 == want) { r.to_key[k] = (unsigned char)pos; r.to_mask[pos] = (unsigned char)k; --pos; }
    }
  return r;
}

__global__ void __launch_bounds__(256) row_mask_kernel(const int* __restrict__ nbr, int64_t n,
                                                       int K, BitRank rank,
                                                       unsigned* __restrict__ keys,
                                                       int* __restrict__ vals) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned m = 0u;
  for (int k = 0; k < K; ++k)
    if (nbr[(int64_t)k * n + j] >= 0) m |= 1u << rank.to_key[k];
  // sorted by Gray rank, not binary value: consecutive keys then differ in few bits (7.0 -> 6.8)
  m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
  keys[j] = m;
  vals[j] = (int)j;
}

__global__ void __launch_bounds__(256) permute_table_kernel(const int* __restrict__ nbr, int64_t n,
                                                            const int* __restrict__ perm,
                                                            int* __restrict__ nbr_perm) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (r >= n) return;
  nbr_perm[(int64_t)k * n + r] = nbr[(int64_t)k * n + perm[r]];
}

// OR of the sorted row masks over each tile of 128 consecutive sorted rows (one wave per tile)
__global__ void __launch_bounds__(256) tile_or_kernel(const unsigned* __restrict__ skeys, int64_t n,
                                                      BitRank rank,
                                                      unsigned* __restrict__ tmask, int64_t tiles) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tiles) return;
  unsigned m = 0u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int64_t r = t * 128 + h * 64 + lane;
    if (r < n) { const unsigned g = skeys[r]; m |= g ^ (g >> 1); }     // Gray rank -> key bits
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, 64);
  if (lane == 0) {                       // back from key bits to offset bits
    unsigned real = 0u;
    for (int b = 0; b < 32; ++b)
      if ((m >> b) & 1u) real |= 1u << rank.to_mask[b];
    tmask[t] = real;
  }
}

// ---------------- input voxelisation (dataset/sk_dataset.py:143-171 on the GPU) ------------------
// 1. affine: p' = p(f32 -> f64) * M (f64, row vector times matrix, k = 0,1,2 in order, no FMA);
//    feats = (f32)p', intensity;  scaled = p' * scale;  per-block min/max of `scaled`
// 2. offset from the global min/max and the host's random draws; voxel = (int)(scaled + offset)
//    (C truncation, as ndarray.astype(int)); key = x << 26 | y << 13 | z  (coords < 8192)
// 3. np.unique(axis=0, return_index, return_inverse): stable radix sort of (key, point id), run
//    heads -> unique rows in lexicographic (x,y,z) order, first-occurrence index, inverse map.
__global__ void __launch_bounds__(256) affine_kernel(const float* __restrict__ pts,
                                                     const float* __restrict__ inten, int64_t p,
                                                     const double* __restrict__ M, double scale,
                                                     float* __restrict__ feats,
                                                     double* __restrict__ scaled,
                                                     double* __restrict__ part /*[blocks][6]*/) {
  __shared__ double red[6][256];
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < p) {
    double x = (double)pts[i * 3 + 0], y = (double)pts[i * 3 + 1], z = (double)pts[i * 3 + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double v = __dadd_rn(__dadd_rn(__dmul_rn(x, M[0 * 3 + j]), __dmul_rn(y, M[1 * 3 + j])),
                           __dmul_rn(z, M[2 * 3 + j]));
      feats[i * 4 + j] = (float)v;
      double sv = __dmul_rn(v, scale);
      scaled[i * 3 + j] = sv;
      lo[j] = sv; hi[j] = sv;
    }
    feats[i * 4 + 3] = inten[i];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { red[j][threadIdx.x] = lo[j]; red[3 + j][threadIdx.x] = hi[j]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        red[j][threadIdx.x] = fmin(red[j][threadIdx.x], red[j][threadIdx.x + w]);
        red[3 + j][threadIdx.x] = fmax(red[3 + j][threadIdx.x], red[3 + j][threadIdx.x + w]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[(int64_t)blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

// single block: global min/max -> offset[3] (sk_dataset.py:154-157)
__global__ void __launch_bounds__(256) voxel_offset_kernel(const double* __restrict__ part,
                                                           int64_t nblocks,
                                                           const double* __restrict__ rnd /*[6]*/,
                                                           double full, double* __restrict__ offset) {
  __shared__ double red[6][256];
  double v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int64_t b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      v[j] = fmin(v[j], part[b * 6 + j]);
      v[3 + j] = fmax(v[3 + j], part[b * 6 + 3 + j]);
    }
#pragma unroll
  for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = v[j];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        red[j][threadIdx.x] = fmin(red[j][threadIdx.x], red[j][threadIdx.x + w]);
        red[3 + j][threadIdx.x] = fmax(red[3 + j][threadIdx.x], red[3 + j][threadIdx.x + w]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    const int j = threadIdx.x;
    const double cmin = red[j][0], cmax = red[3 + j][0];
    // offset = -cmin + clip(full - cmax + cmin - 0.001, 0, None) * r1 + clip(full - cmax + cmin + 0.001, None, 0) * r2
    double a = __dadd_rn(__dadd_rn(__dadd_rn(full, -cmax), cmin), -0.001);
    double b = __dadd_rn(__dadd_rn(__dadd_rn(full, -cmax), cmin), 0.001);
    a = a < 0.0 ? 0.0 : a;
    b = b > 0.0 ? 0.0 : b;
    offset[j] = __dadd_rn(__dadd_rn(-cmin, __dmul_rn(a, rnd[j])), __dmul_rn(b, rnd[3 + j]));
  }
}

__global__ void __launch_bounds__(256) voxel_keys_kernel(const double* __restrict__ scaled,
                                                         int64_t p,
                                                         const double* __restrict__ offset,
                                                         int full, uint64_t* __restrict__ keys,
                                                         int* __restrict__ vals,
                                                         int* __restrict__ n_invalid) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  uint64_t key = 0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double c = __dadd_rn(scaled[i * 3 + j], offset[j]);
    bad |= !(c >= 0.0) || !(c < (double)full);          // sk_dataset.py:160-161 validity assert
    int64_t v = (int64_t)c;                              // astype(int): truncation
    key = (key << 13) | (uint64_t)(v & 0x1FFF);
  }
  keys[i] = key;
  vals[i] = (int)i;
  if (bad) atomicAdd(n_invalid, 1);
}

// after the stable sort: run heads -> unique rows.  uniq_idx[run] = first occurrence (smallest
// original index of the run), inverse[orig] = run, coords_v[run] = unpacked key.
__global__ void __launch_bounds__(kBlock) rows_compact_kernel(const uint64_t* __restrict__ skeys,
                                                              const int* __restrict__ sidx,
                                                              int64_t n,
                                                              const int64_t* __restrict__ offsets,
                                                              int64_t nblocks,
                                                              int* __restrict__ coords_v,
                                                              int64_t* __restrict__ uniq_idx,
                                                              int64_t* __restrict__ inverse,
                                                              int64_t* __restrict__ n_out) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t base = (int64_t)blockIdx.x * kTile;
  int64_t pos = offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    int64_t i = base + it * kBlock + threadIdx.x;
    uint64_t v = (i < n) ? skeys[i] : 0;
    bool head = (i < n) && (i == 0 || v != skeys[i - 1]);
    int tot;
    int r = block_rank(head, wave_cnt, &tot);
    if (i < n) {
      int64_t run = pos + r - (head ? 0 : 1);
      int orig = sidx[i];
      inverse[orig] = run;
      if (head) {
        uniq_idx[run] = orig;
        coords_v[run * 3 + 0] = (int)((v >> 26) & 0x1FFF);
        coords_v[run * 3 + 1] = (int)((v >> 13) & 0x1FFF);
        coords_v[run * 3 + 2] = (int)(v & 0x1FFF);
      }
    }
    pos += tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = offsets[nblocks];
}

size_t voxel_sort_tmp_bytes(int64_t n) { return (size_t)radix_sort_ws_bytes(n > 0 ? n : 1, 8, true); }

}  // namespace

extern "C" int64_t lidal_unique_workspace_bytes(int64_t n) {
  return carve_unique_ws(nullptr, n).total + 256;
}

extern "C" int lidal_unique_sorted_i64(const int64_t* keys, int64_t n, int64_t* out,
                                       int64_t* n_out_dev, void* ws, int64_t ws_bytes,
                                       void* stream) {
  return unique_sorted_u64((const uint64_t*)keys, n, out, n_out_dev, ws, ws_bytes, 64,
                           (hipStream_t)stream);
}

extern "C" int lidal_downsample(const int32_t* coords, int64_t n, int sx, int sy, int sz,
                                int32_t* out, int64_t* n_out_dev, void* ws, int64_t ws_bytes,
                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(sx > 0 && sy > 0 && sz > 0, "downsample: bad stride");
  if (n == 0) {
    LIDAL_HIP(hipMemsetAsync(n_out_dev, 0, 8, s));
    return 0;
  }
  // workspace: [packed keys 8n][unique keys 8n][unique ws]
  int64_t kb = align_up(8 * n, 256);
  int64_t need = 2 * kb + lidal_unique_workspace_bytes(n);
  LIDAL_REQUIRE(ws_bytes >= need, "downsample workspace too small: %lld < %lld",
                (long long)ws_bytes, (long long)need);
  uint64_t* packed = (uint64_t*)ws;
  int64_t* uniq = (int64_t*)((char*)ws + kb);
  void* uws = (char*)ws + 2 * kb;
  pack_coords_kernel<<<(int)cdiv(n, 256), 256, 0, s>>>((const int4*)coords, n, sx, sy, sz, packed);
  LIDAL_CHECK_LAUNCH("pack_coords");
  int rc = unique_sorted_u64(packed, n, uniq, n_out_dev, uws, ws_bytes - 2 * kb, 63, s);
  if (rc) return rc;
  unpack_coords_kernel<<<(int)cdiv(n, 256), 256, 0, s>>>(uniq, n_out_dev, (int4*)out);
  LIDAL_CHECK_LAUNCH("unpack_coords");
  return 0;
}

// ---- every coarser level of a stride-2 pyramid from ONE sort ---------------------------------
// The encoder halves the resolution L times (network/spvcnn.py:28,34,40,46): level l is the sorted unique
// of floor(c / (2^l s)) (2^l s) over the INPUT voxels -- chaining F.spdownsample gives the same sets
// (floor of a floor) in the same (batch, x, y, z) order.  So instead of L dependent sorts with a host round
// trip between them (each level's row count), the L key arrays of the input voxels are one array,
// key = (level - 1) << 61 | batch << 48 | x << 32 | y << 16 | z, sorted once; the run heads are the levels'
// voxels, level l's heads sit between sorted positions (l-1) n and l n.  One sync for all row counts.
namespace {
__global__ void __launch_bounds__(256) pyramid_keys_kernel(const int4* __restrict__ coords, int64_t n, int levels,
                                                           int sx, int sy, int sz,
                                                           uint64_t* __restrict__ keys, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  // the key has 16 bits per coordinate and 13 for the batch index: anything else would spill into its neighbours
  if (((unsigned)c.x | (unsigned)c.y | (unsigned)c.z) >> 16 || (unsigned)c.w >> 13) *bad = 1;
  for (int l = 1; l <= levels; ++l) {
    const int fx = sx << l, fy = sy << l, fz = sz << l;
    const uint64_t x = (uint64_t)((c.x / fx) * fx), y = (uint64_t)((c.y / fy) * fy), z = (uint64_t)((c.z / fz) * fz);
    keys[(int64_t)(l - 1) * n + i] = ((uint64_t)(l - 1) << 61) | ((uint64_t)c.w << 48) | (x << 32) | (y << 16) | z;
  }
}

// run heads of the sorted keys -> coordinates; the thread that owns sorted position l * n also records how
// many heads come before it: the first output row of level l + 1 (starts[levels] = all heads)
__global__ void __launch_bounds__(kBlock) pyramid_compact_kernel(const uint64_t* __restrict__ s, int64_t total,
                                                                 int64_t n, int levels,
                                                                 const int64_t* __restrict__ offsets,
                                                                 int64_t nblocks, int4* __restrict__ out,
                                                                 int64_t* __restrict__ starts,
                                                                 const int* __restrict__ bad) {
  __shared__ int wave_cnt[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  int64_t pos = offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t i = base + it * kBlock + threadIdx.x;
    const uint64_t v = (i < total) ? s[i] : 0;
    const bool head = (i < total) && (i == 0 || v != s[i - 1]);
    int tot;
    const int r = block_rank(head, wave_cnt, &tot);
    if (i < total && i % n == 0) starts[i / n] = pos + r;         // (position l n is always a head: a new level)
    if (head) {
      int4 c;
      c.w = (int)((v >> 48) & 0x1FFF);
      c.x = (int)((v >> 32) & 0xFFFF);
      c.y = (int)((v >> 16) & 0xFFFF);
      c.z = (int)(v & 0xFFFF);
      out[pos + r] = c;
    }
    pos += tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) starts[levels] = *bad ? -1 : offsets[nblocks];
}
}  // namespace

extern "C" int64_t lidal_downsample_pyramid_workspace_bytes(int64_t n, int levels) {
  const int64_t q = (n > 0 ? n : 1) * (levels > 0 ? levels : 1);
  const int64_t nblocks = cdiv(q, kTile);
  return 2 * align_up(8 * q, 256) + align_up(4 * nblocks, 256) + align_up(8 * (nblocks + 1), 256) +
         align_up(radix_sort_ws_bytes(q, 8, false), 256) + 256;
}

// coords i32 [n, 4] at tensor stride (sx, sy, sz); out i32 [levels * n, 4] capacity: level l (1-based: stride
// 2^l s) occupies rows [starts[l-1], starts[l]); starts_dev i64 [levels + 1].  Requires 0 <= x, y, z < 65536,
// 0 <= batch < 8192, levels <= 4; a row outside these ranges makes starts[levels] = -1 (nothing else is valid then).
extern "C" int lidal_downsample_pyramid(const int32_t* coords, int64_t n, int sx, int sy, int sz, int levels,
                                        int32_t* out, int64_t* starts_dev, void* ws, int64_t ws_bytes,
                                        void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(sx > 0 && sy > 0 && sz > 0 && levels >= 1 && levels <= 4, "downsample_pyramid: bad stride / levels");
  if (n == 0) {
    LIDAL_HIP(hipMemsetAsync(starts_dev, 0, 8 * (levels + 1), s));
    return 0;
  }
  LIDAL_REQUIRE(ws_bytes >= lidal_downsample_pyramid_workspace_bytes(n, levels), "downsample_pyramid ws too small");
  const int64_t q = n * levels, nblocks = cdiv(q, kTile);
  char* w = (char*)ws;
  uint64_t* keys = (uint64_t*)w;    w += align_up(8 * q, 256);
  uint64_t* sorted = (uint64_t*)w;  w += align_up(8 * q, 256);
  int* counts = (int*)w;            w += align_up(4 * nblocks, 256);
  int64_t* offs = (int64_t*)w;      w += align_up(8 * (nblocks + 1), 256);
  void* tmp = (void*)w;
  int* bad = (int*)((char*)ws + lidal_downsample_pyramid_workspace_bytes(n, levels) - 256);      // (the slack at the end)
  LIDAL_HIP(hipMemsetAsync(bad, 0, 4, s));
  pyramid_keys_kernel<<<(unsigned)cdiv(n, 256), 256, 0, s>>>((const int4*)coords, n, levels, sx, sy, sz, keys, bad);
  LIDAL_CHECK_LAUNCH("pyramid_keys");
  if (int rc = radix_sort(keys, nullptr, sorted, nullptr, q, 8, 63, tmp, radix_sort_ws_bytes(q, 8, false), s)) return rc;
  head_count_kernel<<<(unsigned)nblocks, kBlock, 0, s>>>(sorted, q, counts);
  LIDAL_CHECK_LAUNCH("head_count");
  scan_counts_kernel<<<1, 1024, 0, s>>>(counts, nblocks, offs);
  LIDAL_CHECK_LAUNCH("scan_counts");
  pyramid_compact_kernel<<<(unsigned)nblocks, kBlock, 0, s>>>(sorted, q, n, levels, offs, nblocks, (int4*)out,
                                                              starts_dev, bad);
  LIDAL_CHECK_LAUNCH("pyramid_compact");
  return 0;
}

extern "C" int64_t lidal_downsample_workspace_bytes(int64_t n) {
  return 2 * align_up(8 * (n > 0 ? n : 1), 256) + lidal_unique_workspace_bytes(n);
}

extern "C" int64_t lidal_kmap_workspace_bytes(int64_t n_out, int k) {
  int64_t nblocks = cdiv(n_out > 0 ? n_out : 1, kTile);
  return align_up(4 * nblocks * k, 256) + align_up(8 * (nblocks * k + 1), 256) + 256;
}

extern "C" int lidal_kmap_build(const void* table, int64_t table_bytes, const int32_t* out_coords,
                                int64_t n_out, const int32_t* offsets, int k, int symmetric,
                                int32_t* nbr_out, int32_t* nbmaps, int32_t* nbsizes, int64_t* koff,
                                int mode, void* ws, int64_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(k > 0 && k <= 1023, "kmap: bad kernel volume %d", k);
  LIDAL_REQUIRE(mode >= 0 && mode <= 2, "kmap: mode must be 0 (table + rules), 1 (table) or 2 (rules)");
  const bool want_table = mode != 2, want_rules = mode != 1;
  if (n_out == 0) {
    if (want_rules) {
      LIDAL_HIP(hipMemsetAsync(nbsizes, 0, 4 * k, s));
      LIDAL_HIP(hipMemsetAsync(koff, 0, 8 * (k + 1), s));
    }
    return 0;
  }
  LIDAL_REQUIRE(ws_bytes >= lidal_kmap_workspace_bytes(n_out, k), "kmap workspace too small");
  int64_t nblocks = cdiv(n_out, kTile);
  int* counts = (int*)ws;
  int64_t* offs = (int64_t*)((char*)ws + align_up(4 * nblocks * k, 256));
  bool counted = false;
  if (want_table) {
    TableView t = table_view(table, table_bytes);
    if (symmetric && (k & 1) && k >= 3) {
      // mirrored entries that receive no hit must read -1
      LIDAL_HIP(hipMemsetAsync(nbr_out + (int64_t)(k / 2 + 1) * n_out, 0xFF, 4 * n_out * (k / 2), s));
      kmap_probe_sym_kernel<<<dim3((unsigned)nblocks, (unsigned)(k / 2)), kBlock, 0, s>>>(
          t, (const int4*)out_coords, n_out, offsets, k, nbr_out);
      LIDAL_CHECK_LAUNCH("kmap_probe_sym");
    } else {
      kmap_probe_kernel<<<dim3((unsigned)nblocks, (unsigned)k), kBlock, 0, s>>>(
          t, (const int4*)out_coords, n_out, offsets, k, nbr_out, counts, nblocks);
      LIDAL_CHECK_LAUNCH("kmap_probe");
      counted = true;
    }
  }
  if (!want_rules) return 0;
  if (!counted) {
    kmap_count_kernel<<<dim3((unsigned)nblocks, (unsigned)k), kBlock, 0, s>>>(nbr_out, n_out, k,
                                                                              counts, nblocks);
    LIDAL_CHECK_LAUNCH("kmap_count");
  }
  scan_counts_kernel<<<1, 1024, 0, s>>>(counts, nblocks * k, offs);
  LIDAL_CHECK_LAUNCH("kmap_scan");
  kmap_compact_kernel<<<dim3((unsigned)nblocks, (unsigned)k), kBlock, 0, s>>>(
      nbr_out, n_out, offs, nblocks, (int2*)nbmaps);
  LIDAL_CHECK_LAUNCH("kmap_compact");
  kmap_sizes_kernel<<<1, 1024, 0, s>>>(offs, nblocks, k, nbsizes, koff);
  LIDAL_CHECK_LAUNCH("kmap_sizes");
  return 0;
}

// torchsparse-order rule lists -> neighbour table: nbr_out[k][out] = in for every rule (in, out) of offset
// k (the rules of offset k are nbmaps[koff[k] .. koff[k] + nbsizes[k])), -1 elsewhere.  For a caller that
// holds what backend.convolution_forward_cuda(in, out, W, nbmaps, nbsizes, transposed) receives.
namespace {
__global__ void __launch_bounds__(256) kmap_from_rules_kernel(const int2* __restrict__ nbmaps,
                                                              const int* __restrict__ nbsizes, int K,
                                                              int64_t total, int64_t n_in, int64_t n_out,
                                                              int* __restrict__ nbr_out, int* __restrict__ bad) {
  __shared__ long long start[33];
  if (threadIdx.x == 0) {
    long long a = 0;
    for (int k = 0; k < K; ++k) { start[k] = a; a += nbsizes[k]; }
    start[K] = a;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (i >= start[K]) return;                 // capacity beyond the true rule count
  int k = 0;
  while (k + 1 < K && i >= start[k + 1]) ++k;
  const int2 r = nbmaps[i];
  if (r.x < 0 || r.x >= n_in || r.y < 0 || r.y >= n_out) { atomicAdd(bad, 1); return; }
  nbr_out[(int64_t)k * n_out + r.y] = r.x;
}
}  // namespace

extern "C" int lidal_kmap_from_rules(const int32_t* nbmaps, const int32_t* nbsizes, int k, int64_t n_rules,
                                     int64_t n_in, int64_t n_out, int32_t* nbr_out, int32_t* n_bad_dev,
                                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(k > 0 && k <= 32, "kmap_from_rules: kernel volume %d must be in 1..32", k);
  LIDAL_REQUIRE(n_rules >= 0 && n_in >= 0 && n_out >= 0, "kmap_from_rules: bad sizes");
  if (n_bad_dev != nullptr) LIDAL_HIP(hipMemsetAsync(n_bad_dev, 0, 4, s));
  if (n_out > 0) LIDAL_HIP(hipMemsetAsync(nbr_out, 0xFF, 4 * n_out * k, s));
  if (n_rules == 0 || n_out == 0) return 0;
  LIDAL_REQUIRE(n_bad_dev != nullptr, "kmap_from_rules: needs the i32 error counter");
  kmap_from_rules_kernel<<<(unsigned)cdiv(n_rules, 256), 256, 0, s>>>((const int2*)nbmaps, nbsizes, k, n_rules,
                                                                      n_in, n_out, nbr_out, n_bad_dev);
  LIDAL_CHECK_LAUNCH("lidal_kmap_from_rules");
  return 0;
}

// ---- all kernel maps of a network in one chain of launches ------------------------------------
// A U-Net builds 9 maps per step, each a chain of 4-6 small launches (fill, probe, count, scan, compact,
// sizes): 46 launches of ~5 us of latency each.  Here every stage is ONE launch over all maps, the map
// descriptors travelling by value in the kernel arguments; each map's results are exactly those of
// lidal_kmap_build (the same device functions, the same block -> rows assignment).
namespace {
constexpr int MAX_KMAP_JOBS = 12;
struct KmapBatch {
  unsigned long long* tkeys[MAX_KMAP_JOBS]; int* tvals[MAX_KMAP_JOBS]; unsigned long long tmask[MAX_KMAP_JOBS];
  unsigned* tbits[MAX_KMAP_JOBS]; unsigned* tsbits[MAX_KMAP_JOBS]; const int* thdr[MAX_KMAP_JOBS];
  const int4* coords[MAX_KMAP_JOBS]; const int* offsets[MAX_KMAP_JOBS];
  int* nbr[MAX_KMAP_JOBS]; int2* nbmaps[MAX_KMAP_JOBS]; int* nbsizes[MAX_KMAP_JOBS]; long long* koff[MAX_KMAP_JOBS];
  int* counts[MAX_KMAP_JOBS]; long long* offs[MAX_KMAP_JOBS];
  long long n_out[MAX_KMAP_JOBS], nblocks[MAX_KMAP_JOBS];
  long long probe0[MAX_KMAP_JOBS + 1];        // first block of map j in the probe launch (nblocks x k, or k/2 if symmetric)
  long long full0[MAX_KMAP_JOBS + 1];         // first block of map j in the count / compact launches (nblocks x k)
  long long fill0[MAX_KMAP_JOBS + 1];         // first element of map j in the fill launch (symmetric maps: the mirrored half)
  int k[MAX_KMAP_JOBS], sym[MAX_KMAP_JOBS], rules[MAX_KMAP_JOBS];
  int n_jobs;
};
__device__ __forceinline__ int kjob_of(const long long* first, int n_jobs, long long i) {
  int j = 0;
#pragma unroll 1
  for (int t = 1; t < n_jobs; ++t) j += (i >= first[t]) ? 1 : 0;
  return j;
}
__global__ void __launch_bounds__(256) kmap_fill_batch_kernel(KmapBatch b) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.fill0[b.n_jobs]) return;
  const int j = kjob_of(b.fill0, b.n_jobs, i);
  const int K = b.k[j];
  b.nbr[j][(long long)(K / 2 + 1) * b.n_out[j] + (i - b.fill0[j])] = -1;
}
__global__ void __launch_bounds__(kBlock) kmap_probe_batch_kernel(KmapBatch b) {
  const long long blk = blockIdx.x;
  const int j = kjob_of(b.probe0, b.n_jobs, blk);
  const long long l = blk - b.probe0[j];
  const long long bx = l % b.nblocks[j];
  const int k = (int)(l / b.nblocks[j]);
  TableView t;
  t.keys = b.tkeys[j]; t.vals = b.tvals[j]; t.mask = b.tmask[j]; t.bits = b.tbits[j]; t.sbits = b.tsbits[j]; t.hdr = b.thdr[j];
  if (b.sym[j] && table_is_spatial(t)) kmap_probe_sym_spatial_body(t, b.coords[j], b.n_out[j], b.offsets[j], b.k[j], b.nbr[j], bx, k);
  else if (b.sym[j]) kmap_probe_sym_body(t, b.coords[j], b.n_out[j], b.offsets[j], b.k[j], b.nbr[j], bx, k);
  else kmap_probe_body(t, b.coords[j], b.n_out[j], b.offsets[j], b.nbr[j], b.counts[j], b.nblocks[j], bx, k);
}
__global__ void __launch_bounds__(kBlock) kmap_count_batch_kernel(KmapBatch b) {
  const long long blk = blockIdx.x;
  const int j = kjob_of(b.full0, b.n_jobs, blk);
  if (!b.sym[j] || !b.rules[j]) return;           // counted by its probe / no rule lists wanted
  const long long l = blk - b.full0[j];
  kmap_count_body(b.nbr[j], b.n_out[j], b.counts[j], b.nblocks[j], l % b.nblocks[j], (int)(l / b.nblocks[j]));
}
__global__ void __launch_bounds__(1024) kmap_scan_batch_kernel(KmapBatch b) {
  const int j = blockIdx.x;
  if (!b.rules[j]) return;
  scan_counts_body(b.counts[j], b.nblocks[j] * b.k[j], (int64_t*)b.offs[j]);
}
__global__ void __launch_bounds__(kBlock) kmap_compact_batch_kernel(KmapBatch b) {
  const long long blk = blockIdx.x;
  const int j = kjob_of(b.full0, b.n_jobs, blk);
  if (!b.rules[j]) return;
  const long long l = blk - b.full0[j];
  kmap_compact_body(b.nbr[j], b.n_out[j], (const int64_t*)b.offs[j], b.nblocks[j], b.nbmaps[j], l % b.nblocks[j],
                    (int)(l / b.nblocks[j]));
}
__global__ void __launch_bounds__(64) kmap_sizes_batch_kernel(KmapBatch b) {
  const int j = blockIdx.x, k = threadIdx.x, K = b.k[j];
  if (!b.rules[j]) return;
  const long long* offsets = b.offs[j];
  const long long nblocks = b.nblocks[j];
  if (k < K) {
    const long long a = offsets[(long long)k * nblocks], e = offsets[(long long)(k + 1) * nblocks];
    b.nbsizes[j][k] = (int)(e - a);
    b.koff[j][k] = a;
  }
  if (k == K) b.koff[j][K] = offsets[(long long)K * nblocks];
}
}  // namespace

extern "C" int64_t lidal_kmap_build_batch_workspace_bytes(const int64_t* n_out, const int32_t* k, int n_jobs) {
  int64_t total = 256;
  for (int j = 0; j < n_jobs; ++j) total += lidal_kmap_workspace_bytes(n_out[j], k[j]);
  return total;
}

// Host arrays of length n_jobs (<= 12): the arguments of lidal_kmap_build per map.  nbmaps[j] == NULL: table only.
extern "C" int lidal_kmap_build_batch(const void* const* tables, const int64_t* table_bytes,
                                      const int32_t* const* out_coords, const int64_t* n_out,
                                      const int32_t* const* offsets, const int32_t* k, const int32_t* symmetric,
                                      int32_t* const* nbr_out, int32_t* const* nbmaps, int32_t* const* nbsizes,
                                      int64_t* const* koff, int n_jobs, void* ws, int64_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(n_jobs >= 0 && n_jobs <= MAX_KMAP_JOBS, "kmap_build_batch: at most %d maps", MAX_KMAP_JOBS);
  if (n_jobs == 0) return 0;
  LIDAL_REQUIRE(ws_bytes >= lidal_kmap_build_batch_workspace_bytes(n_out, k, n_jobs), "kmap_build_batch ws too small");
  KmapBatch b;
  memset(&b, 0, sizeof(b));
  b.n_jobs = n_jobs;
  char* w = (char*)ws;
  long long probe = 0, full = 0, fill = 0;
  bool any_rules = false, any_sym_rules = false;
  for (int j = 0; j < n_jobs; ++j) {
    LIDAL_REQUIRE(k[j] > 0 && k[j] < 64 && n_out[j] > 0, "kmap_build_batch: bad map %d (k=%d, rows=%lld)", j, k[j],
                  (long long)n_out[j]);
    const TableView t = table_view(tables[j], table_bytes[j]);
    b.tkeys[j] = t.keys; b.tvals[j] = t.vals; b.tmask[j] = t.mask; b.tbits[j] = t.bits; b.tsbits[j] = t.sbits; b.thdr[j] = t.hdr;
    b.coords[j] = (const int4*)out_coords[j]; b.offsets[j] = offsets[j];
    b.nbr[j] = nbr_out[j]; b.nbmaps[j] = (int2*)nbmaps[j]; b.nbsizes[j] = nbsizes[j]; b.koff[j] = (long long*)koff[j];
    b.n_out[j] = n_out[j]; b.k[j] = k[j];
    b.sym[j] = (symmetric[j] && (k[j] & 1) && k[j] >= 3) ? 1 : 0;
    b.rules[j] = nbmaps[j] != nullptr ? 1 : 0;
    any_rules |= b.rules[j] != 0;
    any_sym_rules |= b.rules[j] && b.sym[j];
    const int64_t nblocks = cdiv(n_out[j], kTile);
    b.nblocks[j] = nblocks;
    b.counts[j] = (int*)w;
    b.offs[j] = (long long*)(w + align_up(4 * nblocks * k[j], 256));
    w += lidal_kmap_workspace_bytes(n_out[j], k[j]);
    b.probe0[j] = probe; b.full0[j] = full; b.fill0[j] = fill;
    probe += nblocks * (b.sym[j] ? k[j] / 2 : k[j]);
    full += nblocks * k[j];
    fill += b.sym[j] ? n_out[j] * (k[j] / 2) : 0;
  }
  for (int j = n_jobs; j <= MAX_KMAP_JOBS; ++j) { b.probe0[j] = probe; b.full0[j] = full; b.fill0[j] = fill; }
  if (fill > 0) {
    kmap_fill_batch_kernel<<<(unsigned)cdiv(fill, 256), 256, 0, s>>>(b);
    LIDAL_CHECK_LAUNCH("kmap_fill_batch");
  }
  kmap_probe_batch_kernel<<<(unsigned)probe, kBlock, 0, s>>>(b);
  LIDAL_CHECK_LAUNCH("kmap_probe_batch");
  if (!any_rules) return 0;
  if (any_sym_rules) {
    kmap_count_batch_kernel<<<(unsigned)full, kBlock, 0, s>>>(b);
    LIDAL_CHECK_LAUNCH("kmap_count_batch");
  }
  kmap_scan_batch_kernel<<<(unsigned)n_jobs, 1024, 0, s>>>(b);
  LIDAL_CHECK_LAUNCH("kmap_scan_batch");
  kmap_compact_batch_kernel<<<(unsigned)full, kBlock, 0, s>>>(b);
  LIDAL_CHECK_LAUNCH("kmap_compact_batch");
  kmap_sizes_batch_kernel<<<(unsigned)n_jobs, 64, 0, s>>>(b);
  LIDAL_CHECK_LAUNCH("kmap_sizes_batch");
  return 0;
}

extern "C" int lidal_kmap_invert(const int32_t* nbr_out, int64_t n_out, int k, int32_t* nbr_in,
                                 int64_t n_in, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_in > 0) LIDAL_HIP(hipMemsetAsync(nbr_in, 0xFF, 4 * n_in * k, s));
  if (n_out == 0 || n_in == 0) return 0;
  kmap_invert_kernel<<<dim3((unsigned)cdiv(n_out, 256), (unsigned)k), 256, 0, s>>>(
      nbr_out, n_out, k, nbr_in, n_in);
  LIDAL_CHECK_LAUNCH("kmap_invert");
  return 0;
}

static int64_t order_sort_tmp_bytes(int64_t q) { return sort_pairs_ws_bytes(q); }

extern "C" int64_t lidal_kmap_order_workspace_bytes(int64_t n_rows) {
  int64_t q = n_rows > 0 ? n_rows : 1;
  return 4 * align_up(4 * q, 256) + align_up(order_sort_tmp_bytes(q), 256) + 256;
}

extern "C" int lidal_kmap_order(const int32_t* nbr, int64_t n_rows, int k, int32_t* perm,
                                int32_t* nbr_perm, uint32_t* tile_masks, void* ws,
                                int64_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(k > 0 && k <= 32, "kmap_order: kernel volume %d must be <= 32", k);
  if (n_rows == 0) return 0;
  LIDAL_REQUIRE(ws_bytes >= lidal_kmap_order_workspace_bytes(n_rows), "kmap_order ws too small");
  int64_t q = n_rows, a = align_up(4 * q, 256);
  unsigned* keys = (unsigned*)ws;
  unsigned* skeys = (unsigned*)((char*)ws + a);
  int* vals = (int*)((char*)ws + 2 * a);
  void* tmp = (char*)ws + 3 * a;
  const BitRank rank = bit_rank(k);
  row_mask_kernel<<<(unsigned)cdiv(q, 256), 256, 0, s>>>(nbr, q, k, rank, keys, vals);
  LIDAL_CHECK_LAUNCH("row_mask");
  if (int rc = sort_pairs_u32(keys, vals, skeys, perm, q, k, tmp, order_sort_tmp_bytes(q), s)) return rc;
  permute_table_kernel<<<dim3((unsigned)cdiv(q, 256), (unsigned)k), 256, 0, s>>>(nbr, q, perm,
                                                                                 nbr_perm);
  LIDAL_CHECK_LAUNCH("permute_table");
  if (tile_masks != nullptr) {
    int64_t tiles = cdiv(q, 128);
    tile_or_kernel<<<(unsigned)cdiv(tiles, 4), 256, 0, s>>>(skeys, q, rank, tile_masks, tiles);
    LIDAL_CHECK_LAUNCH("tile_or");
  }
  return 0;
}

// ---- the row orders of SEVERAL tables of one kernel volume in one go ---------------------------
// A U-Net builds 5 3x3x3 maps and 4 2x2x2 maps (the latter ordered in both directions): 13 row orders
// per step, each a chain of row_mask -> sort (hist + ceil(k/8) passes) -> permute -> tile_or on a few
// 10^4 .. 10^5 rows, i.e. launch latency.  Here the rows of all tables of a volume are ONE key array,
// key = table index << k | Gray-ranked mask, sorted once; every kernel takes the table descriptors by
// value (no device-side descriptor buffer, nothing to upload).
namespace {
constexpr int MAX_ORDER_JOBS = 16;
struct OrderBatch {
  const int* nbr[MAX_ORDER_JOBS];
  int* perm[MAX_ORDER_JOBS];
  int* nbr_perm[MAX_ORDER_JOBS];
  unsigned* tmask[MAX_ORDER_JOBS];
  long long row0[MAX_ORDER_JOBS + 1];       // first row of table j in the concatenated arrays
  long long tile0[MAX_ORDER_JOBS + 1];      // first 128-row tile of table j
  int n_jobs, k;
};
__device__ __forceinline__ int job_of(const long long* first, int n_jobs, long long i) {
  int j = 0;
#pragma unroll 1
  for (int t = 1; t < n_jobs; ++t) j += (i >= first[t]) ? 1 : 0;
  return j;
}

__global__ void __launch_bounds__(256) row_mask_batch_kernel(OrderBatch b, BitRank rank,
                                                             unsigned* __restrict__ keys,
                                                             int* __restrict__ vals) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.row0[b.n_jobs]) return;
  const int j = job_of(b.row0, b.n_jobs, i);
  const long long r = i - b.row0[j], n = b.row0[j + 1] - b.row0[j];
  const int* nbr = b.nbr[j];
  unsigned m = 0u;
  for (int k = 0; k < b.k; ++k)
    if (nbr[(long long)k * n + r] >= 0) m |= 1u << rank.to_key[k];
  m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
  keys[i] = m | ((unsigned)j << b.k);
  vals[i] = (int)r;
}

__global__ void __launch_bounds__(256) permute_table_batch_kernel(OrderBatch b, const int* __restrict__ svals) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= b.row0[b.n_jobs]) return;
  const int j = job_of(b.row0, b.n_jobs, i);           // sorted by table first: table j's rows sit at [row0, row0 + n)
  const long long r = i - b.row0[j], n = b.row0[j + 1] - b.row0[j];
  const int src = svals[i];
  if (k == 0) b.perm[j][r] = src;
  b.nbr_perm[j][(long long)k * n + r] = b.nbr[j][(long long)k * n + src];
}

__global__ void __launch_bounds__(256) tile_or_batch_kernel(OrderBatch b, BitRank rank,
                                                            const unsigned* __restrict__ skeys) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= b.tile0[b.n_jobs]) return;
  const int j = job_of(b.tile0, b.n_jobs, t);
  if (b.tmask[j] == nullptr) return;
  const long long lt = t - b.tile0[j], n = b.row0[j + 1] - b.row0[j];
  const unsigned kmask = (b.k >= 32) ? ~0u : ((1u << b.k) - 1u);
  unsigned m = 0u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const long long r = lt * 128 + h * 64 + lane;
    if (r < n) { const unsigned g = skeys[b.row0[j] + r] & kmask; m |= g ^ (g >> 1); }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, 64);
  if (lane == 0) {
    unsigned real = 0u;
    for (int bit = 0; bit < 32; ++bit)
      if ((m >> bit) & 1u) real |= 1u << rank.to_mask[bit];
    b.tmask[j][lt] = real;
  }
}
}  // namespace

extern "C" int lidal_kmap_order_batch(const int32_t* const* nbr, const int64_t* n_rows, int n_jobs, int k,
                                      int32_t* const* perm, int32_t* const* nbr_perm,
                                      uint32_t* const* tile_masks, void* ws, int64_t ws_bytes,
                                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(n_jobs >= 0 && n_jobs <= MAX_ORDER_JOBS, "kmap_order_batch: at most %d tables", MAX_ORDER_JOBS);
  int jbits = 0;
  while ((1 << jbits) < n_jobs) ++jbits;
  LIDAL_REQUIRE(k > 0 && k + jbits <= 32, "kmap_order_batch: kernel volume %d with %d tables does not fit a 32-bit key",
                k, n_jobs);
  OrderBatch b;
  memset(&b, 0, sizeof(b));
  b.n_jobs = n_jobs; b.k = k;
  long long rows = 0, tiles = 0;
  for (int j = 0; j < n_jobs; ++j) {
    LIDAL_REQUIRE(n_rows[j] >= 0, "kmap_order_batch: bad row count");
    b.nbr[j] = nbr[j]; b.perm[j] = perm[j]; b.nbr_perm[j] = nbr_perm[j];
    b.tmask[j] = tile_masks ? tile_masks[j] : nullptr;
    b.row0[j] = rows; b.tile0[j] = tiles;
    rows += n_rows[j]; tiles += cdiv(n_rows[j], 128);
  }
  for (int j = n_jobs; j <= MAX_ORDER_JOBS; ++j) { b.row0[j] = rows; b.tile0[j] = tiles; }
  if (rows == 0) return 0;
  LIDAL_REQUIRE(ws_bytes >= lidal_kmap_order_workspace_bytes(rows), "kmap_order_batch ws too small");
  const int64_t a = align_up(4 * rows, 256);
  unsigned* keys = (unsigned*)ws;
  unsigned* skeys = (unsigned*)((char*)ws + a);
  int* vals = (int*)((char*)ws + 2 * a);
  // the sorted row ids land in the first table's perm-sized scratch? no: perm arrays are per table, so
  // the sorted ids go to a scratch slice and permute_table_batch distributes them
  int* svals = (int*)((char*)ws + 3 * a);
  void* tmp = (char*)ws + 4 * a;
  const BitRank rank = bit_rank(k);
  row_mask_batch_kernel<<<(unsigned)cdiv(rows, 256), 256, 0, s>>>(b, rank, keys, vals);
  LIDAL_CHECK_LAUNCH("row_mask_batch");
  if (int rc = sort_pairs_u32(keys, vals, skeys, svals, rows, k + jbits, tmp, order_sort_tmp_bytes(rows), s)) return rc;
  permute_table_batch_kernel<<<dim3((unsigned)cdiv(rows, 256), (unsigned)k), 256, 0, s>>>(b, svals);
  LIDAL_CHECK_LAUNCH("permute_table_batch");
  tile_or_batch_kernel<<<(unsigned)cdiv(tiles, 4), 256, 0, s>>>(b, rank, skeys);
  LIDAL_CHECK_LAUNCH("tile_or_batch");
  return 0;
}

// ---- input voxelisation -------------------------------------------------------------------------
extern "C" int64_t lidal_voxelize_points_workspace_bytes(int64_t p) {
  int64_t q = p > 0 ? p : 1;
  int64_t blocks = cdiv(q, 256), nb = cdiv(q, kTile);
  return align_up(24 * q, 256)              /* scaled f64 [p,3] */
         + align_up(48 * blocks, 256)       /* min/max partials */
         + 256                              /* offset[3], n_invalid */
         + 2 * align_up(8 * q, 256)         /* keys, sorted keys */
         + 2 * align_up(4 * q, 256)         /* ids, sorted ids */
         + align_up(4 * nb, 256) + align_up(8 * (nb + 1), 256)
         + align_up((int64_t)voxel_sort_tmp_bytes(q), 256) + 256;
}

extern "C" int lidal_voxelize_points(const float* points, const float* intensity, int64_t p,
                                     const double* m_dev, const double* rnd_dev, double scale,
                                     int full_scale, float* feats_p, int32_t* coords_v,
                                     int64_t* unique_idx, int64_t* inverse, int64_t* n_out_dev,
                                     int32_t* n_invalid_dev, void* ws, int64_t ws_bytes,
                                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(full_scale > 0 && full_scale <= 8192, "voxelize_points: full_scale must be <= 8192");
  LIDAL_HIP(hipMemsetAsync(n_invalid_dev, 0, 4, s));
  if (p == 0) {
    LIDAL_HIP(hipMemsetAsync(n_out_dev, 0, 8, s));
    return 0;
  }
  LIDAL_REQUIRE(ws_bytes >= lidal_voxelize_points_workspace_bytes(p), "voxelize_points ws too small");
  int64_t blocks = cdiv(p, 256), nb = cdiv(p, kTile);
  char* w = (char*)ws;
  double* scaled = (double*)w;      w += align_up(24 * p, 256);
  double* part = (double*)w;        w += align_up(48 * blocks, 256);
  double* offset = (double*)w;      w += 256;
  uint64_t* keys = (uint64_t*)w;    w += align_up(8 * p, 256);
  uint64_t* skeys = (uint64_t*)w;   w += align_up(8 * p, 256);
  int* ids = (int*)w;               w += align_up(4 * p, 256);
  int* sids = (int*)w;              w += align_up(4 * p, 256);
  int* counts = (int*)w;            w += align_up(4 * nb, 256);
  int64_t* offs = (int64_t*)w;      w += align_up(8 * (nb + 1), 256);
  void* tmp = (void*)w;
  affine_kernel<<<(unsigned)blocks, 256, 0, s>>>(points, intensity, p, m_dev, scale, feats_p, scaled, part);
  LIDAL_CHECK_LAUNCH("affine");
  voxel_offset_kernel<<<1, 256, 0, s>>>(part, blocks, rnd_dev, (double)full_scale, offset);
  LIDAL_CHECK_LAUNCH("voxel_offset");
  voxel_keys_kernel<<<(unsigned)blocks, 256, 0, s>>>(scaled, p, offset, full_scale, keys, ids, n_invalid_dev);
  LIDAL_CHECK_LAUNCH("voxel_keys");
  if (int rc = radix_sort(keys, ids, skeys, sids, p, 8, 39, tmp, (int64_t)voxel_sort_tmp_bytes(p), s)) return rc;
  head_count_kernel<<<(unsigned)nb, kBlock, 0, s>>>(skeys, p, counts);
  LIDAL_CHECK_LAUNCH("head_count");
  scan_counts_kernel<<<1, 1024, 0, s>>>(counts, nb, offs);
  LIDAL_CHECK_LAUNCH("scan_counts");
  rows_compact_kernel<<<(unsigned)nb, kBlock, 0, s>>>(skeys, sids, p, offs, nb, coords_v, unique_idx,
                                                      inverse, n_out_dev);
  LIDAL_CHECK_LAUNCH("rows_compact");
  return 0;
}
