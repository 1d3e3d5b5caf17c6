// Stable LSD radix sort for gfx950 -- the one sort of this library (kernel-map row orders, sorted
// unique of coordinate hashes and packed coordinates, point->voxel contributor lists, the scorer's
// cell keys).  Keys u32 or u64, optional i32 payload, 8 bits per pass, 1 + ceil(bits / 8) launches:
//
//   hist     every digit position's histogram in ONE pass over the keys: <= 256 workgroups, each with
//            private LDS tables written out as part[block][pass][256] (no global atomics, nothing to
//            zero beforehand; more than 32 tables are added up by one more small launch); the same
//            launch zeroes the look-back state of all passes.
//   pass p   "one sweep": a tile of 8192 consecutive items per workgroup (512 threads x 16), taken in
//            order through an atomic ticket.  The tile's items are ranked in index order -- per wave a
//            multisplit by 8 ballots per item, the waves' counts prefixed through LDS -- which gives
//            the tile's digit counts; those are published (AGGREGATE) in status[pass][tile][256], the
//            counts of the tiles before it are collected by decoupled look-back (thread d walks digit d
//            backwards, 8 status rows in flight, until it meets an INCLUSIVE entry) and the tile's own
//            inclusive counts are published.  Items go through an LDS stage in tile-sorted order, so
//            the write-out moves runs (32 items per digit on average), not single words.
//
// Stable, hence unique: bit-equal to torch.sort(stable=True) (tests/test_ops_gpu.py).  The first
// generation of this file (per-range digit tables, next digit counted where an item lands) won only
// its one-pass form against rocPRIM (168 vs 114 us for 397k 27-bit pairs): its passes scattered single
// 4-byte writes and counted with ~400k contended global atomics per pass; this form has neither.
#include "common.h"

using namespace lidal;

namespace {

constexpr int RADIX = 256;
constexpr int THREADS = 512;
constexpr int WAVES = THREADS / 64;
constexpr int ITEMS = 16;
constexpr int TILE = THREADS * ITEMS;            // 8192 items per workgroup
constexpr int MAX_PASSES = 8;
constexpr int HIST_BLOCKS = 256;         // partial tables of the histogram launch, at most
constexpr int HIST_DIRECT = 32;          // up to this many, the passes add the partial tables up themselves

constexpr unsigned FLAG_AGG = 1u << 30, FLAG_INC = 2u << 30, FLAG_MASK = 3u << 30, COUNT_MASK = (1u << 30) - 1u;

struct Layout {
  int passes, hist_blocks;
  int64_t tiles;
  int64_t off_part, off_total, off_ticket, off_status, off_ktmp, off_vtmp, total;
};

Layout layout_for(int64_t n, int key_bytes, bool has_val, int end_bit) {
  Layout L;
  const int64_t q = n > 0 ? n : 1;
  L.passes = (end_bit + 7) / 8;
  if (L.passes < 1) L.passes = 1;
  if (L.passes > MAX_PASSES) L.passes = MAX_PASSES;
  L.tiles = cdiv(q, TILE);
  L.hist_blocks = (int)(L.tiles < HIST_BLOCKS ? L.tiles : HIST_BLOCKS);
  int64_t o = 0;
  L.off_part = o;   o += align_up((int64_t)HIST_BLOCKS * MAX_PASSES * RADIX * 4, 256);
  L.off_total = o;  o += align_up((int64_t)MAX_PASSES * RADIX * 4, 256);
  L.off_ticket = o; o += 256;
  L.off_status = o; o += align_up((int64_t)MAX_PASSES * L.tiles * RADIX * 4, 256);
  L.off_ktmp = o;   o += align_up(q * key_bytes, 256);
  L.off_vtmp = o;   o += has_val ? align_up(q * 4, 256) : 0;
  L.total = o;
  return L;
}

__device__ __forceinline__ unsigned ld_status(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- all digit histograms in one pass; zeroes tickets and look-back state -------------------
// (n_dev != NULL: the item count lives on the device -- min(*n_dev, n); the launch is sized for n)
__device__ __forceinline__ int64_t live_count(int64_t n, const int64_t* n_dev) {
  if (n_dev == nullptr) return n;
  const int64_t v = *n_dev;
  return v < 0 ? 0 : (v < n ? v : n);
}

template <typename K>
__global__ void __launch_bounds__(THREADS) sort_hist_kernel(const K* __restrict__ keys, int64_t n_cap, int passes,
                                                            int end_bit, unsigned* __restrict__ part,
                                                            unsigned* __restrict__ zero_from, int64_t zero_words,
                                                            const int64_t* __restrict__ n_dev) {
  __shared__ unsigned h[MAX_PASSES][RADIX];
  const int tid = threadIdx.x;
  const int64_t n = live_count(n_cap, n_dev);
  for (int i = tid; i < MAX_PASSES * RADIX; i += THREADS) (&h[0][0])[i] = 0u;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < zero_words; i += (int64_t)gridDim.x * THREADS)
    zero_from[i] = 0u;
  __syncthreads();
  // contiguous share of the items per block, 16-item strips per thread
  const int64_t per = align_up(cdiv(n, gridDim.x), THREADS);
  const int64_t beg = (int64_t)blockIdx.x * per, end = (beg + per < n) ? beg + per : n;
  const K top_mask = (end_bit >= (int)sizeof(K) * 8) ? ~(K)0 : (((K)1 << end_bit) - 1);
  for (int64_t i0 = beg; i0 < end; i0 += (int64_t)THREADS * 4) {
    K k[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * THREADS + tid;
      ok[u] = i < end;
      k[u] = ok[u] ? keys[i] : (K)0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const K kk = k[u] & top_mask;
      const unsigned long long live = __ballot(ok[u]);
      for (int p = 0; p < passes; ++p) {
        // Real keys are skewed -- packed coordinates and occupancy masks have constant high bytes,
        // consecutive points of a contributor list fall into one voxel -- and 64 lanes adding to ONE LDS
        // counter serialise (measured: 35-45 us for this kernel on a step's lists against 4 us on uniform
        // keys).  So the lanes that share the first lane's digit are counted by that lane alone; only
        // the others add one by one.  (Peeling up to four values in a loop cost more scalar work than it
        // saved: +37 us on 397k uniform keys.)
        const unsigned d = (unsigned)(kk >> (8 * p)) & 255u;
        unsigned long long rem = live;
        if (live != 0ull) {
          const int leader = __builtin_ctzll(live);
          const unsigned f = (unsigned)__builtin_amdgcn_readlane((int)d, leader);
          const unsigned long long same = __ballot(d == f) & live;
          if ((tid & 63) == leader) atomicAdd(&h[p][f], (unsigned)__popcll(same));
          rem &= ~same;
        }
        if ((rem >> (tid & 63)) & 1ull) atomicAdd(&h[p][d], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < passes * RADIX; i += THREADS)
    part[(int64_t)blockIdx.x * MAX_PASSES * RADIX + i] = (&h[0][0])[i];
}

// many partial tables (large lists: one histogram workgroup per CU) are added up once, not by every tile of
// every pass: total[p][d] = sum_b part[b][p][d]
__global__ void __launch_bounds__(RADIX) sort_hist_reduce_kernel(const unsigned* __restrict__ part, int blocks,
                                                                 unsigned* __restrict__ total) {
  const int p = blockIdx.x, d = threadIdx.x;
  unsigned t = 0;
  for (int b0 = 0; b0 < blocks; b0 += 16) {
    unsigned c[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u] = (b0 + u < blocks) ? part[((int64_t)(b0 + u) * MAX_PASSES + p) * RADIX + d] : 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += c[u];
  }
  total[p * RADIX + d] = t;
}

// exclusive scan of one value per thread over the first 256 threads (4 waves); all threads call it
__device__ __forceinline__ unsigned scan256_exclusive(unsigned v, unsigned* wsum /*[4] LDS*/, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63 && wave < 4) wsum[wave] = incl;
  __syncthreads();
  unsigned off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) off += (w < wave) ? wsum[w] : 0u;
  __syncthreads();
  return off + incl - v;
}

template <typename K, bool HAS_VAL>
__global__ void __launch_bounds__(THREADS) sort_onesweep_kernel(const K* __restrict__ kin, const int* __restrict__ vin,
                                                                K* __restrict__ kout, int* __restrict__ vout,
                                                                int64_t n_cap, int shift, int bits,
                                                                const unsigned* __restrict__ part, int hist_blocks,
                                                                unsigned* __restrict__ ticket,
                                                                unsigned* __restrict__ status,
                                                                const int64_t* __restrict__ n_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];
  K* skey = reinterpret_cast<K*>(stage_raw);
  int* sval = reinterpret_cast<int*>(stage_raw + (size_t)TILE * sizeof(K));
  __shared__ unsigned whist[WAVES][RADIX];
  __shared__ unsigned tstart[RADIX];        // first tile-sorted position of digit d
  __shared__ unsigned goff[RADIX];          // global position = tile-sorted position + goff[d]   (mod 2^32)
  __shared__ unsigned wsum[4];
  __shared__ unsigned tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) tile_s = atomicAdd(ticket, 1u);
  for (int i = tid; i < WAVES * RADIX; i += THREADS) (&whist[0][0])[i] = 0u;
  __syncthreads();
  const int64_t n = live_count(n_cap, n_dev);
  const int64_t tile = tile_s;
  const int64_t base = tile * TILE;
  if (base >= n) return;                 // (a device-side count: the tiles past it have nothing to do -- no tile waits on them)
  const int64_t left = n - base;
  const int nvalid = left >= TILE ? TILE : (int)left;
  const unsigned dmask = (1u << bits) - 1u;

  // ---- load: wave w owns items [w * 1024, w * 1024 + 1024) of the tile, item (i, lane) = i * 64 + lane
  K key[ITEMS];
  int val[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int li = wave * (64 * ITEMS) + i * 64 + lane;
    const bool ok = li < nvalid;
    key[i] = ok ? kin[base + li] : ~(K)0;
    if (HAS_VAL) val[i] = ok ? vin[base + li] : 0;
  }
  // ---- rank inside the wave, items in index order.  Items past the end carry the all-ones digit and
  // sit behind every valid item of that digit (they are the last items of the last tile)
  unsigned short lrank[ITEMS];
  unsigned char dig[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned d = ((unsigned)(key[i] >> shift)) & dmask;
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b < bits) {
        const bool set = (d >> b) & 1u;
        const unsigned long long bb = __ballot(set);
        m &= set ? bb : ~bb;
      }
    }
    const int r = __popcll(m & ((1ull << lane) - 1ull));
    unsigned before = 0;
    if (r == 0) {                       // one lane per digit and step; the wave's steps run in order
      before = whist[wave][d];
      whist[wave][d] = before + (unsigned)__popcll(m);
    }
    before = __shfl(before, __builtin_ctzll(m), 64);
    lrank[i] = (unsigned short)(before + r);
    dig[i] = (unsigned char)d;
  }
  __syncthreads();
  // ---- per digit: prefix over the waves, tile count, tile start, global start
  unsigned cnt = 0;
  if (tid < RADIX) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const unsigned c = whist[w][tid];
      whist[w][tid] = cnt;
      cnt += c;
    }
  }
  unsigned valid_cnt = cnt;
  if (tid == (int)dmask && nvalid < TILE) valid_cnt -= (unsigned)(TILE - nvalid);      // the padding is not data
  const unsigned ts = scan256_exclusive(tid < RADIX ? cnt : 0u, wsum, tid);
  unsigned total = 0;
  if (tid < RADIX) {
    for (int b0 = 0; b0 < hist_blocks; b0 += 16) {
      unsigned c[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        c[u] = (b0 + u < hist_blocks) ? part[(int64_t)(b0 + u) * MAX_PASSES * RADIX + tid] : 0u;
#pragma unroll
      for (int u = 0; u < 16; ++u) total += c[u];
    }
  }
  const unsigned gbase = scan256_exclusive(tid < RADIX ? total : 0u, wsum, tid);
  if (tid < RADIX) {
    unsigned* st = status + tile * RADIX + tid;
    st_status(st, (tile == 0 ? FLAG_INC : FLAG_AGG) | valid_cnt);
    unsigned excl = 0;
    int64_t t = tile - 1;
    bool done = tile == 0;
    while (!done) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (t - u >= 0) ? ld_status(status + (t - u) * RADIX + tid) : FLAG_INC;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (done) break;
        while ((v[u] & FLAG_MASK) == 0u) v[u] = ld_status(status + (t - u) * RADIX + tid);
        excl += v[u] & COUNT_MASK;
        done = (v[u] & FLAG_INC) != 0u;
      }
      t -= 8;
    }
    if (tile != 0) st_status(st, FLAG_INC | (excl + valid_cnt));
    tstart[tid] = ts;
    goff[tid] = gbase + excl - ts;
  }
  __syncthreads();
  // ---- tile-sorted order in LDS, then runs to their global places
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned d = dig[i];
    const unsigned pos = tstart[d] + whist[wave][d] + lrank[i];
    skey[pos] = key[i];
    if (HAS_VAL) sval[pos] = val[i];
  }
  __syncthreads();
#pragma unroll 4
  for (int j = tid; j < nvalid; j += THREADS) {
    const K k = skey[j];
    const unsigned d = ((unsigned)(k >> shift)) & dmask;
    const unsigned pos = (unsigned)j + goff[d];
    kout[pos] = k;
    if (HAS_VAL) vout[pos] = sval[j];
  }
}

template <typename K, bool HAS_VAL>
int run_sort(const K* keys_in, const int* vals_in, K* keys_out, int* vals_out, int64_t n, int end_bit,
             void* ws, const Layout& L, hipStream_t s, const int64_t* n_dev) {
  char* w = (char*)ws;
  unsigned* part = (unsigned*)(w + L.off_part);
  unsigned* ticket = (unsigned*)(w + L.off_ticket);
  unsigned* status = (unsigned*)(w + L.off_status);
  K* ktmp = (K*)(w + L.off_ktmp);
  int* vtmp = HAS_VAL ? (int*)(w + L.off_vtmp) : nullptr;
  const int64_t zero_words = (L.off_status - L.off_ticket) / 4 + (int64_t)L.passes * L.tiles * RADIX;   // tickets + the passes' status
  sort_hist_kernel<K><<<L.hist_blocks, THREADS, 0, s>>>(keys_in, n, L.passes, end_bit, part, ticket, zero_words, n_dev);
  LIDAL_CHECK_LAUNCH("sort_hist");
  int hist_rows = L.hist_blocks;
  if (L.hist_blocks > HIST_DIRECT) {
    unsigned* total = (unsigned*)(w + L.off_total);
    sort_hist_reduce_kernel<<<L.passes, RADIX, 0, s>>>(part, L.hist_blocks, total);
    LIDAL_CHECK_LAUNCH("sort_hist_reduce");
    part = total;
    hist_rows = 1;
  }
  const size_t lds = (size_t)TILE * (sizeof(K) + (HAS_VAL ? 4 : 0));
  auto kern = sort_onesweep_kernel<K, HAS_VAL>;
  static size_t attr_set[MAX_DEVICES] = {};
  const int dev = current_device();
  if (attr_set[dev] < lds) {
    LIDAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = lds;
  }
  const K* kin = keys_in;
  const int* vin = vals_in;
  for (int p = 0; p < L.passes; ++p) {
    const bool to_out = ((L.passes - 1 - p) & 1) == 0;       // the last pass lands in the caller's buffers
    K* ko = to_out ? keys_out : ktmp;
    int* vo = to_out ? vals_out : vtmp;
    int bits = end_bit - 8 * p;
    if (bits > 8) bits = 8;
    if (bits < 1) bits = 1;
    kern<<<(unsigned)L.tiles, THREADS, lds, s>>>(kin, vin, ko, vo, n, 8 * p, bits, part + (int64_t)p * RADIX,
                                                 hist_rows, ticket + p, status + (int64_t)p * L.tiles * RADIX, n_dev);
    LIDAL_CHECK_LAUNCH("sort_onesweep");
    kin = ko;
    vin = vo;
  }
  return 0;
}

}  // namespace

namespace lidal {

int64_t radix_sort_ws_bytes(int64_t n, int key_bytes, bool has_val) {
  return layout_for(n, key_bytes, has_val, 8 * key_bytes).total;
}

// keys_in / vals_in are not written; keys_out (and vals_out if vals_in != NULL) receive the pairs sorted
// by key bits [0, end_bit) -- the higher bits are ignored by the order but travel with the key.
// n_dev (device i64, may be NULL): the live item count min(*n_dev, n) is only known on the device -- the launches and the
// workspace are sized for n, the tiles past the live count exit; items past it are neither read nor written.
int radix_sort(const void* keys_in, const int* vals_in, void* keys_out, int* vals_out, int64_t n,
               int key_bytes, int end_bit, void* ws, int64_t ws_bytes, hipStream_t s, const int64_t* n_dev) {
  if (n == 0) return 0;
  LIDAL_REQUIRE(key_bytes == 4 || key_bytes == 8, "sort: keys of %d bytes", key_bytes);
  LIDAL_REQUIRE(end_bit >= 1 && end_bit <= 8 * key_bytes && n < (1ll << 30), "sort: %d bits, %lld items", end_bit,
                (long long)n);
  const bool has_val = vals_in != nullptr;
  const Layout L = layout_for(n, key_bytes, has_val, end_bit);
  LIDAL_REQUIRE(ws_bytes >= L.total, "sort workspace too small: %lld < %lld", (long long)ws_bytes, (long long)L.total);
  if (key_bytes == 4)
    return has_val ? run_sort<unsigned, true>((const unsigned*)keys_in, vals_in, (unsigned*)keys_out, vals_out, n,
                                              end_bit, ws, L, s, n_dev)
                   : run_sort<unsigned, false>((const unsigned*)keys_in, nullptr, (unsigned*)keys_out, nullptr, n,
                                               end_bit, ws, L, s, n_dev);
  return has_val ? run_sort<unsigned long long, true>((const unsigned long long*)keys_in, vals_in,
                                                      (unsigned long long*)keys_out, vals_out, n, end_bit, ws, L, s, n_dev)
                 : run_sort<unsigned long long, false>((const unsigned long long*)keys_in, nullptr,
                                                       (unsigned long long*)keys_out, nullptr, n, end_bit, ws, L, s, n_dev);
}

int64_t sort_pairs_ws_bytes(int64_t n) { return radix_sort_ws_bytes(n, 4, true); }

int sort_pairs_u32(const unsigned* keys_in, const int* vals_in, unsigned* keys_out, int* vals_out,
                   int64_t n, int bits, void* ws, int64_t ws_bytes, hipStream_t s, const int64_t* n_dev) {
  return radix_sort(keys_in, vals_in, keys_out, vals_out, n, 4, bits, ws, ws_bytes, s, n_dev);
}

}  // namespace lidal

// test / bench entries: stable sort of device (key, value) pairs by the low `bits` bits of the keys
extern "C" int64_t lidal_sort_pairs_workspace_bytes(int64_t n) { return radix_sort_ws_bytes(n, 8, true); }
extern "C" int lidal_sort_pairs(const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out,
                                int32_t* vals_out, int64_t n, int bits, void* ws, int64_t ws_bytes,
                                void* stream) {
  return radix_sort(keys_in, vals_in, keys_out, vals_out, n, 4, bits, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int lidal_sort_pairs_u64(const uint64_t* keys_in, const int32_t* vals_in, uint64_t* keys_out,
                                    int32_t* vals_out, int64_t n, int bits, void* ws, int64_t ws_bytes,
                                    void* stream) {
  return radix_sort(keys_in, vals_in, keys_out, vals_out, n, 8, bits, ws, ws_bytes, (hipStream_t)stream);
}
