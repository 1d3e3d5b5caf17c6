// Shared helpers for liblidal_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lidal_amd.h"

namespace lidal {

void set_error(const char* fmt, ...);

#define LIDAL_CHECK_LAUNCH(name)                                             \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      lidal::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

#define LIDAL_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      lidal::set_error(__VA_ARGS__);  \
      return 2;                       \
    }                                 \
  } while (0)

#define LIDAL_HIP(call)                                                       \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      lidal::set_error("%s failed: %s", #call, hipGetErrorString(e__));       \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__host__ __device__ static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

constexpr int kWave = 64;

// ---- FNV-1a-64 folded to 60 bits (torchsparse backend/hash) ----
__device__ __forceinline__ int64_t fnv60(int32_t x, int32_t y, int32_t z, int32_t b) {
  uint64_t h = 14695981039346656037ULL;
  h ^= (uint32_t)x; h *= 1099511628211ULL;
  h ^= (uint32_t)y; h *= 1099511628211ULL;
  h ^= (uint32_t)z; h *= 1099511628211ULL;
  h ^= (uint32_t)b; h *= 1099511628211ULL;
  h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
  return (int64_t)h;
}

// ---- open-addressing hash table: slots of {key u64, val i32} in two arrays, + an occupancy bitmap ----
// Layout of a table of capacity cap = 2^k >= 2 n: keys u64 [cap] | vals i32 [cap] | bits u32 [cap / 4] (round 5).
// The bitmap has 8 bits per slot (16 per key at the fullest); key K sets bit (mixed(K) >> 32) & (8 cap - 1) -- other
// bits of the mixed key than its slot.  A look-up whose answer is mostly "absent" (the kernel-map probes: 11 of the 13
// neighbours probed per voxel do not exist) tests the bit first: 1-2 MB that stay in the L2s instead of a slot of a
// 12-25 MB table beyond them, and a clear bit IS the answer (no false negatives; ~6 % of the absent keys pass and are
// resolved by the slots as before).  Same results by construction.  A view without a bitmap (bits == NULL: the
// scorer's cell tables, tables of the old 12-byte size) skips the test.
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFULL;

struct TableView {
  unsigned long long* keys;
  int* vals;
  uint64_t mask;
  unsigned* bits;
  unsigned* sbits;        // spatial occupancy bitmap (below), or NULL
  const int* hdr;         // {magic, log2 stride, x bits, y bits} behind the bitmaps, or NULL
};

// A table built from COORDINATES (lidal_hash_table_build_coords, round 5) also carries a SPATIAL occupancy bitmap: bit
// ((b & 3) * 32 + (z' & 31)) * 2^yb + (y' & (2^yb - 1))) * 2^xb + (x' & (2^xb - 1)), x' = x >> log2(stride): direct
// mapped, aliased (a 1 MB map holds 256 x 256 x 32 x 4 cells; ~5 % of the bits of a LiDAR level are set), x fastest --
// the three x-neighbours of a voxel sit in ONE 32-bit word (two at a word boundary), so the 13 probed offsets of a
// 3x3x3 map are answered by 5 word reads of a map that stays in the L2s.  A 16-byte header behind the bitmaps says
// whether the spatial map is valid (tables built from bare keys: it is not) and holds the stride.
constexpr int kSpatialMagic = 0x53424D31;
static inline int64_t table_sbytes(int64_t cap) { return cap > 16384 ? cap : 16384; }
static inline int64_t table_total_bytes(int64_t cap) { return cap * 13 + table_sbytes(cap) + 64; }

static inline int64_t table_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static inline TableView table_view(const void* table, int64_t table_bytes) {
  // capacity: the largest power of two whose full layout fits (buffers of lidal_hash_table_bytes); a smaller buffer
  // is taken for bare slots (12 bytes each, no bitmaps)
  int64_t c = 1024;
  const bool full = table_bytes >= table_total_bytes(1024);
  if (full) {
    while (table_total_bytes(c * 2) <= table_bytes) c <<= 1;
  } else {
    c = 1;
    while (c * 2 * 12 <= table_bytes) c <<= 1;
  }
  TableView t;
  t.keys = (unsigned long long*)table;
  t.vals = (int*)((char*)table + c * 8);
  t.mask = (uint64_t)c - 1;
  t.bits = full ? (unsigned*)((char*)table + c * 12) : nullptr;
  t.sbits = full ? (unsigned*)((char*)table + c * 13) : nullptr;
  t.hdr = full ? (const int*)((char*)table + c * 13 + table_sbytes(c)) : nullptr;
  return t;
}
// x / y bits of the spatial bitmap of a table of capacity `cap` (z: 5 bits, batch: 2 bits)
static inline void table_spatial_dims(int64_t cap, int* xb, int* yb) {
  int lg = 0;
  while ((1ll << lg) < table_sbytes(cap)) ++lg;
  const int xy = lg + 3 - 7;
  *xb = (xy + 1) / 2;
  *yb = xy / 2;
}

__device__ __forceinline__ uint64_t mix_key(uint64_t key) {
  // murmur3 finaliser: the FNV low bits are well mixed already but arbitrary i64 keys are allowed
  key ^= key >> 33; key *= 0xff51afd7ed558ccdULL; key ^= key >> 33;
  return key;
}
__device__ __forceinline__ uint64_t slot_of(uint64_t key, uint64_t mask) { return mix_key(key) & mask; }
// bit of `key` in the occupancy bitmap of a table with slot mask `mask` (8 bits per slot)
__device__ __forceinline__ uint64_t bit_of(uint64_t mixed, uint64_t mask) { return (mixed >> 32) & (mask * 8 + 7); }
// bit of voxel (x, y, z, b) in the spatial bitmap; shift = log2 of the tensor stride
__device__ __forceinline__ unsigned sbit_of(int x, int y, int z, int b, int shift, int xb, int yb) {
  const unsigned xi = (unsigned)(x >> shift) & ((1u << xb) - 1u), yi = (unsigned)(y >> shift) & ((1u << yb) - 1u);
  const unsigned zi = (unsigned)(z >> shift) & 31u, bi = (unsigned)b & 3u;
  return ((((bi << 5) | zi) << yb | yi) << xb) | xi;
}

__device__ __forceinline__ int table_lookup(const TableView& t, uint64_t key) {
  uint64_t s = slot_of(key, t.mask);
  while (true) {
    unsigned long long k = t.keys[s];
    if (k == key) return t.vals[s];
    if (k == kEmptyKey) return -1;
    s = (s + 1) & t.mask;
  }
}

// ---- wave / block helpers ----
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// exclusive rank of this lane among the set lanes of a wave ballot
__device__ __forceinline__ int ballot_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
  return __uint_as_float(((unsigned)v) << 16);
}

// ---- per-device one-time kernel attributes (hipFuncSetAttribute is per device) ----
constexpr int MAX_DEVICES = 16;
static inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
  return d;
}

// ---- weight-gradient decomposition shared by conv.hip, wgrad_dma.hip and the reducer ----
// Split-K policy: offset k with nk rules is cut into ceil(nk / target_chunk) slabs (at least 1, at
// most the `splits` slabs the caller allocated), so the centre offset (every row has a rule) gets
// proportionally more workgroups than the others.
__host__ __device__ __forceinline__ int splits_for(int64_t nk, int max_splits, int target_chunk) {
  int64_t s = (nk + target_chunk - 1) / target_chunk;
  if (s < 1) s = 1;
  if (s > max_splits) s = max_splits;
  return (int)s;
}
// 16-wide MFMA blocks per wave along one channel dimension (x2 waves = the workgroup tile)
static inline int wgrad_blocks(int c) {
  if (c <= 32) return 1;
  if (c <= 64) return 2;
  if (c % 128 == 0) return 4;
  if (c % 96 == 0 || c < 128) return 3;
  return 4;
}
// wgrad_dma.hip: the bf16 weight gradient with LDS-DMA gathers (whole 16-byte channel segments,
// operands below 4 GiB); W workgroups, W + K slabs of scratch
bool wgrad_dma_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb);
int wgrad_dma_workgroups(int64_t n_a, int64_t n_b, int k, int ca, int cb);
int wgrad_dma(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* pairs,
              const int64_t* koff, int a_col, float* gw, float* partial, int W, int K, int ca,
              int cb, hipStream_t s);
// ... and its split form for f32 operands (round 6): the operands are cut into three bf16 pieces each (into `scratch`,
// wgrad_split_scratch_bytes) and multiplied as six bf16 products per f32 product; W workgroups, W + K slabs
bool wgrad_split_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb);
int wgrad_split_workgroups(int64_t n_a, int64_t n_b, int k, int ca, int cb);
int64_t wgrad_split_scratch_bytes(int64_t n_a, int64_t n_b, int ca, int cb);
int wgrad_split(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* pairs, const int64_t* koff, int a_col,
                float* gw, float* partial, void* scratch, int W, int K, int ca, int cb, hipStream_t s);

// ... and its streamed form (round 6): rule lists re-ordered into one stream per workgroup (lidal_wgrad_streams_build), 2 W slabs
bool wgrad_stream_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb);
int wgrad_stream(const void* a, const void* b, int64_t n_a, int64_t n_b, const int* spairs, const int* sdesc, int W, int a_col,
                 float* gw, float* partial, int K, int ca, int cb, hipStream_t s);

// sort.hip: stable LSD radix sort (Onesweep) of u32 / u64 keys with an optional i32 payload by key bits
// [0, end_bit): 1 + ceil(end_bit / 8) launches; keys_in / vals_in are not written; vals_in == NULL sorts
// keys only
int64_t radix_sort_ws_bytes(int64_t n, int key_bytes, bool has_val);
// n_dev (device i64, optional): the live item count min(*n_dev, n) when only the device knows it (launches sized for n)
int radix_sort(const void* keys_in, const int* vals_in, void* keys_out, int* vals_out, int64_t n,
               int key_bytes, int end_bit, void* ws, int64_t ws_bytes, hipStream_t s, const int64_t* n_dev = nullptr);
int64_t sort_pairs_ws_bytes(int64_t n);
int sort_pairs_u32(const unsigned* keys_in, const int* vals_in, unsigned* keys_out, int* vals_out,
                   int64_t n, int bits, void* ws, int64_t ws_bytes, hipStream_t s, const int64_t* n_dev = nullptr);

}  // namespace lidal
