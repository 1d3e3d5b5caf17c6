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

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

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

// ---- open-addressing hash table: slots of {key u64, val i32} in two arrays ----
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFULL;

struct TableView {
  unsigned long long* keys;
  int* vals;
  uint64_t mask;
};

static inline int64_t table_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static inline TableView table_view(const void* table, int64_t table_bytes) {
  int64_t cap = table_bytes / 12;
  // capacity is the largest power of two with 12*cap <= table_bytes
  int64_t c = 1;
  while (c * 2 <= cap) c <<= 1;
  TableView t;
  t.keys = (unsigned long long*)table;
  t.vals = (int*)((char*)table + c * 8);
  t.mask = (uint64_t)c - 1;
  return t;
}

__device__ __forceinline__ uint64_t slot_of(uint64_t key, uint64_t mask) {
  // murmur3 finaliser: the FNV low bits are well mixed already but arbitrary i64 keys are allowed
  key ^= key >> 33; key *= 0xff51afd7ed558ccdULL; key ^= key >> 33;
  return key & mask;
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

}  // namespace lidal
