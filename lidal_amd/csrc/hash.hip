// sphash / kernel hash / hash-table build + query for gfx950.
//
// HBM-bound integer work: every kernel reads the [n,4] int32 coordinate rows as one 16-byte
// load per lane (coalesced 1 KiB per wave instruction) and writes 8-byte hashes.
#include "common.h"

using namespace lidal;

__global__ void __launch_bounds__(256) hash_kernel(const int4* __restrict__ coords, int64_t n,
                                                   int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int4 c = coords[i];
    out[i] = fnv60(c.x, c.y, c.z, c.w);
  }
}

// out [k, n]: thread per row, loops over the k offsets (held in SGPRs via uniform loads) so the
// coordinate row is read from HBM once and each of the k output rows is written coalesced.
__global__ void __launch_bounds__(256) kernel_hash_kernel(const int4* __restrict__ coords,
                                                          int64_t n,
                                                          const int* __restrict__ offsets, int k,
                                                          int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int4 c = coords[i];
    for (int kk = 0; kk < k; ++kk) {
      int ox = offsets[kk * 3 + 0], oy = offsets[kk * 3 + 1], oz = offsets[kk * 3 + 2];
      out[(int64_t)kk * n + i] = fnv60(c.x + ox, c.y + oy, c.z + oz, c.w);
    }
  }
}

__global__ void __launch_bounds__(256) table_insert_kernel(const int64_t* __restrict__ keys,
                                                           int64_t n, TableView t) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t key = (uint64_t)keys[i];
    const uint64_t mixed = mix_key(key);
    uint64_t s = mixed & t.mask;
    if (t.bits != nullptr) {
      const uint64_t b = bit_of(mixed, t.mask);
      atomicOr(&t.bits[b >> 5], 1u << (b & 31));
    }
    while (true) {
      unsigned long long prev = atomicCAS(&t.keys[s], (unsigned long long)kEmptyKey,
                                          (unsigned long long)key);
      if (prev == kEmptyKey || prev == key) {
        atomicMin(&t.vals[s], (int)i);   // first occurrence wins, independent of arrival order
        break;
      }
      s = (s + 1) & t.mask;
    }
  }
}

__global__ void __launch_bounds__(256) table_query_kernel(TableView t,
                                                          const int64_t* __restrict__ q,
                                                          int64_t nq, int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nq; i += stride) out[i] = (int64_t)table_lookup(t, (uint64_t)q[i]);
}

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
  int64_t g = cdiv(n, block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

extern "C" int lidal_hash(const int32_t* coords, int64_t n, int64_t* out, void* stream) {
  if (n == 0) return 0;
  hash_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const int4*)coords, n, out);
  LIDAL_CHECK_LAUNCH("lidal_hash");
  return 0;
}

// out[i] = (floor(x / s) s, floor(y / s) s, floor(z / s) s, (int)b) of a float row (x, y, z, b):
// network/utils.py:44-47,72-75 (`torch.floor(z.C[:, :3] / s).int() * s` + cat of the batch column) as
// ONE pass instead of six elementwise launches per point<->voxel exchange.  IEEE division, as torch
// computes it for the power-of-two strides of the model (where a reciprocal multiply is exact too).
__global__ void __launch_bounds__(256) floor_coords_kernel(const float4* __restrict__ c, int64_t n, float s,
                                                           int si, int4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = c[i];
  out[i] = make_int4((int)floorf(v.x / s) * si, (int)floorf(v.y / s) * si, (int)floorf(v.z / s) * si, (int)v.w);
}

extern "C" int lidal_floor_coords(const float* coords, int64_t n, int stride, int32_t* out, void* stream) {
  if (n == 0) return 0;
  LIDAL_REQUIRE(stride > 0, "floor_coords: stride %d", stride);
  floor_coords_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const float4*)coords, n, (float)stride,
                                                                    stride, (int4*)out);
  LIDAL_CHECK_LAUNCH("lidal_floor_coords");
  return 0;
}

// network/utils.py:14-17: new = ((x, y, z) * init_res) / after_res, batch column kept; floored = floor(new).int().
// Separately rounded multiply and divide (torch computes two element-wise passes).
__global__ void __launch_bounds__(256) revoxelize_coords_kernel(const float4* __restrict__ c, int64_t n, float init_res,
                                                                float after_res, float4* __restrict__ out_float,
                                                                int4* __restrict__ out_floor) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = c[i];
  const float x = __fdiv_rn(__fmul_rn(v.x, init_res), after_res);
  const float y = __fdiv_rn(__fmul_rn(v.y, init_res), after_res);
  const float z = __fdiv_rn(__fmul_rn(v.z, init_res), after_res);
  out_float[i] = make_float4(x, y, z, v.w);
  out_floor[i] = make_int4((int)floorf(x), (int)floorf(y), (int)floorf(z), (int)floorf(v.w));
}

extern "C" int lidal_revoxelize_coords(const float* coords, int64_t n, float init_res, float after_res, float* out_float,
                                       int32_t* out_floor, void* stream) {
  if (n == 0) return 0;
  LIDAL_REQUIRE(after_res != 0.f, "revoxelize_coords: after_res is zero");
  revoxelize_coords_kernel<<<grid_for(n, 256, 1 << 30), 256, 0, (hipStream_t)stream>>>(
      (const float4*)coords, n, init_res, after_res, (float4*)out_float, (int4*)out_floor);
  LIDAL_CHECK_LAUNCH("lidal_revoxelize_coords");
  return 0;
}

extern "C" int lidal_kernel_hash(const int32_t* coords, int64_t n, const int32_t* offsets, int k,
                                 int64_t* out, void* stream) {
  if (n == 0 || k == 0) return 0;
  kernel_hash_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const int4*)coords, n,
                                                                    offsets, k, out);
  LIDAL_CHECK_LAUNCH("lidal_kernel_hash");
  return 0;
}

extern "C" int64_t lidal_hash_table_bytes(int64_t n_keys) {
  return table_total_bytes(table_capacity(n_keys));       // slots + hashed bitmap + spatial bitmap + header, common.h
}

extern "C" int lidal_hash_table_build(const int64_t* keys, int64_t n, void* table,
                                      int64_t table_bytes, void* stream) {
  LIDAL_REQUIRE(table_bytes >= lidal_hash_table_bytes(n), "hash table too small: %lld < %lld",
                (long long)table_bytes, (long long)lidal_hash_table_bytes(n));
  TableView t = table_view(table, table_bytes);
  int64_t cap = (int64_t)t.mask + 1;
  hipStream_t s = (hipStream_t)stream;
  LIDAL_HIP(hipMemsetAsync(t.keys, 0xFF, cap * 8, s));
  LIDAL_HIP(hipMemsetAsync(t.vals, 0x7F, cap * 4, s));   // 0x7F7F7F7F > any index
  if (t.bits != nullptr) LIDAL_HIP(hipMemsetAsync(t.bits, 0, cap, s));
  if (t.hdr != nullptr) LIDAL_HIP(hipMemsetAsync((void*)t.hdr, 0, 64, s));      // (bare keys: no spatial bitmap)
  if (n == 0) return 0;
  table_insert_kernel<<<grid_for(n), 256, 0, s>>>(keys, n, t);
  LIDAL_CHECK_LAUNCH("lidal_hash_table_build");
  return 0;
}

// the same table from the coordinates themselves -- key i = lidal_hash(coords)[i] -- plus the spatial occupancy
// bitmap (common.h) the symmetric kernel-map probes read
__global__ void __launch_bounds__(256) table_insert_coords_kernel(const int4* __restrict__ coords, int64_t n, TableView t,
                                                                  int shift, int xb, int yb, int* __restrict__ hdr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { hdr[0] = kSpatialMagic; hdr[1] = shift; hdr[2] = xb; hdr[3] = yb; }
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int4 c = coords[i];
    const uint64_t key = (uint64_t)fnv60(c.x, c.y, c.z, c.w);
    const uint64_t mixed = mix_key(key);
    uint64_t s = mixed & t.mask;
    const uint64_t b = bit_of(mixed, t.mask);
    atomicOr(&t.bits[b >> 5], 1u << (b & 31));
    const unsigned sb = sbit_of(c.x, c.y, c.z, c.w, shift, xb, yb);
    atomicOr(&t.sbits[sb >> 5], 1u << (sb & 31));
    while (true) {
      unsigned long long prev = atomicCAS(&t.keys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (prev == kEmptyKey || prev == key) {
        atomicMin(&t.vals[s], (int)i);
        break;
      }
      s = (s + 1) & t.mask;
    }
  }
}

extern "C" int lidal_hash_table_build_coords(const int32_t* coords, int64_t n, int stride, void* table,
                                             int64_t table_bytes, void* stream) {
  LIDAL_REQUIRE(table_bytes >= lidal_hash_table_bytes(n), "hash table too small: %lld < %lld",
                (long long)table_bytes, (long long)lidal_hash_table_bytes(n));
  LIDAL_REQUIRE(stride >= 1 && (stride & (stride - 1)) == 0, "hash_table_build_coords: the tensor stride must be a power of "
                "two (got %d)", stride);
  TableView t = table_view(table, table_bytes);
  const int64_t cap = (int64_t)t.mask + 1;
  hipStream_t s = (hipStream_t)stream;
  LIDAL_HIP(hipMemsetAsync(t.keys, 0xFF, cap * 8, s));
  LIDAL_HIP(hipMemsetAsync(t.vals, 0x7F, cap * 4, s));
  LIDAL_HIP(hipMemsetAsync(t.bits, 0, cap + table_sbytes(cap) + 64, s));         // both bitmaps and the header
  if (n == 0) return 0;
  int shift = 0;
  while ((1 << shift) < stride) ++shift;
  int xb, yb;
  table_spatial_dims(cap, &xb, &yb);
  table_insert_coords_kernel<<<grid_for(n), 256, 0, s>>>((const int4*)coords, n, t, shift, xb, yb, (int*)t.hdr);
  LIDAL_CHECK_LAUNCH("lidal_hash_table_build_coords");
  return 0;
}

extern "C" int lidal_hash_table_query(const void* table, int64_t table_bytes, const int64_t* q,
                                      int64_t nq, int64_t* out, void* stream) {
  if (nq == 0) return 0;
  TableView t = table_view(table, table_bytes);
  table_query_kernel<<<grid_for(nq), 256, 0, (hipStream_t)stream>>>(t, q, nq, out);
  LIDAL_CHECK_LAUNCH("lidal_hash_table_query");
  return 0;
}
