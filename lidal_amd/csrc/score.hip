// Probability-inference post-processing and LiDAL inter-frame divergence / entropy scoring
// for gfx950.  Replaces score/prob_inference.py:100-113 and score/sv_level/LiDAL.py:59-98.
//
// All kernels are HBM/L2-bound gathers over [points, classes] f32 rows (76-byte rows for 19
// classes); the arithmetic follows numpy/scipy's float widths and summation orders
// (f32 pairwise-8 row sums, f64 KL terms rounded to f32, f64 divergence accumulator).
#include <cstring>


#include "common.h"

// The reference computes these quantities with numpy (separately rounded products and sums); hipcc
// contracts a*b+c into fma even through __dmul_rn/__dadd_rn, so this unit is built with
// -ffp-contract=off (lidal_amd/build.py).

using namespace lidal;

namespace {

constexpr int MAXC = 32;    // classes (19 SemanticKITTI, 16 nuScenes)

// numpy's pairwise float32 add-reduce of n <= 128 contiguous values (n = number of classes)
__device__ __forceinline__ float np_sum_f32(const float* a, int n) {
  if (n < 8) {
    float r = 0.f;
    for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
    return r;
  }
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}

// ---------------- view-mean softmax ----------------
__global__ void __launch_bounds__(256) view_mean_softmax_kernel(const float* __restrict__ logits,
                                                                const int64_t* __restrict__ inverse,
                                                                int reps, int64_t p, int c,
                                                                float* __restrict__ prob,
                                                                int64_t* __restrict__ pred) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
  for (int v = 0; v < reps; ++v) {
    const float* row = logits + inverse[(int64_t)v * p + i] * c;
    float x[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < c) { x[j] = row[j]; mx = fmaxf(mx, x[j]); }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < c) { x[j] = expf(x[j] - mx); s += x[j]; }
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < c) acc[j] = (v == 0) ? (x[j] / s) : __fadd_rn(acc[j], x[j] / s);
  }
  float best = -INFINITY;
  int arg = 0;
  const float inv = (float)reps;
#pragma unroll
  for (int j = 0; j < MAXC; ++j)
    if (j < c) {
      float m = acc[j] / inv;
      prob[i * c + j] = m;
      if (m > best) { best = m; arg = j; }
    }
  pred[i] = arg;
}

// ---------------- confusion matrix of evaluate.py:100-109 + utils/iou_sk.py:14-19 ----------------
// per point: voxel row through the inverse index, argmax over the c logits (first maximum, as
// torch.max / np.argmax), and conf[pred * c + gt] += 1 for labelled points (gt < 100).  Integer
// counts: per-workgroup LDS histogram, then global integer atomics (order independent, exact).
__global__ void __launch_bounds__(256) confusion_kernel(const float* __restrict__ logits,
                                                        const int64_t* __restrict__ inverse,
                                                        const int64_t* __restrict__ labels,
                                                        int64_t p, int c, int* __restrict__ conf) {
  __shared__ int hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < c * c; i += 256) hist[i] = 0;
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < p) {
    const int64_t gt = labels[i];
    if (gt >= 0 && gt < 100 && gt < c) {
      const float* row = logits + inverse[i] * c;
      float best = row[0];
      int arg = 0;
      for (int j = 1; j < c; ++j) {
        float v = row[j];
        if (v > best) { best = v; arg = j; }
      }
      atomicAdd(&hist[arg * c + (int)gt], 1);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < c * c; j += 256)
    if (hist[j]) atomicAdd(&conf[j], hist[j]);
}

// ---------------- world-frame registration (dataset/prepare_kdtree_sk.py:76-80) ----------------
// world[p][j] = ((h0*P[j][0] + h1*P[j][1]) + h2*P[j][2]) + 1*P[j][3] with h = (f64)point: the
// reference's np.sum(expand_dims(hcoords, 2) * pose.T, axis=1), products rounded individually.
__global__ void __launch_bounds__(256) register_kernel(const float* __restrict__ pts, int64_t p,
                                                       const double* __restrict__ pose,
                                                       double* __restrict__ world) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  const double h0 = (double)pts[i * 3 + 0], h1 = (double)pts[i * 3 + 1], h2 = (double)pts[i * 3 + 2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double v = __dadd_rn(__dmul_rn(h0, pose[j * 4 + 0]), __dmul_rn(h1, pose[j * 4 + 1]));
    v = __dadd_rn(v, __dmul_rn(h2, pose[j * 4 + 2]));
    world[i * 3 + j] = __dadd_rn(v, pose[j * 4 + 3]);
  }
}

// ---------------- uniform grid for radius-limited nearest neighbour ----------------
// grid buffer: [header 64 B][table 12*cap][sorted_keys 8*p][sorted_idx 4*p]
struct GridHeader {
  int64_t p;
  int64_t cap;
  double cell;
  double inv_cell_unused;
};

struct __attribute__((aligned(16))) GridRec { unsigned long long key; double x, y, z; int idx; int pad; };

struct GridView {
  TableView t;
  const uint64_t* keys;
  const int* idx;
  const struct GridRec* rec;     // the points in cell order, one 48-byte record each (round 5: a candidate -- its cell key, its
                                 // coordinates, its id -- is ONE access instead of three arrays and a gathered point)
  int64_t p;
  double cell;
};
// byte offsets inside a grid buffer (after the 64-byte header) for p points and capacity cap:
//   keys u64 [cap] | vals i32 [cap] | sorted cell keys u64 [p] | sorted point ids i32 [p] | records GridRec [p] |
//   occupancy bitmap u32 [cap / 4] (8 bits per slot, csrc/common.h: most probed cells are empty)
__host__ __device__ inline int64_t grid_off_skeys(int64_t cap) { return cap * 12; }
__host__ __device__ inline int64_t grid_off_sidx(int64_t cap, int64_t q) { return cap * 12 + ((8 * q + 255) / 256) * 256; }
__host__ __device__ inline int64_t grid_off_spts(int64_t cap, int64_t q) { return grid_off_sidx(cap, q) + ((4 * q + 255) / 256) * 256; }
__host__ __device__ inline int64_t grid_off_bits(int64_t cap, int64_t q) { return grid_off_spts(cap, q) + ((48 * q + 255) / 256) * 256; }

constexpr int64_t kBias = 1 << 20;

__device__ __forceinline__ uint64_t cell_key(int64_t ix, int64_t iy, int64_t iz) {
  return ((uint64_t)(ix + kBias) << 42) | ((uint64_t)(iy + kBias) << 21) | (uint64_t)(iz + kBias);
}

static inline int64_t grid_cap(int64_t p) { return table_capacity(p); }

static inline GridView grid_view(const void* grid, int64_t p, int64_t cap, double cell) {
  GridView g;
  char* base = (char*)grid + 64;
  g.t.keys = (unsigned long long*)base;
  g.t.vals = (int*)(base + cap * 8);
  g.t.mask = (uint64_t)cap - 1;
  const int64_t q = p > 0 ? p : 1;
  g.t.bits = (unsigned*)(base + grid_off_bits(cap, q));
  g.t.sbits = nullptr; g.t.hdr = nullptr;
  g.keys = (const uint64_t*)(base + grid_off_skeys(cap));
  g.idx = (const int*)(base + grid_off_sidx(cap, q));
  g.rec = (const GridRec*)(base + grid_off_spts(cap, q));
  g.p = p;
  g.cell = cell;
  return g;
}

__global__ void __launch_bounds__(256) grid_keys_kernel(const double* __restrict__ pts, int64_t p,
                                                        double cell, uint64_t* __restrict__ keys,
                                                        int* __restrict__ idx, GridHeader* __restrict__ hdr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  if (i == 0) { hdr->p = p; hdr->cap = 0; hdr->cell = cell; hdr->inv_cell_unused = 0.; }     // the queries read the cell size
  int64_t ix = (int64_t)floor(pts[i * 3 + 0] / cell), iy = (int64_t)floor(pts[i * 3 + 1] / cell),
          iz = (int64_t)floor(pts[i * 3 + 2] / cell);
  keys[i] = cell_key(ix, iy, iz);
  idx[i] = (int)i;
}

__global__ void __launch_bounds__(256) grid_heads_kernel(const uint64_t* __restrict__ skeys,
                                                         int64_t p, TableView t, const double* __restrict__ pts,
                                                         const int* __restrict__ sidx, GridRec* __restrict__ rec) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  {                     // the point itself, into cell order
    const int64_t j = sidx[i];
    GridRec r;
    r.key = skeys[i]; r.x = pts[j * 3 + 0]; r.y = pts[j * 3 + 1]; r.z = pts[j * 3 + 2]; r.idx = (int)j; r.pad = 0;
    rec[i] = r;
  }
  uint64_t key = skeys[i];
  if (i > 0 && skeys[i - 1] == key) return;
  const uint64_t mixed = mix_key(key);
  uint64_t s = mixed & t.mask;
  if (t.bits != nullptr) {
    const uint64_t b = bit_of(mixed, t.mask);
    atomicOr(&t.bits[b >> 5], 1u << (b & 31));
  }
  while (true) {       // cell keys are unique among heads
    unsigned long long prev = atomicCAS(&t.keys[s], (unsigned long long)kEmptyKey,
                                        (unsigned long long)key);
    if (prev == kEmptyKey) { t.vals[s] = (int)i; break; }
    s = (s + 1) & t.mask;
  }
}

// nearest point of the grid's frame among the cells that meet the cube [q - r, q + r]^3 (27 cells when the cell is the
// radius r, at most 8 when it is 2 r: round 4 -- a look-up is a random probe of a hash table, and LiDAR cells are mostly
// empty); returns index or -1, *d2out.  The candidates are a superset of the ball's points either way and ties go to the
// lowest index, so the answer does not depend on the cell size.
__device__ __forceinline__ void grid_scan_cell(const GridView& g, uint64_t key, int start, double qx, double qy, double qz,
                                               double& best, int& arg) {
  for (int64_t s = start; s < g.p; ++s) {
    const GridRec r = g.rec[s];
    if (r.key != key) break;
    const int j = r.idx;
    double ex = r.x - qx, ey = r.y - qy, ez = r.z - qz;
    double d2 = __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez));
    if (d2 < best || (d2 == best && j < arg)) { best = d2; arg = j; }
  }
}

__device__ __forceinline__ int grid_nearest(const GridView& g, const double* __restrict__ npts,
                                            double qx, double qy, double qz, double r, double* d2out) {
  const double rr = r * (1.0 + 1e-9) + 1e-12;            // (a point at exactly r must not fall off the cube through rounding)
  const int64_t x0 = (int64_t)floor((qx - rr) / g.cell), x1 = (int64_t)floor((qx + rr) / g.cell);
  const int64_t y0 = (int64_t)floor((qy - rr) / g.cell), y1 = (int64_t)floor((qy + rr) / g.cell);
  const int64_t z0 = (int64_t)floor((qz - rr) / g.cell), z1 = (int64_t)floor((qz + rr) / g.cell);
  double best = INFINITY;
  int arg = -1;
  // (round 5, measured: probing the eight cells of a query in stages -- eight bitmap words, then the slots, then the values,
  // each stage's loads in flight together -- 287.8 us against this loop's 289.1: the kernel is bound by the ~20 random
  // 64-byte sectors a query touches beyond the L2s, 1.5 GB per launch, not by a thread's dependent chain)
  for (int64_t ix = x0; ix <= x1; ++ix)
    for (int64_t iy = y0; iy <= y1; ++iy)
      for (int64_t iz = z0; iz <= z1; ++iz) {
        uint64_t key = cell_key(ix, iy, iz);
        if (g.t.bits != nullptr) {      // an empty cell is the common answer: told by the bitmap, from the L2s
          const uint64_t b = bit_of(mix_key(key), g.t.mask);
          if (!((g.t.bits[b >> 5] >> (b & 31)) & 1u)) continue;
        }
        int start = table_lookup(g.t, key);
        if (start < 0) continue;
        grid_scan_cell(g, key, start, qx, qy, qz, best, arg);
      }
  *d2out = best;
  return arg;
}

constexpr int MAXNEI = 32;
struct NeiArgs {
  const void* grid[MAXNEI];
  const double* pts[MAXNEI];
  const float* prob[MAXNEI];
  int64_t p[MAXNEI];
  int64_t cap[MAXNEI];
  int n;
};

__device__ __forceinline__ GridView nei_grid(const NeiArgs& nei, int n, double cell) {
  GridView g;
  char* base = (char*)nei.grid[n] + 64;
  g.t.keys = (unsigned long long*)base;
  g.t.vals = (int*)(base + nei.cap[n] * 8);
  g.t.mask = (uint64_t)nei.cap[n] - 1;
  const int64_t q = nei.p[n] > 0 ? nei.p[n] : 1;
  g.t.bits = (unsigned*)(base + grid_off_bits(nei.cap[n], q));
  g.t.sbits = nullptr; g.t.hdr = nullptr;
  g.keys = (const uint64_t*)(base + grid_off_skeys(nei.cap[n]));
  g.idx = (const int*)(base + grid_off_sidx(nei.cap[n], q));
  g.rec = (const GridRec*)(base + grid_off_spts(nei.cap[n], q));
  g.p = nei.p[n];
  g.cell = cell;
  return g;
}

// Stage 1: one thread per (neighbour frame, query point): the nearest point of that frame within
// the match radius, or -1.  The 24 (or 10) look-ups of a query point are independent, so they run
// as p * n_nei threads instead of one serial chain of 27 * n_nei hash probes per point.
// q_order (round 6): the query frame's point ids in CELL order (the sorted ids of its own grid), or NULL.  Thread t then
// takes point q_order[t]: the 64 queries of a wave sit in a handful of neighbouring cells, so their bitmap words, slots and
// candidate records are the SAME few cache lines (2-3 points share a cell, neighbouring cells share half of the eight cells
// they probe) instead of ~20 random 64-byte sectors per query in scan order.  match[] is kept in thread order; stage 2
// walks the same order.  Same per-point arithmetic: same results bit for bit.
__global__ void __launch_bounds__(256)
interframe_match_kernel(const double* __restrict__ q_pts, int64_t p, NeiArgs nei, double dis_thresh,
                        int* __restrict__ match /*[n_nei][p]*/, const int* __restrict__ q_order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (t >= p) return;
  if (nei.p[n] <= 0) { match[(int64_t)n * p + t] = -1; return; }
  const int64_t i = q_order != nullptr ? (int64_t)q_order[t] : t;
  const double qx = q_pts[i * 3 + 0], qy = q_pts[i * 3 + 1], qz = q_pts[i * 3 + 2];
  const GridView g = nei_grid(nei, n, reinterpret_cast<const GridHeader*>(nei.grid[n])->cell);
  double d2;
  int j = grid_nearest(g, nei.pts[n], qx, qy, qz, dis_thresh, &d2);
  if (j >= 0 && !(sqrt(d2) <= dis_thresh)) j = -1;
  match[(int64_t)n * p + t] = j;
}

// Stage 2: one thread per query point walks its matches in the reference's neighbour order and
// accumulates with numpy's widths and summation orders.
__global__ void __launch_bounds__(256)
interframe_kernel(const float* __restrict__ q_prob, int64_t p, int c, NeiArgs nei,
                  const int* __restrict__ match, double* __restrict__ interd,
                  float* __restrict__ intere, int* __restrict__ map_count, const int* __restrict__ q_order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p) return;
  const int64_t i = q_order != nullptr ? (int64_t)q_order[t] : t;       // (match[] is in thread order)
  float q[MAXC], sum[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    q[j] = (j < c) ? q_prob[i * c + j] : 0.f;
    sum[j] = q[j];
  }
  const float eps = 0.00001f;
  double div = 0.0;
  int cnt = 0;
  for (int n = 0; n < nei.n; ++n) {
    const int j = match[(int64_t)n * p + t];
    if (j < 0) continue;
    const float* np = nei.prob[n] + (int64_t)j * c;
    float term[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < c) {
        float nv = np[k];
        sum[k] = __fadd_rn(sum[k], nv);
        double x = (double)__fadd_rn(q[k], eps), y = (double)__fadd_rn(nv, eps);
        // scipy.special.kl_div(x, y) = x log(x/y) - x + y  (x, y > 0 here), f32 result
        term[k] = (float)(x * log(x / y) - x + y);
      }
    div += (double)np_sum_f32(term, c);
    cnt += 1;
  }
  // sum_prob /= map_count (f64 divide, f32 store); entropy = scipy.stats.entropy(sum_prob)
  float pk[MAXC];
  const double mc = (double)(cnt + 1);
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
    if (k < c) pk[k] = (float)((double)sum[k] / mc);
  float tot = np_sum_f32(pk, c);
  float ent[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
    if (k < c) {
      float v = pk[k] / tot;
      ent[k] = (v > 0.f) ? (float)(-(double)v * log((double)v)) : 0.f;
    }
  intere[i] = np_sum_f32(ent, c);
  interd[i] = (cnt > 0) ? div / (double)cnt : div;
  map_count[i] = cnt;
}

// ---------------- per-supervoxel means ----------------
__global__ void __launch_bounds__(256)
supervoxel_reduce_kernel(const double* __restrict__ interd, const float* __restrict__ intere,
                         const double* __restrict__ pts, const int64_t* __restrict__ sv_ptr,
                         const int64_t* __restrict__ sv_idx, float* __restrict__ sv_interd,
                         float* __restrict__ sv_intere, float* __restrict__ sv_center) {
  __shared__ double red[5][256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = sv_ptr[s], end = sv_ptr[s + 1];
  double a[5] = {0, 0, 0, 0, 0};
  for (int64_t t = beg + tid; t < end; t += 256) {
    int64_t i = sv_idx[t];
    a[0] += interd[i];
    a[1] += (double)intere[i];
    a[2] += pts[i * 3 + 0];
    a[3] += pts[i * 3 + 1];
    a[4] += pts[i * 3 + 2];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][tid] = a[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w)
#pragma unroll
      for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    double n = (double)(end - beg);
    sv_interd[s] = (float)(red[0][0] / n);
    sv_intere[s] = (float)(red[1][0] / n);
    sv_center[s * 3 + 0] = (float)(red[2][0] / n);
    sv_center[s * 3 + 1] = (float)(red[3][0] / n);
    sv_center[s * 3 + 2] = (float)(red[4][0] / n);
  }
}

size_t sort_pairs_tmp_bytes(int64_t p) { return (size_t)radix_sort_ws_bytes(p > 0 ? p : 1, 8, true); }

}  // namespace

extern "C" int lidal_view_mean_softmax(const float* logits, const int64_t* inverse, int reps,
                                       int64_t p, int c, float* prob, int64_t* pred,
                                       void* stream) {
  LIDAL_REQUIRE(c > 0 && c <= MAXC, "view_mean_softmax: classes must be in 1..%d", MAXC);
  LIDAL_REQUIRE(reps > 0, "view_mean_softmax: reps must be positive");
  if (p == 0) return 0;
  view_mean_softmax_kernel<<<(unsigned)cdiv(p, 256), 256, 0, (hipStream_t)stream>>>(
      logits, inverse, reps, p, c, prob, pred);
  LIDAL_CHECK_LAUNCH("lidal_view_mean_softmax");
  return 0;
}

extern "C" int64_t lidal_nn_grid_bytes(int64_t p) {
  int64_t q = p > 0 ? p : 1;
  return 64 + grid_off_bits(grid_cap(q), q) + grid_cap(q);
}

extern "C" int64_t lidal_nn_grid_workspace_bytes(int64_t p) {
  int64_t q = p > 0 ? p : 1;
  return align_up(8 * q, 256) + align_up(4 * q, 256) + align_up((int64_t)sort_pairs_tmp_bytes(q), 256);
}

extern "C" int lidal_nn_grid_build(const double* pts, int64_t p, double cell, void* grid,
                                   int64_t grid_bytes, void* ws, int64_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(cell > 0, "nn_grid_build: cell must be positive");
  LIDAL_REQUIRE(grid_bytes >= lidal_nn_grid_bytes(p), "nn grid buffer too small");
  LIDAL_REQUIRE(ws_bytes >= lidal_nn_grid_workspace_bytes(p), "nn grid workspace too small");
  int64_t q = p > 0 ? p : 1;
  int64_t cap = grid_cap(q);
  char* base = (char*)grid + 64;
  LIDAL_HIP(hipMemsetAsync(base, 0xFF, cap * 8, s));
  LIDAL_HIP(hipMemsetAsync(base + cap * 8, 0xFF, cap * 4, s));
  if (p == 0) return 0;
  uint64_t* keys = (uint64_t*)ws;
  int* idx = (int*)((char*)ws + align_up(8 * q, 256));
  void* tmp = (char*)ws + align_up(8 * q, 256) + align_up(4 * q, 256);
  uint64_t* skeys = (uint64_t*)(base + grid_off_skeys(cap));
  int* sidx = (int*)(base + grid_off_sidx(cap, q));
  LIDAL_HIP(hipMemsetAsync(base + grid_off_bits(cap, q), 0, cap, s));
  grid_keys_kernel<<<(unsigned)cdiv(p, 256), 256, 0, s>>>(pts, p, cell, keys, idx, (GridHeader*)grid);
  LIDAL_CHECK_LAUNCH("grid_keys");
  if (int rc = radix_sort(keys, idx, skeys, sidx, p, 8, 63, tmp, (int64_t)sort_pairs_tmp_bytes(q), s)) return rc;
  TableView t;
  t.keys = (unsigned long long*)base;
  t.vals = (int*)(base + cap * 8);
  t.mask = (uint64_t)cap - 1;
  t.bits = (unsigned*)(base + grid_off_bits(cap, q));
  t.sbits = nullptr; t.hdr = nullptr;
  grid_heads_kernel<<<(unsigned)cdiv(p, 256), 256, 0, s>>>(skeys, p, t, pts, sidx, (GridRec*)(base + grid_off_spts(cap, q)));
  LIDAL_CHECK_LAUNCH("grid_heads");
  return 0;
}

extern "C" int64_t lidal_interframe_workspace_bytes(int64_t p, int n_nei) {
  return (int64_t)(n_nei > 0 ? n_nei : 1) * (p > 0 ? p : 1) * 4 + 256;
}

static int interframe_score(const double* q_pts, const float* q_prob, int64_t p, int c,
                            const void* const* nei_grids_host, const double* const* nei_pts_host,
                            const float* const* nei_prob_host, const int64_t* nei_p_host,
                            int n_nei, double dis_thresh, double* interd, float* intere,
                            int32_t* map_count, void* ws, int64_t ws_bytes, const void* q_grid,
                            void* stream) {
  LIDAL_REQUIRE(c > 0 && c <= MAXC, "interframe_score: classes must be in 1..%d", MAXC);
  LIDAL_REQUIRE(n_nei >= 0 && n_nei <= MAXNEI, "interframe_score: at most %d neighbours", MAXNEI);
  if (p == 0) return 0;
  LIDAL_REQUIRE(ws_bytes >= lidal_interframe_workspace_bytes(p, n_nei), "interframe workspace too small");
  int* match = (int*)ws;
  NeiArgs a;
  memset(&a, 0, sizeof(a));
  a.n = n_nei;
  for (int n = 0; n < n_nei; ++n) {
    a.grid[n] = nei_grids_host[n];
    a.pts[n] = nei_pts_host[n];
    a.prob[n] = nei_prob_host[n];
    a.p[n] = nei_p_host[n];
    a.cap[n] = grid_cap(nei_p_host[n] > 0 ? nei_p_host[n] : 1);
  }
  hipStream_t s = (hipStream_t)stream;
  // the query frame's points in cell order: the sorted ids inside its own grid buffer (lidal_nn_grid_build of q_pts)
  const int* q_order = nullptr;
  if (q_grid != nullptr) {
    const int64_t cap = grid_cap(p);
    q_order = (const int*)((const char*)q_grid + 64 + grid_off_sidx(cap, p));
  }
  if (n_nei > 0) {
    interframe_match_kernel<<<dim3((unsigned)cdiv(p, 256), (unsigned)n_nei), 256, 0, s>>>(
        q_pts, p, a, dis_thresh, match, q_order);
    LIDAL_CHECK_LAUNCH("interframe_match");
  }
  interframe_kernel<<<(unsigned)cdiv(p, 256), 256, 0, s>>>(q_prob, p, c, a, match, interd, intere,
                                                           map_count, q_order);
  LIDAL_CHECK_LAUNCH("lidal_interframe_score");
  return 0;
}

extern "C" int lidal_interframe_score(const double* q_pts, const float* q_prob, int64_t p, int c,
                                      const void* const* nei_grids_host,
                                      const double* const* nei_pts_host,
                                      const float* const* nei_prob_host, const int64_t* nei_p_host,
                                      int n_nei, double dis_thresh, double* interd, float* intere,
                                      int32_t* map_count, void* ws, int64_t ws_bytes,
                                      void* stream) {
  return interframe_score(q_pts, q_prob, p, c, nei_grids_host, nei_pts_host, nei_prob_host, nei_p_host, n_nei, dis_thresh,
                          interd, intere, map_count, ws, ws_bytes, nullptr, stream);
}

extern "C" int lidal_interframe_score_ordered(const double* q_pts, const float* q_prob, int64_t p, int c,
                                              const void* const* nei_grids_host,
                                              const double* const* nei_pts_host,
                                              const float* const* nei_prob_host, const int64_t* nei_p_host,
                                              int n_nei, double dis_thresh, double* interd, float* intere,
                                              int32_t* map_count, void* ws, int64_t ws_bytes,
                                              const void* q_grid, void* stream) {
  return interframe_score(q_pts, q_prob, p, c, nei_grids_host, nei_pts_host, nei_prob_host, nei_p_host, n_nei, dis_thresh,
                          interd, intere, map_count, ws, ws_bytes, q_grid, stream);
}

extern "C" int lidal_supervoxel_reduce(const double* interd, const float* intere,
                                       const double* pts, const int64_t* sv_ptr,
                                       const int64_t* sv_idx, int s, float* sv_interd,
                                       float* sv_intere, float* sv_center, void* stream) {
  if (s == 0) return 0;
  supervoxel_reduce_kernel<<<(unsigned)s, 256, 0, (hipStream_t)stream>>>(
      interd, intere, pts, sv_ptr, sv_idx, sv_interd, sv_intere, sv_center);
  LIDAL_CHECK_LAUNCH("lidal_supervoxel_reduce");
  return 0;
}

extern "C" int lidal_register_points(const float* points, int64_t p, const double* pose_dev,
                                     double* world, void* stream) {
  if (p == 0) return 0;
  register_kernel<<<(unsigned)cdiv(p, 256), 256, 0, (hipStream_t)stream>>>(points, p, pose_dev, world);
  LIDAL_CHECK_LAUNCH("lidal_register_points");
  return 0;
}

extern "C" int lidal_confusion_accumulate(const float* logits, const int64_t* inverse,
                                          const int64_t* labels, int64_t p, int c, int32_t* conf,
                                          void* stream) {
  LIDAL_REQUIRE(c > 0 && c <= MAXC, "confusion: classes must be in 1..%d", MAXC);
  if (p == 0) return 0;
  confusion_kernel<<<(unsigned)cdiv(p, 256), 256, 0, (hipStream_t)stream>>>(logits, inverse, labels,
                                                                            p, c, conf);
  LIDAL_CHECK_LAUNCH("lidal_confusion_accumulate");
  return 0;
}
