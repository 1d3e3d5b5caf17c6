// Launch plans: a whole forward (or backward) pass of the sparse U-Net as ONE call across the C-ABI.
//
// The reference queues its kernels one Python call at a time (train.py:127-140 -> torchsparse's autograd
// Functions -> torchsparse.backend); so did rounds 1-3 of this library, and on one ~120 k-point scan the
// step was bound by the host: ~800 launches issued from one Python thread, each behind a ctypes call, a
// few tensor allocations and an autograd node (11 ms of wall time for ~6 ms of GPU work).  With the
// coordinate tables of a step built ahead (network/geometry.py) every size of the step is known before its
// first feature kernel, so the host can lay out every buffer and every argument up front: the step becomes
// a flat stream of 64-bit words
//        kind | flags << 16,  arg 0, arg 1, ...        (argument count fixed per kind)
// -- pointers as addresses, integers as such, floats as the bits of a double -- and lidal_plan_run walks
// it, calling the SAME entry points the per-operator path calls, with the same arguments, in the same
// order: results are bitwise those of the per-operator path (tests/test_plan_gpu.py).  No graph capture:
// sizes change every step (the reference draws a new augmentation per iteration).
//
// Also here: the few row-wise helpers a planned step needs in place of torch glue (strided 2-D copy =
// channel concatenation / padding / slicing, strided sum of two gradients, f32 transposition, f32 -> bf16
// row cast with channel padding).
#include <string.h>
#include <mutex>

#include "common.h"

using namespace lidal;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// ---- dst[r][0 : row_units) = src[r][...]; dst[r][row_units : row_units + zero_units) = 0 ------------
template <typename U>
__global__ void __launch_bounds__(256) copy2d_kernel(const U* __restrict__ src, int64_t src_pitch,
                                                     U* __restrict__ dst, int64_t dst_pitch, int64_t rows,
                                                     int row_units, int zero_units) {
  const int per = row_units + zero_units;
  const int64_t total = rows * per;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / per;
    const int u = (int)(i - r * per);
    U v;
    if (u < row_units) v = src[r * src_pitch + u];
    else memset(&v, 0, sizeof(U));
    dst[r * dst_pitch + u] = v;
  }
}

template <typename T> struct EW;
template <> struct EW<float> {
  static constexpr int VEC = 4;
  typedef float4 vec;
  __device__ static void unpack(const vec& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  __device__ static vec pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct EW<__bf16> {
  static constexpr int VEC = 8;
  typedef bf16x8_t vec;
  __device__ static void unpack(const vec& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  __device__ static vec pack(const float (&f)[8]) {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    return v;
  }
};

// out[r][:] = a[r][:] + b[r][:]  (f32 sum, rounded once to T: what autograd's gradient accumulation computes);
// pitches and cv in VEC-wide chunks
template <typename T>
__global__ void __launch_bounds__(256) add2d_kernel(const T* __restrict__ a, int64_t a_pitch,
                                                    const T* __restrict__ b, int64_t b_pitch,
                                                    T* __restrict__ out, int64_t out_pitch, int64_t rows, int cv) {
  constexpr int VEC = EW<T>::VEC;
  typedef typename EW<T>::vec vec;
  const int64_t total = rows * cv;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cv;
    const int u = (int)(i - r * cv);
    float fa[VEC], fb[VEC];
    EW<T>::unpack(reinterpret_cast<const vec*>(a)[r * a_pitch + u], fa);
    EW<T>::unpack(reinterpret_cast<const vec*>(b)[r * b_pitch + u], fb);
#pragma unroll
    for (int e = 0; e < VEC; ++e) fa[e] = fa[e] + fb[e];
    reinterpret_cast<vec*>(out)[r * out_pitch + u] = EW<T>::pack(fa);
  }
}

// dst[c][r] = src[r][c], src rows `src_pitch` floats apart (small matrices: weight gradients)
__global__ void __launch_bounds__(256) transpose_f32_kernel(const float* __restrict__ src, int64_t src_pitch,
                                                            float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = src[(int64_t)(r0 + j) * src_pitch + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < cols && r0 + tx < rows) dst[(int64_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// dst bf16 [n][c_dst] = (bf16) src f32 [n][c_src], channels c_src.. zero  (round to nearest even, as torch's .to)
__global__ void __launch_bounds__(256) cast_rows_kernel(const float* __restrict__ src, int c_src,
                                                        __bf16* __restrict__ dst, int c_dst, int64_t n) {
  const int64_t total = n * c_dst;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / c_dst;
    const int c = (int)(i - r * c_dst);
    dst[i] = c < c_src ? (__bf16)src[r * c_src + c] : (__bf16)0.f;
  }
}

inline unsigned grid_for(int64_t items) {
  int64_t g = cdiv(items, 256);
  return (unsigned)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" int lidal_copy2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t rows,
                            int64_t row_bytes, int64_t zero_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(rows >= 0 && row_bytes >= 0 && zero_bytes >= 0 && row_bytes + zero_bytes <= dst_pitch
                && row_bytes <= src_pitch, "copy2d: rows of %lld + %lld bytes do not fit pitches %lld / %lld",
                (long long)row_bytes, (long long)zero_bytes, (long long)src_pitch, (long long)dst_pitch);
  if (rows == 0 || row_bytes + zero_bytes == 0) return 0;
  const uint64_t all = (uint64_t)(uintptr_t)src | (uint64_t)(uintptr_t)dst | (uint64_t)src_pitch | (uint64_t)dst_pitch |
                       (uint64_t)row_bytes | (uint64_t)zero_bytes;
  LIDAL_REQUIRE(row_bytes + zero_bytes < (1ll << 30), "copy2d: rows too long");
#define LIDAL_COPY2D(U)                                                                                        \
  copy2d_kernel<U><<<grid_for(rows * ((row_bytes + zero_bytes) / (int64_t)sizeof(U))), 256, 0, s>>>(          \
      (const U*)src, src_pitch / (int64_t)sizeof(U), (U*)dst, dst_pitch / (int64_t)sizeof(U), rows,           \
      (int)(row_bytes / (int64_t)sizeof(U)), (int)(zero_bytes / (int64_t)sizeof(U)))
  if (all % 16 == 0) LIDAL_COPY2D(uint4);
  else if (all % 4 == 0) LIDAL_COPY2D(uint32_t);
  else if (all % 2 == 0) LIDAL_COPY2D(uint16_t);
  else LIDAL_COPY2D(uint8_t);
#undef LIDAL_COPY2D
  LIDAL_CHECK_LAUNCH("lidal_copy2d");
  return 0;
}

extern "C" int lidal_add2d(const void* a, int64_t a_stride, const void* b, int64_t b_stride, void* out,
                           int64_t out_stride, int64_t rows, int c, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == LIDAL_BF16 ? 8 : 4;
  LIDAL_REQUIRE(dtype == LIDAL_F32 || dtype == LIDAL_BF16, "add2d: bad dtype %d", dtype);
  LIDAL_REQUIRE(c > 0 && c % vec == 0 && a_stride % vec == 0 && b_stride % vec == 0 && out_stride % vec == 0
                && a_stride >= c && b_stride >= c && out_stride >= c,
                "add2d: channels and row strides must be whole 16-byte vectors (c %d, strides %lld %lld %lld)", c,
                (long long)a_stride, (long long)b_stride, (long long)out_stride);
  LIDAL_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "add2d: operands must be 16-byte aligned");
  if (rows == 0) return 0;
  const int cv = c / vec;
  if (dtype == LIDAL_F32)
    add2d_kernel<float><<<grid_for(rows * cv), 256, 0, s>>>((const float*)a, a_stride / vec, (const float*)b,
                                                            b_stride / vec, (float*)out, out_stride / vec, rows, cv);
  else
    add2d_kernel<__bf16><<<grid_for(rows * cv), 256, 0, s>>>((const __bf16*)a, a_stride / vec, (const __bf16*)b,
                                                             b_stride / vec, (__bf16*)out, out_stride / vec, rows, cv);
  LIDAL_CHECK_LAUNCH("lidal_add2d");
  return 0;
}

extern "C" int lidal_transpose_f32(const float* src, int64_t src_stride, float* dst, int rows, int cols,
                                   void* stream) {
  LIDAL_REQUIRE(rows > 0 && cols > 0 && src_stride >= cols, "transpose_f32: bad shape %d x %d (stride %lld)", rows,
                cols, (long long)src_stride);
  dim3 grid((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
  transpose_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, src_stride, dst, rows, cols);
  LIDAL_CHECK_LAUNCH("lidal_transpose_f32");
  return 0;
}

extern "C" int lidal_cast_rows_bf16(const float* src, int c_src, void* dst, int c_dst, int64_t n, void* stream) {
  LIDAL_REQUIRE(c_src > 0 && c_dst >= c_src, "cast_rows: bad channel counts %d -> %d", c_src, c_dst);
  if (n == 0) return 0;
  cast_rows_kernel<<<grid_for(n * c_dst), 256, 0, (hipStream_t)stream>>>(src, c_src, (__bf16*)dst, c_dst, n);
  LIDAL_CHECK_LAUNCH("lidal_cast_rows_bf16");
  return 0;
}

// Test / debug aid: a synchronous copy of device memory to the host by ADDRESS -- a launch plan names its buffers by
// address only (they are carved out of large blocks, not tensors), so a test that replays the plan's operators
// against the oracle reads their operands back with this.
extern "C" int lidal_debug_read(const void* dev, void* host, int64_t nbytes) {
  LIDAL_REQUIRE(dev != nullptr && host != nullptr && nbytes >= 0, "debug_read: bad arguments");
  LIDAL_HIP(hipDeviceSynchronize());
  LIDAL_HIP(hipMemcpy(host, dev, (size_t)nbytes, hipMemcpyDeviceToHost));
  return 0;
}

// ---- the runner ------------------------------------------------------------------------------------
namespace {

struct OpInfo { int n_args; const char* name; };

const OpInfo kOps[] = {
    /* 0 */ {-1, "(none)"},
    /* LIDAL_OP_CONV_WEIGHT_IMAGE_BATCH 1 */ {5, "conv_weight_image_batch"},
    /* LIDAL_OP_CONV_APPLY_IMAGE 2 */ {18, "conv_apply_image"},
    /* LIDAL_OP_CONV_DGRAD_BN_SUMS 3 */ {20, "conv_dgrad_bn_sums"},
    /* LIDAL_OP_CONV_WGRAD 4 */ {14, "conv_wgrad"},
    /* LIDAL_OP_BN_TRAIN_FWD 5 */ {18, "bn_train_fwd"},
    /* LIDAL_OP_BN_TRAIN_FWD_TILES 6 */ {18, "bn_train_fwd_tiles"},
    /* LIDAL_OP_BN_BWD 7 */ {16, "bn_bwd"},
    /* LIDAL_OP_BN_BWD_TILES 8 */ {16, "bn_bwd_tiles"},
    /* LIDAL_OP_BN_EVAL_FWD 9 */ {11, "bn_eval_fwd"},
    /* LIDAL_OP_BN_FOLD 10 */ {8, "bn_fold"},
    /* LIDAL_OP_COLSUM 11 */ {7, "colsum"},
    /* LIDAL_OP_ADD_RELU_FWD 12 */ {5, "add_relu_fwd"},
    /* LIDAL_OP_ADD_RELU_BWD 13 */ {5, "add_relu_bwd"},
    /* LIDAL_OP_VOXELIZE_FWD_1TO1 14 */ {6, "voxelize_fwd_1to1"},
    /* LIDAL_OP_VOXELIZE_FWD_SORTED 15 */ {11, "voxelize_fwd_sorted"},
    /* LIDAL_OP_VOXELIZE_BWD 16 */ {9, "voxelize_bwd"},
    /* LIDAL_OP_DEVOXELIZE_FWD 17 */ {8, "devoxelize_fwd"},
    /* LIDAL_OP_DEVOXELIZE_BWD_SORTED 18 */ {11, "devoxelize_bwd_sorted"},
    /* LIDAL_OP_CE_FWD 19 */ {9, "ce_fwd"},
    /* LIDAL_OP_CE_BWD 20 */ {9, "ce_bwd"},
    /* LIDAL_OP_COPY2D 21 */ {7, "copy2d"},
    /* LIDAL_OP_ADD2D 22 */ {9, "add2d"},
    /* LIDAL_OP_TRANSPOSE_F32 23 */ {5, "transpose_f32"},
    /* LIDAL_OP_CAST_ROWS_BF16 24 */ {5, "cast_rows_bf16"},
    /* LIDAL_OP_VIEW_MEAN_SOFTMAX 25 */ {7, "view_mean_softmax"},
    /* LIDAL_OP_FORK_SIDE 26 */ {0, "fork_side"},
    /* LIDAL_OP_JOIN_SIDE 27 */ {0, "join_side"},
    /* LIDAL_OP_CONV_APPLY_IMAGE_WS 28 */ {20, "conv_apply_image_ws"},
    /* LIDAL_OP_CONV_DGRAD_BN_SUMS_WS 29 */ {22, "conv_dgrad_bn_sums_ws"},
    /* LIDAL_OP_ADD_RELU_BWD_BN_SUMS 30 */ {15, "add_relu_bwd_bn_sums"},
    /* LIDAL_OP_BN_BWD_FROM_SUMS 31 */ {16, "bn_bwd_from_sums"},
    /* LIDAL_OP_ADD_RELU_BWD_BN_TILE_SUMS 32 */ {15, "add_relu_bwd_bn_tile_sums"},
    /* LIDAL_OP_DEVOXELIZE_BWD_CELLS 33 */ {12, "devoxelize_bwd_cells"},
    /* LIDAL_OP_CONV_WGRAD_STREAMS 34 */ {15, "conv_wgrad_streams"},
};
constexpr int kNumOps = (int)(sizeof(kOps) / sizeof(kOps[0]));

inline double as_double(int64_t w) {
  double d;
  memcpy(&d, &w, sizeof(d));
  return d;
}

// one pair of events per device and side stream for fork / join (created on first use, never timed).  An edge is
// "record on A, then make B wait for that record": the wait captures the record it follows, so the same event serves
// every later edge -- but two host threads running plans on one device must not interleave their record / wait
// pairs, hence one mutex per device around each pair (and around the creation).
constexpr int MAX_SIDE = 8;
hipEvent_t g_fork_ev[MAX_DEVICES][MAX_SIDE], g_join_ev[MAX_DEVICES][MAX_SIDE];
bool g_ev_ready[MAX_DEVICES][MAX_SIDE];
std::mutex g_ev_mutex[MAX_DEVICES];

// `to` waits for everything queued on `from` so far (join = false: the fork event of side stream idx, true: its join event)
int stream_edge(int idx, bool join, hipStream_t from, hipStream_t to) {
  const int d = current_device();
  std::lock_guard<std::mutex> lock(g_ev_mutex[d]);
  if (!g_ev_ready[d][idx]) {
    LIDAL_HIP(hipEventCreateWithFlags(&g_fork_ev[d][idx], hipEventDisableTiming));
    LIDAL_HIP(hipEventCreateWithFlags(&g_join_ev[d][idx], hipEventDisableTiming));
    g_ev_ready[d][idx] = true;
  }
  hipEvent_t ev = join ? g_join_ev[d][idx] : g_fork_ev[d][idx];
  LIDAL_HIP(hipEventRecord(ev, from));
  LIDAL_HIP(hipStreamWaitEvent(to, ev, 0));
  return 0;
}

}  // namespace

extern "C" int lidal_plan_op_args(int kind) { return kind > 0 && kind < kNumOps ? kOps[kind].n_args : -1; }

extern "C" int lidal_plan_run(const int64_t* words, int64_t n_words, int64_t n_ops, void* stream, void* side_stream) {
  void* streams[2] = {stream, side_stream};
  return lidal_plan_run_streams(words, n_words, n_ops, streams, side_stream != nullptr ? 2 : 1);
}

namespace {
int run_ops(const int64_t* words, int64_t n_words, int64_t n_ops, void* const* streams, int n_streams, bool* side_open);
}

extern "C" int lidal_plan_run_streams(const int64_t* words, int64_t n_words, int64_t n_ops, void* const* streams,
                                      int n_streams) {
  LIDAL_REQUIRE(streams != nullptr && n_streams >= 1 && n_streams <= MAX_SIDE, "plan_run: 1..%d streams", MAX_SIDE);
  if (int rc = lidal_bn_check_device()) return rc;         // (bn.hip: a fused BatchNorm launch timed out earlier)
  bool side_open[MAX_SIDE] = {};
  const int rc = run_ops(words, n_words, n_ops, streams, n_streams, side_open);
  if (rc != 0) {
    // a plan that fails half way must not leave a side stream forked: the caller unwinds and hands the plan's arena
    // and scratch back to its allocator, whose re-use is ordered by the MAIN stream only -- so the main stream waits
    // for whatever the open side streams were given before the error (the error message stays the first one's)
    char msg[400];
    snprintf(msg, sizeof(msg), "%s", lidal_last_error());
    for (int i = 1; i < n_streams; ++i)
      if (side_open[i] && streams[i] != nullptr) stream_edge(i, true, (hipStream_t)streams[i], (hipStream_t)streams[0]);
    set_error("%s", msg);
  }
  return rc;
}

namespace {
int run_ops(const int64_t* words, int64_t n_words, int64_t n_ops, void* const* streams, int n_streams, bool* side_open) {
  void* const stream = streams[0];
#define P(i) ((void*)(uintptr_t)a[i])
#define CP(T, i) ((const T*)(uintptr_t)a[i])
#define MP(T, i) ((T*)(uintptr_t)a[i])
#define I(i) ((int)a[i])
#define L(i) ((int64_t)a[i])
#define F(i) ((float)as_double(a[i]))
  int64_t pos = 0;
  for (int64_t op = 0; op < n_ops; ++op) {
    LIDAL_REQUIRE(pos < n_words, "plan_run: op %lld starts past the end of the stream (%lld words)", (long long)op,
                  (long long)n_words);
    const int kind = (int)(words[pos] & 0xFFFF);
    const int flags = (int)((words[pos] >> 16) & 0xFFFF);
    LIDAL_REQUIRE(kind > 0 && kind < kNumOps, "plan_run: op %lld has unknown kind %d", (long long)op, kind);
    const int na = kOps[kind].n_args;
    LIDAL_REQUIRE(pos + 1 + na <= n_words, "plan_run: op %lld (%s) is truncated", (long long)op, kOps[kind].name);
    const int64_t* a = words + pos + 1;
    pos += 1 + na;
    void* st = stream;
    const int sidx = flags & 0xF;               // 0 = the main stream, i = side stream i
    if (sidx != 0 && kind != LIDAL_OP_FORK_SIDE && kind != LIDAL_OP_JOIN_SIDE) {
      LIDAL_REQUIRE(sidx < n_streams && streams[sidx] != nullptr && side_open[sidx], "plan_run: op %lld (%s) wants side "
                    "stream %d outside a fork / join bracket", (long long)op, kOps[kind].name, sidx);
      st = streams[sidx];
    }
    int rc = 0;
    switch (kind) {
      case LIDAL_OP_CONV_WEIGHT_IMAGE_BATCH:
        rc = lidal_conv_weight_image_batch(P(0), I(1), L(2), I(3), I(4), st);
        break;
      case LIDAL_OP_CONV_APPLY_IMAGE:
        rc = lidal_conv_apply_image(P(0), P(1), CP(int32_t, 2), CP(int32_t, 3), CP(uint32_t, 4), P(5), L(6), L(7), I(8),
                                    I(9), I(10), I(11), I(12), CP(float, 13), CP(float, 14), I(15), P(16),
                                    MP(float, 17), st);
        break;
      case LIDAL_OP_CONV_DGRAD_BN_SUMS:
        rc = lidal_conv_dgrad_bn_sums(P(0), P(1), CP(int32_t, 2), CP(int32_t, 3), CP(uint32_t, 4), P(5), L(6), L(7),
                                      I(8), I(9), I(10), I(11), I(12), P(13), CP(float, 14), CP(float, 15),
                                      CP(float, 16), CP(float, 17), I(18), MP(float, 19), st);
        break;
      case LIDAL_OP_CONV_APPLY_IMAGE_WS:
        rc = lidal_conv_apply_image_ws(P(0), P(1), CP(int32_t, 2), CP(int32_t, 3), CP(uint32_t, 4), P(5), L(6), L(7),
                                       I(8), I(9), I(10), I(11), I(12), CP(float, 13), CP(float, 14), I(15), P(16),
                                       MP(float, 17), P(18), L(19), st);
        break;
      case LIDAL_OP_CONV_DGRAD_BN_SUMS_WS:
        rc = lidal_conv_dgrad_bn_sums_ws(P(0), P(1), CP(int32_t, 2), CP(int32_t, 3), CP(uint32_t, 4), P(5), L(6), L(7),
                                         I(8), I(9), I(10), I(11), I(12), P(13), CP(float, 14), CP(float, 15),
                                         CP(float, 16), CP(float, 17), I(18), MP(float, 19), P(20), L(21), st);
        break;
      case LIDAL_OP_CONV_WGRAD:
        rc = lidal_conv_wgrad(P(0), P(1), L(2), L(3), CP(int32_t, 4), CP(int64_t, 5), I(6), MP(float, 7), MP(float, 8),
                              L(9), I(10), I(11), I(12), I(13), st);
        break;
      case LIDAL_OP_CONV_WGRAD_STREAMS:
        rc = lidal_conv_wgrad_streams(P(0), P(1), L(2), L(3), CP(int32_t, 4), CP(int32_t, 5), I(6), I(7), MP(float, 8),
                                      MP(float, 9), L(10), I(11), I(12), I(13), I(14), st);
        break;
      case LIDAL_OP_BN_TRAIN_FWD:
        rc = lidal_bn_train_fwd(P(0), I(1), L(2), I(3), CP(float, 4), CP(float, 5), F(6), F(7), MP(float, 8),
                                MP(float, 9), MP(int64_t, 10), I(11), P(12), P(13), MP(float, 14), MP(float, 15), P(16),
                                L(17), st);
        break;
      case LIDAL_OP_BN_TRAIN_FWD_TILES:
        rc = lidal_bn_train_fwd_tiles(P(0), I(1), L(2), I(3), CP(float, 4), CP(float, 5), F(6), F(7), MP(float, 8),
                                      MP(float, 9), MP(int64_t, 10), I(11), P(12), P(13), MP(float, 14), MP(float, 15),
                                      CP(float, 16), L(17), st);
        break;
      case LIDAL_OP_BN_BWD:
        rc = lidal_bn_bwd(P(0), P(1), L(2), I(3), L(4), I(5), CP(float, 6), CP(float, 7), I(8), CP(float, 9),
                          CP(float, 10), P(11), MP(float, 12), MP(float, 13), P(14), L(15), st);
        break;
      case LIDAL_OP_BN_BWD_FROM_SUMS:
        rc = lidal_bn_bwd_from_sums(P(0), P(1), L(2), I(3), L(4), I(5), CP(float, 6), CP(float, 7), I(8), CP(float, 9),
                                    CP(float, 10), P(11), MP(float, 12), MP(float, 13), P(14), L(15), st);
        break;
      case LIDAL_OP_ADD_RELU_BWD_BN_SUMS:
        rc = lidal_add_relu_bwd_bn_sums(P(0), P(1), P(2), I(3), L(4), I(5), P(6), CP(float, 7), CP(float, 8), P(9), P(10),
                                        CP(float, 11), CP(float, 12), P(13), L(14), st);
        break;
      case LIDAL_OP_ADD_RELU_BWD_BN_TILE_SUMS:
        rc = lidal_add_relu_bwd_bn_tile_sums(P(0), P(1), P(2), I(3), L(4), I(5), P(6), CP(float, 7), CP(float, 8),
                                             MP(float, 9), P(10), CP(float, 11), CP(float, 12), MP(float, 13), L(14), st);
        break;
      case LIDAL_OP_BN_BWD_TILES:
        rc = lidal_bn_bwd_tiles(P(0), P(1), L(2), I(3), L(4), I(5), CP(float, 6), CP(float, 7), I(8), CP(float, 9),
                                CP(float, 10), P(11), MP(float, 12), MP(float, 13), CP(float, 14), L(15), st);
        break;
      case LIDAL_OP_BN_EVAL_FWD:
        rc = lidal_bn_eval_fwd(P(0), I(1), L(2), I(3), CP(float, 4), CP(float, 5), CP(float, 6), CP(float, 7), F(8),
                               I(9), P(10), st);
        break;
      case LIDAL_OP_BN_FOLD:
        rc = lidal_bn_fold(CP(float, 0), CP(float, 1), CP(float, 2), CP(float, 3), F(4), I(5), MP(float, 6),
                           MP(float, 7), st);
        break;
      case LIDAL_OP_COLSUM:
        rc = lidal_colsum(P(0), I(1), L(2), I(3), MP(float, 4), P(5), L(6), st);
        break;
      case LIDAL_OP_ADD_RELU_FWD:
        rc = lidal_add_relu_fwd(P(0), P(1), P(2), L(3), I(4), st);
        break;
      case LIDAL_OP_ADD_RELU_BWD:
        rc = lidal_add_relu_bwd(P(0), P(1), P(2), L(3), I(4), st);
        break;
      case LIDAL_OP_VOXELIZE_FWD_1TO1:
        rc = lidal_voxelize_fwd_1to1(P(0), CP(int32_t, 1), P(2), L(3), I(4), I(5), st);
        break;
      case LIDAL_OP_VOXELIZE_FWD_SORTED:
        rc = lidal_voxelize_fwd_sorted(P(0), CP(int32_t, 1), CP(int64_t, 2), CP(int32_t, 3), P(4), L(5), I(6), I(7),
                                       L(8), P(9), L(10), st);
        break;
      case LIDAL_OP_VOXELIZE_BWD:
        rc = lidal_voxelize_bwd(P(0), CP(int32_t, 1), CP(int32_t, 2), P(3), P(4), L(5), L(6), I(7), I(8), st);
        break;
      case LIDAL_OP_DEVOXELIZE_FWD:
        rc = lidal_devoxelize_fwd(P(0), CP(int32_t, 1), CP(float, 2), P(3), L(4), L(5), I(6), I(7), st);
        break;
      case LIDAL_OP_DEVOXELIZE_BWD_SORTED:
        rc = lidal_devoxelize_bwd_sorted(P(0), CP(int32_t, 1), CP(int64_t, 2), CP(float, 3), P(4), L(5), I(6), I(7),
                                         L(8), P(9), L(10), st);
        break;
      case LIDAL_OP_DEVOXELIZE_BWD_CELLS:
        rc = lidal_devoxelize_bwd_cells(P(0), CP(int32_t, 1), CP(int64_t, 2), CP(float, 3), CP(int32_t, 4), CP(int64_t, 5),
                                        P(6), L(7), I(8), I(9), P(10), L(11), st);
        break;
      case LIDAL_OP_CE_FWD:
        rc = lidal_ce_fwd(P(0), I(1), CP(int64_t, 2), L(3), I(4), L(5), MP(float, 6), P(7), L(8), st);
        break;
      case LIDAL_OP_CE_BWD:
        rc = lidal_ce_bwd(P(0), I(1), CP(int64_t, 2), L(3), I(4), L(5), CP(float, 6), CP(float, 7), P(8), st);
        break;
      case LIDAL_OP_COPY2D:
        rc = lidal_copy2d(P(0), L(1), P(2), L(3), L(4), L(5), L(6), st);
        break;
      case LIDAL_OP_ADD2D:
        rc = lidal_add2d(P(0), L(1), P(2), L(3), P(4), L(5), L(6), I(7), I(8), st);
        break;
      case LIDAL_OP_TRANSPOSE_F32:
        rc = lidal_transpose_f32(CP(float, 0), L(1), MP(float, 2), I(3), I(4), st);
        break;
      case LIDAL_OP_CAST_ROWS_BF16:
        rc = lidal_cast_rows_bf16(CP(float, 0), I(1), P(2), I(3), L(4), st);
        break;
      case LIDAL_OP_VIEW_MEAN_SOFTMAX:
        rc = lidal_view_mean_softmax(CP(float, 0), CP(int64_t, 1), I(2), L(3), I(4), MP(float, 5), MP(int64_t, 6), st);
        break;
      case LIDAL_OP_FORK_SIDE: {         // side stream i (the flags; 0 means 1) waits for everything queued on the main stream so far
        const int i = sidx ? sidx : 1;
        LIDAL_REQUIRE(i < n_streams && streams[i] != nullptr, "plan_run: op %lld forks to side stream %d, which was not "
                      "given", (long long)op, i);
        if (stream_edge(i, false, (hipStream_t)stream, (hipStream_t)streams[i])) return 1;
        side_open[i] = true;
        break;
      }
      case LIDAL_OP_JOIN_SIDE: {         // the main stream waits for everything queued on side stream i so far
        const int i = sidx ? sidx : 1;
        LIDAL_REQUIRE(i < n_streams && streams[i] != nullptr, "plan_run: op %lld joins side stream %d, which was not "
                      "given", (long long)op, i);
        if (stream_edge(i, true, (hipStream_t)streams[i], (hipStream_t)stream)) return 1;
        side_open[i] = false;
        break;
      }
      default:
        set_error("plan_run: op %lld: kind %d not handled", (long long)op, kind);
        return 2;
    }
    if (rc != 0) {          // keep the callee's message, say where in the plan it happened
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", lidal_last_error());
      set_error("plan_run: op %lld (%s): %s", (long long)op, kOps[kind].name, msg);
      return rc;
    }
  }
  LIDAL_REQUIRE(pos == n_words, "plan_run: %lld ops used %lld of %lld words", (long long)n_ops, (long long)pos,
                (long long)n_words);
  return 0;
#undef P
#undef CP
#undef MP
#undef I
#undef L
#undef F
}
}  // namespace
