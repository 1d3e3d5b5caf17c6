// Stream tables of the weight gradient (round 6): the rule lists of a stride-1 kernel map re-ordered into one run of
// 64-rule stages per workgroup of csrc/wgrad_dma.hip's wgrad_stream_kernel.
//
// Why.  gw[k] = sum over the rules (i, j) of offset k of a[i]^T b[j] reads two rows per rule.  In the offset-major
// decomposition (wgrad_dma_kernel) the ~4.7 rules that touch a row lie in different offsets, which different workgroups
// work on at different times: every row comes from the fabric once per rule (96 -> 96 on 396 662 rows: 2 x 425 MB FETCH_SIZE
// per launch for 152 MB of operands, 5 % L2 hits).  Here a row's rules meet in ONE XCD's L2 at about the same time:
//
//   * rows get a spatial key (the row index where rows are numbered in coordinate order; max over a [key_k, n] table
//     otherwise -- the strided map's inverse neighbour table gives a row's PARENT on the next level, which is numbered in
//     coordinate order -- SPVCNN's level 0 is numbered by coordinate hash); NB blocks of consecutive keys; block b belongs
//     to XCD b % 8 (workgroup w runs on the XCD of all workgroups = w mod 8: round-robin dispatch, a matter of speed only);
//   * the rules are sorted by (XCD, offset, block of the OUTPUT row), stable: per (XCD, offset) one list in block order;
//   * the W / 8 workgroups of an XCD divide the 27 offsets between them in proportion to the XCD's own rule counts:
//     slots in 1/65536 units, at least one slot per offset that has rules, so that a workgroup serves at most two
//     offsets (two accumulator sets); an offset's list is cut into 32-rule steps (one MFMA reduction step) and the steps
//     are dealt to the workgroups that share the offset by a low-discrepancy sequence, so every one of them walks the
//     WHOLE list -- all blocks, in order -- at the pace of the others;
//   * a workgroup's steps (of both its sets) are stored in block order; bit 31 of a rule's first index = its set; a
//     partial last step of a list and an odd number of steps are padded with out-of-range rules.
//
// Everything is computed on the device from koff: nothing travels through the host, the launches are sized for the
// capacity k * n_rows of the rule list (sort.hip takes the live counts from device memory).  Deterministic: stable sorts,
// integer arithmetic, counts by integer atomics.
#include "common.h"

using namespace lidal;

namespace {

constexpr int UNIT = 65536;          // fixed-point slot
constexpr int MAXK = 32;             // offsets (5 bits of the sort key)
constexpr int MAXBX = 64;            // blocks per XCD (6 bits)
constexpr int HDR = 4;               // words ahead of soff in the descriptor (wgrad_dma.hip STREAM_HDR)
constexpr int BLOCK_ROWS = 1024;     // rows per block where that leaves <= 64 blocks per XCD (scripts/exp/wgrad_streams.py: 512 -- 4096
                                     // within 10 % of each other on 397 k rows; smaller blocks even out the XCDs' shares)

struct Ws {
  int64_t off_rowblk, off_key1, off_val1, off_key1s, off_val1s, off_cnt, off_tab, off_tcount, off_perw, off_wfirst, off_key2,
      off_val2, off_key2s, off_val2s, off_dest, off_sort1, off_sort2, total;
  int64_t m_cap, t_cap;
};
// device tables of the plan kernel (i32 words)
constexpr int TAB_LSTART = 0, TAB_SBASE = 256, TAB_START = 512, TAB_LEN = 768, TAB_FIRST = 1024, TAB_WORDS = 1024 + 8 * 64 * 2;

Ws layout(int64_t n_rows, int k) {
  Ws w;
  w.m_cap = n_rows * k;
  w.t_cap = cdiv(w.m_cap, 32) + 8 * MAXK;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t at = o; o += align_up(bytes, 256); return at; };
  w.off_rowblk = take(n_rows * 4);
  w.off_key1 = take(w.m_cap * 4);
  w.off_val1 = take(w.m_cap * 4);
  w.off_key1s = take(w.m_cap * 4);
  w.off_val1s = take(w.m_cap * 4);
  w.off_cnt = take(256 * 4);
  w.off_tab = take(TAB_WORDS * 4);
  w.off_tcount = take(8);
  w.off_perw = take(2048 * 4);
  w.off_wfirst = take(2048 * 4);
  w.off_key2 = take(w.t_cap * 4);
  w.off_val2 = take(w.t_cap * 4);
  w.off_key2s = take(w.t_cap * 4);
  w.off_val2s = take(w.t_cap * 4);
  w.off_dest = take(w.t_cap * 4);
  w.off_sort1 = take(radix_sort_ws_bytes(w.m_cap, 4, true));
  w.off_sort2 = take(radix_sort_ws_bytes(w.t_cap, 4, true));
  w.total = o;
  return w;
}

// block of every row: key * nb / key_range, key = the row index or the maximum over the key_k rows of key_tab [key_k, n]
__global__ void __launch_bounds__(256) rowblk_kernel(const int* __restrict__ key_tab, int key_k, int64_t n, int64_t key_range,
                                                     int nb, int* __restrict__ rowblk) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t key = i;
  if (key_tab != nullptr) {
    int best = -1;
    for (int kk = 0; kk < key_k; ++kk) {
      const int v = key_tab[(int64_t)kk * n + i];
      best = v > best ? v : best;
    }
    key = best < 0 ? 0 : best;
  }
  int64_t b = key * nb / key_range;
  rowblk[i] = (int)(b >= nb ? nb - 1 : b);
}

// sort key of every rule: (XCD, offset, block in the XCD) of its output row; rules of (XCD, offset) counted
__global__ void __launch_bounds__(256) rule_key_kernel(const int2* __restrict__ pairs, const int64_t* __restrict__ koff, int K,
                                                       const int* __restrict__ rowblk, unsigned* __restrict__ key1,
                                                       int* __restrict__ val1, int* __restrict__ cnt) {
  __shared__ int64_t sk[MAXK + 1];
  __shared__ int h[256];
  const int tid = threadIdx.x;
  if (tid <= K) sk[tid] = koff[tid];
  h[tid] = 0;
  __syncthreads();
  const int64_t m = sk[K];
  const int lane = tid & 63;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < m; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + tid;
    int bin = -1;
    if (i < m) {
      int lo = 0, hi = K - 1;               // the offset whose rules hold position i (empty offsets share a start: the last one)
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sk[mid] <= i) lo = mid; else hi = mid - 1;
      }
      const int b = rowblk[pairs[i].y];
      const int x = b & 7;
      bin = x * MAXK + lo;
      key1[i] = (unsigned)((bin * MAXBX) + (b >> 3));
      val1[i] = (int)i;
    }
    // a wave's rules share their offset and fall on 8 XCDs: one LDS add per distinct bin, not one per lane
    unsigned long long todo = __ballot(bin >= 0);
    while (todo) {
      const int b0 = __builtin_amdgcn_readlane(bin, __builtin_ctzll(todo));
      const unsigned long long same = __ballot(bin == b0);
      if (lane == __builtin_ctzll(todo)) atomicAdd(&h[b0], (int)__popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (h[tid]) atomicAdd(&cnt[tid], h[tid]);
}

__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// exclusive prefix over the 256 threads of the workgroup (wsum: 4 words of LDS); all threads call it
__device__ __forceinline__ int scan256(int v, int* wsum, int tid, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) off += (w < wave) ? wsum[w] : 0;
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return off + incl - v;
}

// one workgroup: slots per (XCD, offset), list / step prefixes, workgroup -> offsets, the reducer's table
__global__ void __launch_bounds__(256) plan_kernel(const int* __restrict__ cnt, int K, int W, int* __restrict__ tab,
                                                   int64_t* __restrict__ tcount, int* __restrict__ sdesc, int* __restrict__ perw) {
  __shared__ int s_len[256], s_start[256], s_first[8 * 64], s_second[8 * 64], wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wx = W / 8;
  for (int i = tid; i < 2048; i += 256) perw[i] = 0;
  // slots: wave v serves XCDs 2 v and 2 v + 1, lane k = offset k
  for (int x = 2 * wave; x < 2 * wave + 2; ++x) {
    const int c = lane < K ? cnt[x * MAXK + lane] : 0;
    const int64_t total = wave_sum64(c);
    bool clamped = false;
    int64_t free = wx, rest = total;
    for (int it = 0; it < K; ++it) {              // to the fixed point: every offset with rules ends with >= one slot
      const bool move = !clamped && c > 0 && (int64_t)c * free < rest;
      clamped = clamped || move;
      free = wx - __popcll(__ballot(clamped));
      rest = total - wave_sum64(clamped ? c : 0);
      if (__ballot(move) == 0ull) break;
    }
    int len = 0;
    if (c > 0) len = clamped ? UNIT : (int)((int64_t)c * free * UNIT / (rest > 0 ? rest : 1));
    int incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane < MAXK) {
      s_len[x * MAXK + lane] = len;
      s_start[x * MAXK + lane] = incl - len;
    }
  }
  __syncthreads();
  for (int i = tid; i < 8 * 64; i += 256) {       // slot j of XCD x: the offset that holds its start, the one that starts inside
    const int x = i >> 6, j = i & 63;
    int f = -1, sec = -1;
    if (j < wx)
      for (int k = 0; k < K; ++k) {
        const int st = s_start[x * MAXK + k], len = s_len[x * MAXK + k];
        if (len == 0) continue;
        if (st <= j * UNIT && j * UNIT < st + len) f = k;
        if (j * UNIT < st && st < (j + 1) * UNIT) sec = k;
      }
    s_first[i] = f;
    s_second[i] = sec;
  }
  // prefixes in sorted order: XCD-major, offset-minor
  const int c = cnt[tid];
  int t_total = 0;
  const int l = scan256(c, wsum, tid, nullptr);
  const int st = scan256((c + 31) / 32, wsum, tid, &t_total);
  if (tid == 0) *tcount = t_total;
  tab[TAB_LSTART + tid] = l;
  tab[TAB_SBASE + tid] = st;
  tab[TAB_START + tid] = s_start[tid];
  tab[TAB_LEN + tid] = s_len[tid];
  for (int i = tid; i < 8 * 64; i += 256) {
    tab[TAB_FIRST + i] = s_first[i];
    tab[TAB_FIRST + 8 * 64 + i] = s_second[i];
  }
  // descriptor: [W, K, stages (offsets_kernel), 0] soff[W + 1] wk[W][2] kred[K][8][3]
  if (tid == 0) { sdesc[0] = W; sdesc[1] = K; sdesc[3] = 0; }
  int* wk = sdesc + HDR + (W + 1);
  for (int w = tid; w < W; w += 256) {
    wk[2 * w] = s_first[(w & 7) * 64 + (w >> 3)];
    wk[2 * w + 1] = s_second[(w & 7) * 64 + (w >> 3)];
  }
  int* kred = wk + 2 * W;
  if (tid < K * 8) {
    const int k = tid >> 3, x = tid & 7;
    const int st0 = s_start[x * MAXK + k], len = s_len[x * MAXK + k];
    int j0 = 0, nj = 0, set0 = 0;
    if (len > 0) {
      j0 = st0 >> 16;
      nj = ((st0 + len - 1) >> 16) - j0 + 1;
      set0 = s_first[x * 64 + j0] != k;
    }
    kred[(k * 8 + x) * 3] = j0;
    kred[(k * 8 + x) * 3 + 1] = nj;
    kred[(k * 8 + x) * 3 + 2] = set0;
  }
}

// every 32-rule step: its workgroup, set and the block of its first rule -> second sort key; steps per workgroup counted
__global__ void __launch_bounds__(256) step_key_kernel(const unsigned* __restrict__ key1s, const int* __restrict__ tab,
                                                       const int64_t* __restrict__ tcount, unsigned* __restrict__ key2,
                                                       int* __restrict__ val2, int* __restrict__ perw) {
  __shared__ int s_sbase[256];
  const int tid = threadIdx.x;
  s_sbase[tid] = tab[TAB_SBASE + tid];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + tid;
  int w = -1;
  if (t < *tcount) {
    int lo = 0, hi = 255;                       // last (XCD, offset) whose first step is <= t (empty lists share a start: take the last)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_sbase[mid] <= (int)t) lo = mid; else hi = mid - 1;
    }
    const int xk = lo, x = xk / MAXK, k = xk % MAXK;
    const unsigned s_in = (unsigned)((int)t - s_sbase[xk]);
    const int first_rule = tab[TAB_LSTART + xk] + 32 * (int)s_in;
    const int blk = (int)(key1s[first_rule] & (MAXBX - 1));
    const unsigned u = s_in * 2654435769u;
    const int pos = tab[TAB_START + xk] + (int)(((unsigned long long)u * (unsigned)tab[TAB_LEN + xk]) >> 32);
    const int j = pos >> 16;
    const int set = tab[TAB_FIRST + x * 64 + j] != k;
    w = 8 * j + x;
    key2[t] = (unsigned)(((w * MAXBX) + blk) * 2 + set);
    val2[t] = (int)t;
  }
  // consecutive steps belong to one list and go to the few workgroups that share its offset: one add per distinct workgroup
  unsigned long long todo = __ballot(w >= 0);
  while (todo) {
    const int w0 = __builtin_amdgcn_readlane(w, __builtin_ctzll(todo));
    const unsigned long long same = __ballot(w == w0);
    if ((tid & 63) == __builtin_ctzll(todo)) atomicAdd(&perw[w0], (int)__popcll(same));
    todo &= ~same;
  }
}

// one workgroup: stages and first step of every workgroup (thread t: workgroups 8 t .. 8 t + 7)
__global__ void __launch_bounds__(256) offsets_kernel(const int* __restrict__ perw, int W, int* __restrict__ sdesc,
                                                      int* __restrict__ wfirst) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x;
  int st[8], fs[8], a = 0, b = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int w = 8 * tid + e;
    const int c = w < W ? perw[w] : 0;
    st[e] = a;
    fs[e] = b;
    a += (c + 1) / 2;
    b += c;
  }
  int total = 0;
  const int a0 = scan256(a, wsum, tid, &total);
  const int b0 = scan256(b, wsum, tid, nullptr);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int w = 8 * tid + e;
    if (w < W) {
      sdesc[HDR + w] = a0 + st[e];
      wfirst[w] = b0 + fs[e];
    }
  }
  if (tid == 0) { sdesc[HDR + W] = total; sdesc[2] = total; }
}

// where every step goes; the padding of a list's partial last step and of a workgroup's odd step
__global__ void __launch_bounds__(256) place_kernel(const unsigned* __restrict__ key2s, const int* __restrict__ val2s,
                                                    const int64_t* __restrict__ tcount, const int* __restrict__ sdesc,
                                                    const int* __restrict__ wfirst, const int* __restrict__ perw, int W,
                                                    int* __restrict__ dest, int2* __restrict__ spairs) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < W && (perw[p] & 1)) {                // the second half of the workgroup's last stage
    int2* q = spairs + ((int64_t)sdesc[HDR + p + 1] * 64 - 32);
    for (int e = 0; e < 32; ++e) q[e] = make_int2(0x7FFFFFFF, 0x7FFFFFFF);
  }
  if (p >= *tcount) return;
  const int w = (int)(key2s[p] >> 7);
  dest[val2s[p]] = sdesc[HDR + w] * 2 + ((int)p - wfirst[w]);
}

__global__ void __launch_bounds__(256) scatter_kernel(const int2* __restrict__ pairs, const int64_t* __restrict__ koff, int K,
                                                      const unsigned* __restrict__ key1s, const int* __restrict__ val1s,
                                                      const unsigned* __restrict__ key2, const int* __restrict__ tab,
                                                      const int* __restrict__ cnt, const int* __restrict__ dest,
                                                      int2* __restrict__ spairs) {
  const int64_t m = koff[K];
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < m; g += (int64_t)gridDim.x * 256) {
  const int xk = (int)(key1s[g] >> 6);
  const int r_in = (int)g - tab[TAB_LSTART + xk];
  const int t = tab[TAB_SBASE + xk] + (r_in >> 5);
  const int64_t at = (int64_t)dest[t] * 32 + (r_in & 31);
  const unsigned set = key2[t] & 1u;
  const int2 r = pairs[val1s[g]];
  spairs[at] = make_int2((int)((unsigned)r.x | (set << 31)), r.y);
  if (r_in + 1 == cnt[xk])                    // the list's last rule pads the rest of its step
    for (int e = (r_in & 31) + 1; e < 32; ++e) spairs[at - (r_in & 31) + e] = make_int2((int)(0x7FFFFFFFu | (set << 31)), 0x7FFFFFFF);
  }
}

int blocks_for(int64_t n_rows) {
  int64_t bx = cdiv(n_rows, 8ll * BLOCK_ROWS);
  if (bx < 1) bx = 1;
  if (bx > MAXBX) bx = MAXBX;
  return (int)bx * 8;
}

}  // namespace

extern "C" int lidal_wgrad_streams_workgroups(void) { return 512; }       // two per CU; W / 8 = 64 slots per XCD >= 27 offsets

extern "C" int64_t lidal_wgrad_streams_rules(int64_t n_rows, int k, int n_wg) {
  return n_rows * k + 32ll * 8 * MAXK + 64ll * n_wg;            // every rule, a partial step per list, a padded stage per workgroup
}
extern "C" int64_t lidal_wgrad_streams_desc_words(int k, int n_wg) { return HDR + (n_wg + 1) + 2ll * n_wg + 24ll * k; }
extern "C" int64_t lidal_wgrad_streams_workspace_bytes(int64_t n_rows, int k) { return layout(n_rows, k).total; }

extern "C" int lidal_wgrad_streams_build(const int32_t* pairs, const int64_t* koff, int k, int64_t n_rows, const int32_t* key_tab,
                                         int key_k, int64_t key_range, int n_wg, int32_t* spairs, int64_t spairs_rules,
                                         int32_t* sdesc, void* ws, int64_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  LIDAL_REQUIRE(k >= 1 && k <= MAXK, "wgrad_streams_build: %d offsets (1..%d)", k, MAXK);
  LIDAL_REQUIRE(n_wg >= 8 * k && n_wg % 8 == 0 && n_wg <= 2048 && n_wg / 8 <= 64,
                "wgrad_streams_build: %d workgroups (a multiple of 8, 8 k .. 512)", n_wg);
  LIDAL_REQUIRE(n_rows > 0 && n_rows * k < (1ll << 30), "wgrad_streams_build: %lld rows x %d offsets", (long long)n_rows, k);
  LIDAL_REQUIRE(key_tab == nullptr ? key_range == n_rows : (key_k >= 1 && key_range >= 1),
                "wgrad_streams_build: key_range %lld (the row count without a key table)", (long long)key_range);
  LIDAL_REQUIRE(spairs_rules >= lidal_wgrad_streams_rules(n_rows, k, n_wg), "wgrad_streams_build: room for %lld rules, %lld needed",
                (long long)spairs_rules, (long long)lidal_wgrad_streams_rules(n_rows, k, n_wg));
  const Ws L = layout(n_rows, k);
  LIDAL_REQUIRE(ws_bytes >= L.total, "wgrad_streams_build: workspace too small: %lld < %lld", (long long)ws_bytes, (long long)L.total);
  char* w = (char*)ws;
  int* rowblk = (int*)(w + L.off_rowblk);
  unsigned* key1 = (unsigned*)(w + L.off_key1);
  int* val1 = (int*)(w + L.off_val1);
  unsigned* key1s = (unsigned*)(w + L.off_key1s);
  int* val1s = (int*)(w + L.off_val1s);
  int* cnt = (int*)(w + L.off_cnt);
  int* tab = (int*)(w + L.off_tab);
  int64_t* tcount = (int64_t*)(w + L.off_tcount);
  int* perw = (int*)(w + L.off_perw);
  int* wfirst = (int*)(w + L.off_wfirst);
  unsigned* key2 = (unsigned*)(w + L.off_key2);
  int* val2 = (int*)(w + L.off_val2);
  unsigned* key2s = (unsigned*)(w + L.off_key2s);
  int* val2s = (int*)(w + L.off_val2s);
  int* dest = (int*)(w + L.off_dest);
  const int nb = blocks_for(n_rows);
  LIDAL_HIP(hipMemsetAsync(cnt, 0, 256 * 4, s));
  rowblk_kernel<<<(unsigned)cdiv(n_rows, 256), 256, 0, s>>>(key_tab, key_k, n_rows, key_range, nb, rowblk);
  // (1024 workgroups: each ends with up to 216 global adds on the same 216 counters -- 65 us with 4096 of them)
  rule_key_kernel<<<(unsigned)(cdiv(L.m_cap, 256) < 1024 ? cdiv(L.m_cap, 256) : 1024), 256, 0, s>>>((const int2*)pairs, koff, k, rowblk, key1, val1, cnt);
  LIDAL_CHECK_LAUNCH("wgrad_streams_build (keys)");
  if (int rc = radix_sort(key1, val1, key1s, val1s, L.m_cap, 4, 14, w + L.off_sort1, L.off_sort2 - L.off_sort1, s, koff + k)) return rc;
  plan_kernel<<<1, 256, 0, s>>>(cnt, k, n_wg, tab, tcount, sdesc, perw);
  step_key_kernel<<<(unsigned)cdiv(L.t_cap, 256), 256, 0, s>>>(key1s, tab, tcount, key2, val2, perw);
  LIDAL_CHECK_LAUNCH("wgrad_streams_build (steps)");
  if (int rc = radix_sort(key2, val2, key2s, val2s, L.t_cap, 4, 16, w + L.off_sort2, L.total - L.off_sort2, s, tcount)) return rc;
  offsets_kernel<<<1, 256, 0, s>>>(perw, n_wg, sdesc, wfirst);
  place_kernel<<<(unsigned)cdiv(L.t_cap > n_wg ? L.t_cap : n_wg, 256), 256, 0, s>>>(key2s, val2s, tcount, sdesc, wfirst, perw, n_wg, dest,
                                                                                   (int2*)spairs);
  scatter_kernel<<<(unsigned)(cdiv(L.m_cap, 256) < 4096 ? cdiv(L.m_cap, 256) : 4096), 256, 0, s>>>((const int2*)pairs, koff, k, key1s, val1s, key2, tab, cnt, dest, (int2*)spairs);
  LIDAL_CHECK_LAUNCH("wgrad_streams_build (scatter)");
  return 0;
}
