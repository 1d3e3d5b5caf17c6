// Stable LSD radix sort of (u32 key, i32 value) pairs for the sizes of this path (10^3 .. 5*10^6 items,
// 8 .. 32 significant key bits), written for the occupancy-pattern row orders (lidal_kmap_order).
//
// Where it stands (scripts/sort_bench.py, us per sort, against rocPRIM through torch.sort):
//     397k pairs,  8 bits   28 (Onesweep path of rocPRIM: 36)      226k pairs, 8 bits   19
//     397k pairs, 27 bits  168 (merge sort: 114-137)               3.2M pairs, 19 bits 364 (175)
// One pass is competitive; several are not: every pass scatters its items as single 4-byte writes
// (a 1024-item round holds ~4 items per digit, so runs are 16 bytes), where Onesweep first orders
// ~8k items per workgroup in LDS and writes runs of 128 bytes.  (Tried here too: 8192-item rounds
// ordered in LDS before the write-out -- 177 us for the 27-bit case: with coalesced writes the
// counting of the next digit at the landing place, ~400k contended global atomics per pass, is what
// remains; counting in a launch of its own instead costs what a pass costs.)  The library therefore uses this
// sort for the 8-bit occupancy masks of the 2x2x2 maps only; the 27-bit masks, the voxel
// lists and the 60-bit coordinate keys stay on rocPRIM.  1 + P launches for P = ceil(bits / 8):
//   * the items are cut into NB <= 256 contiguous ranges, one workgroup each;
//   * `hist0` counts the first digit per range (a private row per workgroup: no atomics) and zeroes
//     the rows of the later digits' tables;
//   * pass p: every workgroup derives its 256 start offsets from the table of digit p (thread d adds
//     the column of digit d over the ranges before its own: NB coalesced 1-KiB reads), then walks its
//     range in index order, 1024 items per round: a wave ranks each 64 of its 256 items among equal
//     digits with 8 ballots (multisplit), the waves' counts are prefixed through LDS, and each item
//     is written to its final place of this pass -- where its NEXT digit is counted
//     into the next table for the range it landed in (integer atomics: the counts do not depend on
//     the order of arrival), so no pass needs a counting launch of its own.
// Stable (equal keys keep their input order), hence the output is unique: bit-equal to any other
// stable sort (tests/test_ops_gpu.py compares with torch.sort(stable=True)).
#include "common.h"

using namespace lidal;

namespace {

constexpr int SB = 256;          // threads per workgroup = items per round = radix
constexpr int MAX_RANGES = 256;

struct Plan { int nb; int64_t per; int passes; };

static Plan plan_for(int64_t n, int bits) {
  Plan p;
  p.passes = (bits + 7) / 8;
  if (p.passes < 1) p.passes = 1;
  // ranges: >= 2 rounds each; up to 64 (one batch of table rows per workgroup) while that keeps a
  // range within ~8 rounds, then up to MAX_RANGES
  int64_t nb = cdiv(n, 2 * 1024);
  if (nb < 1) nb = 1;
  if (nb > 64) nb = cdiv(n, 8 * 1024) > 64 ? cdiv(n, 8 * 1024) : 64;
  if (nb > MAX_RANGES) nb = MAX_RANGES;
  p.per = align_up(cdiv(n, nb), SB);          // (ranges need not be whole rounds)
  p.nb = (int)cdiv(n, p.per);
  return p;
}

__global__ void __launch_bounds__(SB) sort_hist0_kernel(const unsigned* __restrict__ keys, int64_t n,
                                                        int64_t per, int passes,
                                                        int* __restrict__ tables) {
  __shared__ int h[256];
  const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int64_t beg = (int64_t)b * per, end = (beg + per < n) ? beg + per : n;
  for (int64_t i = beg + tid; i < end; i += SB) atomicAdd(&h[keys[i] & 255u], 1);
  __syncthreads();
  tables[(int64_t)b * 256 + tid] = h[tid];
  for (int p = 1; p < passes; ++p) tables[((int64_t)p * nb + b) * 256 + tid] = 0;
}

// One pass.  A round = ROUND = 1024 consecutive items: wave w owns items [256 w, 256 w + 256) of the
// round and walks them as four 64-item steps, so the items of a digit keep their index order when
//   place = start of the digit for this range (off)  +  the digit's count in the earlier waves of
//           the round (phase 2)  +  its count in the wave's earlier steps  +  rank inside the step.
// Three barriers per 1024 items (the first form of this kernel took a turn per wave and per 256
// items, 16 barriers per 1024, with the loads of a round exposed: 110 us for 396k 27-bit pairs).
constexpr int ROUND = 1024;
__global__ void __launch_bounds__(SB) sort_pass_kernel(const unsigned* __restrict__ kin,
                                                       const int* __restrict__ vin,
                                                       unsigned* __restrict__ kout,
                                                       int* __restrict__ vout, int64_t n, int64_t per,
                                                       int shift, const int* __restrict__ table,
                                                       int* __restrict__ next_table) {
  __shared__ int off[256];            // running start of every digit for this range
  __shared__ int scan[256];
  __shared__ int wh[4][256];          // per wave: digit counts of the round, then running places
  const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // start offset of digit `tid` for this range: all smaller digits everywhere + this digit in the
  // ranges before this one
  // (up to 64 coalesced 1-KiB rows in flight per batch: with 8 in flight the 129 rows of a 396k-item
  // sort were 16 dependent batches, ~13 us of every pass)
  int before = 0, total = 0;
  for (int r0 = 0; r0 < nb; r0 += 64) {
    int c[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) c[u] = (r0 + u < nb) ? table[(int64_t)(r0 + u) * 256 + tid] : 0;
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      before += (r0 + u < b) ? c[u] : 0;
      total += c[u];
    }
  }
  scan[tid] = total;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? scan[tid - d] : 0;
    __syncthreads();
    scan[tid] += t;
    __syncthreads();
  }
  off[tid] = scan[tid] - total + before;

  const int64_t beg = (int64_t)b * per, end = (beg + per < n) ? beg + per : n;
  // this lane's four items of a round: steps t = 0..3, item = i0 + 256 wave + 64 t + lane
  unsigned key[4], nkey[4];
  int val[4], nval[4];
  auto load = [&](int64_t i0, unsigned (&k)[4], int (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t i = i0 + 256 * wave + 64 * t + lane;
      const bool ok = i < end;
      k[t] = ok ? kin[i] : 0u;
      v[t] = ok ? vin[i] : 0;
    }
  };
  load(beg, key, val);
  for (int64_t i0 = beg; i0 < end; i0 += ROUND) {
    if (i0 + ROUND < end) load(i0 + ROUND, nkey, nval);        // next round's items travel meanwhile
#pragma unroll
    for (int w = 0; w < 4; ++w) wh[w][tid] = 0;
    __syncthreads();
    // phase 1: digit counts of this wave's 256 items
    unsigned long long peers[4];
    int rank[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool valid = i0 + 256 * wave + 64 * t + lane < end;
      const unsigned digit = (key[t] >> shift) & 255u;
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool set = (digit >> bit) & 1u;
        const unsigned long long bb = __ballot(set);
        m &= set ? bb : ~bb;
      }
      peers[t] = valid ? m : 0ull;
      rank[t] = __popcll(m & ((1ull << lane) - 1ull));
      if (valid && rank[t] == 0) wh[wave][digit] += __popcll(m);       // one lane per digit and step; steps in order
    }
    __syncthreads();
    // phase 2: wave w's places of digit `tid` start behind the earlier waves'; the range's running
    // start moves past the round
    {
      int run = off[tid];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int c = wh[w][tid];
        wh[w][tid] = run;
        run += c;
      }
      off[tid] = run;
    }
    __syncthreads();
    // phase 3: the wave's steps in order; the leader lane of a digit takes its places and moves
    // the wave's running place past them
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned digit = (key[t] >> shift) & 255u;
      const bool valid = peers[t] != 0ull;
      int base = 0;
      if (valid && rank[t] == 0) {
        base = wh[wave][digit];
        wh[wave][digit] = base + __popcll(peers[t]);
      }
      base = __shfl(base, valid ? __builtin_ctzll(peers[t]) : 0, 64);
      if (valid) {
        const int64_t pos = (int64_t)base + rank[t];
        kout[pos] = key[t];
        vout[pos] = val[t];
        if (next_table != nullptr)
          atomicAdd(&next_table[(pos / per) * 256 + ((key[t] >> (shift + 8)) & 255u)], 1);
      }
    }
    __syncthreads();                    // wh is zeroed again at the top
#pragma unroll
    for (int t = 0; t < 4; ++t) { key[t] = nkey[t]; val[t] = nval[t]; }
  }
}

}  // namespace

namespace lidal {

// scratch: ping-pong buffers for keys and values + the digit tables
int64_t sort_pairs_ws_bytes(int64_t n) {
  const int64_t q = n > 0 ? n : 1;
  return 2 * align_up(4 * q, 256) + align_up((int64_t)4 * MAX_RANGES * 256 * 4, 256);
}

// keys_in / vals_in are not written; keys_out / vals_out receive the sorted pairs (`bits` low key bits
// significant, the higher ones must be zero).  keys_out is also the ping-pong partner of the
// scratch buffer, so it cannot be omitted.
int sort_pairs_u32(const unsigned* keys_in, const int* vals_in, unsigned* keys_out, int* vals_out,
                   int64_t n, int bits, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (n == 0) return 0;
  LIDAL_REQUIRE(bits >= 1 && bits <= 32 && n < (1ll << 31), "sort: %d bits, %lld items", bits, (long long)n);
  LIDAL_REQUIRE(ws_bytes >= sort_pairs_ws_bytes(n), "sort workspace too small");
  const Plan p = plan_for(n, bits);
  const int64_t a = align_up(4 * n, 256);
  unsigned* ktmp = (unsigned*)ws;
  int* vtmp = (int*)((char*)ws + a);
  int* tables = (int*)((char*)ws + 2 * a);
  sort_hist0_kernel<<<p.nb, SB, 0, s>>>(keys_in, n, p.per, p.passes, tables);
  LIDAL_CHECK_LAUNCH("sort_hist0");
  const unsigned* kin = keys_in;
  const int* vin = vals_in;
  for (int pass = 0; pass < p.passes; ++pass) {
    // the last pass must land in the caller's buffers: alternate backwards from there
    const bool to_out = ((p.passes - 1 - pass) & 1) == 0;
    unsigned* ko = to_out ? keys_out : ktmp;
    int* vo = to_out ? vals_out : vtmp;
    sort_pass_kernel<<<p.nb, SB, 0, s>>>(kin, vin, ko, vo, n, p.per, 8 * pass,
                                         tables + (int64_t)pass * p.nb * 256,
                                         pass + 1 < p.passes ? tables + (int64_t)(pass + 1) * p.nb * 256 : nullptr);
    LIDAL_CHECK_LAUNCH("sort_pass");
    kin = ko;
    vin = vo;
  }
  return 0;
}

}  // namespace lidal

// test / bench entry: sorts device pairs, stable, by the low `bits` bits of the keys
extern "C" int64_t lidal_sort_pairs_workspace_bytes(int64_t n) { return sort_pairs_ws_bytes(n); }
extern "C" int lidal_sort_pairs(const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out,
                                int32_t* vals_out, int64_t n, int bits, void* ws, int64_t ws_bytes,
                                void* stream) {
  return sort_pairs_u32(keys_in, vals_in, keys_out, vals_out, n, bits, ws, ws_bytes, (hipStream_t)stream);
}
