// A victim kernel for the cross-kernel disturbance seen in round 3 (scripts/exp/tiw_repro3.py): every wave holds
// known patterns in VGPRs, SGPRs and LDS, re-reads a known global buffer, waits, and counts what changed.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void __launch_bounds__(256) victim_kernel(const unsigned* __restrict__ src, int64_t n,
                                                                int spins, unsigned* __restrict__ report) {
  __shared__ unsigned lds[1024];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  // patterns
  unsigned v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { v[k] = 0xA5000000u ^ (unsigned)(i * 16 + k); asm volatile("" : "+v"(v[k])); }
  unsigned s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = __builtin_amdgcn_readfirstlane(0x5A000000u ^ (blockIdx.x * 8 + k)); asm volatile("" : "+s"(s[k])); }
  for (int k = tid; k < 1024; k += 256) lds[k] = 0xC3000000u ^ (unsigned)(k + blockIdx.x);
  __syncthreads();
  unsigned bad_load = 0, bad_mask = 0;
  for (int it = 0; it < spins; ++it) {
    // global loads of a known buffer: src[j] == j * 2654435761u
    const int64_t j = (i * 7 + it * 131) % n;
    const unsigned x = src[j];
    if (x != (unsigned)j * 2654435761u) ++bad_load;
    // a VALU compare -> SGPR mask -> select, as in the victim of the finding
    unsigned long long q = ((unsigned long long)x << 32) | (unsigned)j;
    asm volatile("" : "+v"(q));
    const unsigned sel = (q != ~0ull) ? 0x1234u : 0u;
    if (sel != 0x1234u) ++bad_mask;
    __builtin_amdgcn_s_sleep(8);
  }
  unsigned bad_v = 0, bad_s = 0, bad_l = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { asm volatile("" : "+v"(v[k])); if (v[k] != (0xA5000000u ^ (unsigned)(i * 16 + k))) ++bad_v; }
#pragma unroll
  for (int k = 0; k < 8; ++k) { asm volatile("" : "+s"(s[k])); if (s[k] != (0x5A000000u ^ (blockIdx.x * 8 + k))) ++bad_s; }
  __syncthreads();
  for (int k = tid; k < 1024; k += 256) if (lds[k] != (0xC3000000u ^ (unsigned)(k + blockIdx.x))) ++bad_l;
  if (bad_v) atomicAdd(&report[0], bad_v);
  if (bad_s) atomicAdd(&report[1], bad_s);
  if (bad_l) atomicAdd(&report[2], bad_l);
  if (bad_load) atomicAdd(&report[3], bad_load);
  if (bad_mask) atomicAdd(&report[4], bad_mask);
}

extern "C" int victim_launch(const unsigned* src, int64_t n, int blocks, int spins, unsigned* report, void* stream) {
  victim_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, n, spins, report);
  return (int)hipGetLastError();
}
