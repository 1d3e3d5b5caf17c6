// Victim 2: which VALU results go wrong beside the convolution?  Each thread repeats, on known inputs: a packed f32
// multiply with op_sel (as hipcc emitted in ti_weights_kernel), a plain packed multiply, scalar multiplies, packed adds.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float float2v __attribute__((ext_vector_type(2)));

extern "C" __global__ void __launch_bounds__(256) victim2_kernel(const float* __restrict__ src, int64_t n, int spins,
                                                                 unsigned* __restrict__ report) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned bad_pk_sel = 0, bad_pk = 0, bad_mul = 0, bad_pk_add = 0, bad_cnd = 0;
  for (int it = 0; it < spins; ++it) {
    const int64_t j = (i + (int64_t)it * 8191) % (n - 4);
    float a = src[j], b = src[j + 1], c = src[j + 2], d = src[j + 3];
    float2v ab = {a, b}, cd = {c, d}, r;
    // r = (a * d, a * d): low half of src0, high half of src1 for both results
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(ab), "v"(cd));
    const float ad = a * d;
    if (r.x != ad || r.y != ad) ++bad_pk_sel;
    float2v p;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ab), "v"(cd));
    if (p.x != a * c || p.y != b * d) ++bad_pk;
    float m;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(c));
    if (m != a * c) ++bad_mul;
    float2v s;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(s) : "v"(ab), "v"(cd));
    if (s.x != a - c || s.y != b - d) ++bad_pk_add;
    unsigned long long q = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    asm volatile("" : "+v"(q));
    float e = (q != ~0ull) ? a : 0.f;
    if (e != a) ++bad_cnd;
  }
  if (bad_pk_sel) atomicAdd(&report[0], bad_pk_sel);
  if (bad_pk) atomicAdd(&report[1], bad_pk);
  if (bad_mul) atomicAdd(&report[2], bad_mul);
  if (bad_pk_add) atomicAdd(&report[3], bad_pk_add);
  if (bad_cnd) atomicAdd(&report[4], bad_cnd);
}

extern "C" int victim2_launch(const float* src, int64_t n, int blocks, int spins, unsigned* report, void* stream) {
  victim2_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, n, spins, report);
  return (int)hipGetLastError();
}
