// Victim 3: which FORMS of the packed f32 instructions go wrong beside v_mfma_f32_16x16x32_bf16?
// report[k] counts wrong results of form k (see FORMS in run_forms.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(k, lo, hi) do { if (r.x != (lo) || r.y != (hi)) ++bad[k]; } while (0)

extern "C" __global__ void __launch_bounds__(256) victim3_kernel(const float* __restrict__ src, int64_t n, int spins,
                                                                 unsigned* __restrict__ report) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned bad[12];
  for (int k = 0; k < 12; ++k) bad[k] = 0;
  for (int it = 0; it < spins; ++it) {
    const int64_t j = (i + (int64_t)it * 8191) % (n - 8);
    float a = src[j], b = src[j + 1], c = src[j + 2], d = src[j + 3], e = src[j + 4], f = src[j + 5];
    float2v ab = {a, b}, cd = {c, d}, ef = {e, f}, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(ab), "v"(cd));  CHECK(0, a * d, a * d);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(ab), "v"(cd));              CHECK(1, a * c, a * d);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(ab), "v"(cd));              CHECK(2, a * c, b * c);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(ab), "v"(cd));  CHECK(3, b * c, b * c);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(ab), "v"(cd));  CHECK(4, b * d, a * c);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(ab), "v"(cd));              CHECK(5, a + c, b + c);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(ab), "v"(cd));  CHECK(6, a + d, b + c);
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(ab), "v"(cd), "v"(ef)); CHECK(7, fmaf(a, c, e), fmaf(a, d, f));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(ab), "v"(cd), "v"(ef)); CHECK(8, fmaf(a, d, e), fmaf(a, d, f));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(ab), "v"(cd), "v"(ef));                  CHECK(9, fmaf(a, c, e), fmaf(b, d, f));
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(ab), "v"(cd));                               CHECK(10, a * c, b * d);
    double x = (double)a * (double)c + (double)e;
    double y; asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"((double)a), "v"((double)c), "v"((double)e));
    if (x != y) ++bad[11];
  }
  for (int k = 0; k < 12; ++k) if (bad[k]) atomicAdd(&report[k], bad[k]);
}

extern "C" int victim3_launch(const float* src, int64_t n, int blocks, int spins, unsigned* report, void* stream) {
  victim3_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, n, spins, report);
  return (int)hipGetLastError();
}
