// Victim 4 (round 6): do f64 VALU instructions (and the f32 forms the f32 BatchNorm / element-wise kernels use) go wrong
// beside another kernel's MFMA loop?  Every form is evaluated TWICE on the same operands (two asm volatile statements: a
// sporadic fault shows as a mismatch) and, where the compiler's own code gives the value, compared with that too.
// report[k] counts mismatches of form k (see FORMS in run_f64.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TWICE(k, stmt1, stmt2, eq) do { stmt1; stmt2; if (!(eq)) ++bad[k]; } while (0)

extern "C" __global__ void __launch_bounds__(256) victim4_kernel(const float* __restrict__ src, int64_t n, int spins,
                                                                 unsigned* __restrict__ report) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned bad[16];
  for (int k = 0; k < 16; ++k) bad[k] = 0;
  for (int it = 0; it < spins; ++it) {
    const int64_t j = (i + (int64_t)it * 8191) % (n - 8);
    const float fa = src[j], fb = src[j + 1], fc = src[j + 2];
    const double a = (double)fa * 1.000000123, b = (double)fb * 0.999999871 + 1e-9, c = (double)fc - 0.25;
    double r1, r2;
    float s1, s2;
    TWICE(0, asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c)),
             asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r2) : "v"(a), "v"(b), "v"(c)), r1 == r2 && r1 == __fma_rn(a, b, c));
    TWICE(1, asm volatile("v_add_f64 %0, %1, %2" : "=v"(r1) : "v"(a), "v"(b)),
             asm volatile("v_add_f64 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b)), r1 == r2 && r1 == __dadd_rn(a, b));
    TWICE(2, asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r1) : "v"(a), "v"(b)),
             asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b)), r1 == r2 && r1 == __dmul_rn(a, b));
    TWICE(3, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r1) : "v"(fa)),
             asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r2) : "v"(fa)), r1 == r2 && r1 == (double)fa);
    TWICE(4, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s1) : "v"(a)),
             asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s2) : "v"(a)), s1 == s2 && s1 == (float)a);
    TWICE(5, asm volatile("v_rcp_f64 %0, %1" : "=v"(r1) : "v"(a)),
             asm volatile("v_rcp_f64 %0, %1" : "=v"(r2) : "v"(a)), r1 == r2);
    TWICE(6, asm volatile("v_rsq_f64 %0, %1" : "=v"(r1) : "v"(a)),
             asm volatile("v_rsq_f64 %0, %1" : "=v"(r2) : "v"(a)), r1 == r2);
    TWICE(7, asm volatile("v_sqrt_f64 %0, %1" : "=v"(r1) : "v"(a)),
             asm volatile("v_sqrt_f64 %0, %1" : "=v"(r2) : "v"(a)), r1 == r2);
    {   // the compiler's f64 division and square root (v_div_scale / v_div_fmas / v_div_fixup chains), twice
      volatile double va = a, vb = b;
      const double q1 = va / vb, q2 = va / vb;
      if (q1 != q2) ++bad[8];
      const double t1 = sqrt(va), t2 = sqrt(va);
      if (t1 != t2) ++bad[9];
    }
    {   // a 64-lane f64 sum by butterfly shuffles (the reductions of the BatchNorm kernels), twice
      double x1 = a, x2 = a;
      for (int d = 32; d > 0; d >>= 1) x1 += __shfl_xor(x1, d, 64);
      for (int d = 32; d > 0; d >>= 1) x2 += __shfl_xor(x2, d, 64);
      if (x1 != x2) ++bad[10];
    }
    TWICE(11, asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(fa), "v"(fb), "v"(fc)),
              asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s2) : "v"(fa), "v"(fb), "v"(fc)), s1 == s2 && s1 == fmaf(fa, fb, fc));
    TWICE(12, asm volatile("v_rcp_f32 %0, %1" : "=v"(s1) : "v"(fa)),
              asm volatile("v_rcp_f32 %0, %1" : "=v"(s2) : "v"(fa)), s1 == s2);
    TWICE(13, asm volatile("v_rsq_f32 %0, %1" : "=v"(s1) : "v"(fa)),
              asm volatile("v_rsq_f32 %0, %1" : "=v"(s2) : "v"(fa)), s1 == s2);
    {
      volatile float va = fa, vb = fb;
      const float q1 = va / vb, q2 = va / vb;
      if (q1 != q2) ++bad[14];
    }
    TWICE(15, asm volatile("v_max_f64 %0, %1, %2" : "=v"(r1) : "v"(a), "v"(b)),
              asm volatile("v_max_f64 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b)), r1 == r2 && r1 == fmax(a, b));
  }
  for (int k = 0; k < 16; ++k) if (bad[k]) atomicAdd(&report[k], bad[k]);
}

extern "C" int victim4_launch(const float* src, int64_t n, int blocks, int spins, unsigned* report, void* stream) {
  victim4_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, n, spins, report);
  return (int)hipGetLastError();
}
