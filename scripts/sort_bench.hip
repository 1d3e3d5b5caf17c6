// Micro-benchmark: rocprim radix_sort_pairs default dispatch (merge sort below 1M items) against
// forced Onesweep, at the sizes / key widths of the kernel-map pipeline.
//   hipcc --offload-arch=gfx950 -O3 scripts/sort_bench.hip -o scripts/_abl/sort_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <rocprim/device/device_radix_sort.hpp>

template <class Config, class K>
float run(size_t n, int bits, int reps) {
  std::vector<K> h(n);
  std::mt19937_64 g(1);
  K mask = bits >= (int)(8 * sizeof(K)) ? ~K(0) : ((K(1) << bits) - 1);
  for (auto& x : h) x = (K)g() & mask;
  K *k0, *k1; int *v0, *v1;
  hipMalloc(&k0, n * sizeof(K)); hipMalloc(&k1, n * sizeof(K));
  hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
  hipMemset(v0, 0, n * 4);
  size_t tmp = 0;
  rocprim::radix_sort_pairs<Config>(nullptr, tmp, k0, k1, v0, v1, n, 0, bits, 0);
  void* t; hipMalloc(&t, tmp);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, n, 0, bits, 0);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, n, 0, bits, 0);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
  return ms / reps * 1e3f;
}

using Dflt = rocprim::default_config;
using Force = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                         rocprim::default_config, 4096>;
int main() {
  size_t ns[] = {6000, 17000, 45000, 105000, 400000, 640000, 3200000};
  printf("%10s %5s %5s %12s %12s\n", "n", "key", "bits", "default us", "onesweep us");
  for (size_t n : ns) {
    for (int bits : {8, 17, 20, 27, 32})
      printf("%10zu %5s %5d %12.1f %12.1f\n", n, "u32", bits, run<Dflt, unsigned>(n, bits, 20), run<Force, unsigned>(n, bits, 20));
    for (int bits : {39, 60})
      printf("%10zu %5s %5d %12.1f %12.1f\n", n, "u64", bits, run<Dflt, unsigned long long>(n, bits, 20), run<Force, unsigned long long>(n, bits, 20));
  }
  return 0;
}
