// Aggressors for victim2: which instruction class of the convolution kernels disturbs another wave's
// v_pk_mul_f32 ... op_sel?  mode 0: v_mfma_f32_16x16x32_bf16 loop; 1: v_mfma_f32_32x32x16_bf16 loop; 2: LDS-DMA
// (buffer_load_dwordx4 ... lds) loop; 3: ds_read_b128 loop; 4: plain VALU fma loop; 5: v_mfma_f32_16x16x16_bf16 (gfx942 form);
// 6: v_mfma_f32_16x16x4_f32
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

extern "C" __global__ void __launch_bounds__(256) aggressor_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                                   int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (mode == 0) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(src[tid + k] * 0.01f); b[k] = (__bf16)(src[tid + 8 + k] * 0.01f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (mode == 1) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(src[tid + k] * 0.01f); b[k] = (__bf16)(src[tid + 8 + k] * 0.01f); }
    f32x16 c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    acc = c0[0] + c1[5];
  } else if (mode == 2) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
    for (int it = 0; it < iters; ++it) {
      auto* dst = (__attribute__((address_space(3))) void*)(smem + (threadIdx.x >> 6) * 4096);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (tid & 63) * 16, (it & 255) * 1024, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (tid & 63) * 16, (it & 255) * 1024, 1024, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();
    acc = ((float*)smem)[tid];
  } else if (mode == 3) {
    for (int k = tid; k < 4096; k += 256) ((float*)smem)[k] = src[k];
    __syncthreads();
    float4 s = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      float4 v = *reinterpret_cast<float4*>(smem + ((tid * 16 + it * 4096) & 16383));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    acc = s.x + s.y + s.z + s.w;
  } else if (mode == 4) {
    float a = src[tid], b = src[tid + 1];
    for (int it = 0; it < iters * 8; ++it) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
    acc = a + b;
  } else if (mode == 5) {
    bf16x4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = (__bf16)(src[tid + k] * 0.01f); b[k] = (__bf16)(src[tid + 8 + k] * 0.01f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
    }
    acc = c0[0] + c1[1];
  } else if (mode == 6) {
    float a = src[tid], b = src[tid + 1];
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    }
    acc = c0[0] + c1[1];
  }
  out[(int64_t)blockIdx.x * 256 + tid] = acc;
}

extern "C" int aggressor_launch(const float* src, float* out, int blocks, int iters, int mode, void* stream) {
  aggressor_kernel<<<blocks, 256, 16384, (hipStream_t)stream>>>(src, out, iters, mode);
  return (int)hipGetLastError();
}
