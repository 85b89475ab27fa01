// Issue cost of the legacy v_mfma_f32_32x32x8_f16 (K = 8, two-register operands) against v_mfma_f32_32x32x16_f16 on gfx950: one
// wavefront per SIMD, back-to-back on four accumulators, wall clock ticks (s_memtime) per instruction.
// hipcc --offload-arch=gfx950 -O3 -o mfma_k8_probe mfma_k8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int K16>
__global__ __launch_bounds__(256) void k(float* out, long long* ticks, int iters) {
  f16v acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h4 a4, b4; h8 a8, b8;
  for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)(float)(threadIdx.x % 7 + j); b4[j] = (_Float16)(float)(threadIdx.x % 5 + j); }
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(float)(threadIdx.x % 7 + j); b8[j] = (_Float16)(float)(threadIdx.x % 5 + j); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (K16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[K16] = t1 - t0;
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 16);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms8, ms16;
    hipEventRecord(e0); k<0><<<1024, 256>>>(out, ticks, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms8, e0, e1);
    hipEventRecord(e0); k<1><<<1024, 256>>>(out, ticks, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms16, e0, e1);
    long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    printf("32x32x8_f16: %.3f ms, %.1f clock ticks per instruction | 32x32x16_f16: %.3f ms, %.1f ticks per instruction (1024 workgroups x 4 wavefronts, %d instructions each)\n",
           ms8, (double)h[0] / (iters * 16), ms16, (double)h[1] / (iters * 16), iters * 16);
  }
  return 0;
}
