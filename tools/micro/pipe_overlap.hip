// Do f32 MFMAs and f32 VALU FMAs of DIFFERENT wavefronts on one SIMD overlap, or do their times add?
// 8 wavefronts per workgroup (2 per SIMD), 256 workgroups (one per CU).  role(w) decides what a wavefront runs:
//   mode 0: all wavefronts a dependent chain of v_mfma_f32_32x32x2_f32 (64 cycles each)
//   mode 1: all wavefronts independent v_fma_f32 chains (VALU)
//   mode 2: wavefronts 0-3 MFMA chain, 4-7 VALU (one of each per SIMD)
//   mode 3: as 2 but the MFMA is the bf16 v_mfma_f32_32x32x16_bf16 (32 cycles)
//   mode 4 / 5 / 6: only the f32-MFMA wavefronts / only the VALU wavefronts / only the bf16-MFMA wavefronts (the rest idle)
// hipcc --offload-arch=gfx950 -O3 -o pipe_overlap pipe_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3 || mode == 4 || mode == 6) && wave < 4);
  const bool do_valu = mode == 1 || ((mode == 2 || mode == 3 || mode == 5) && wave >= 4);
  float s = 0.f;
  if (do_mfma) {
    f16v acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    if (mode == 3 || mode == 6) {
      bf8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
      for (int it = 0; it < 2 * n_mfma; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
      const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
      for (int it = 0; it < n_mfma; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) s += acc[j];
  }
  if (do_valu) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3f + j;
    const float m = 1.0001f, c = 1e-5f;
    for (int it = 0; it < n_valu; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], m, c);
    }
    for (int j = 0; j < 16; ++j) s += x[j];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_mfma = 20000, n_valu = 40000;        // 20000 x 64 cycles = 1.28 M ; 40000 x 16 x 2 cycles = 1.28 M cycles
  for (int mode = 0; mode < 7; ++mode) {
    k<<<256, 512>>>(out, n_mfma, n_valu, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(out, n_mfma, n_valu, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const char* names[] = {"all f32 MFMA (2 chains per SIMD)", "all VALU fma (2 wavefronts per SIMD)", "f32 MFMA + VALU per SIMD", "bf16 MFMA + VALU per SIMD",
                           "one f32-MFMA wavefront per SIMD alone", "one VALU wavefront per SIMD alone", "one bf16-MFMA wavefront per SIMD alone"};
    printf("mode %d  %-40s %8.3f ms\n", mode, names[mode], ms);
  }
  return 0;
}
