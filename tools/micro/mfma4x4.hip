// Layout and rate probe of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer products per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma4x4.hip -o tools/micro/mfma4x4 && tools/micro/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32 __attribute__((ext_vector_type(32)));

__global__ void probe(float* out) {
  const int l = threadIdx.x;
  const float a = 100.f + l, b = 1000.f + l;          // identify the supplying lane
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

__global__ void rate4(float* out, int iters) {
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0];
}

__global__ void rate32(float* out, int iters) {
  f32 c;
  for (int i = 0; i < 32; ++i) c[i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, c, 0, 0, 0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[31];
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 22);
  probe<<<1, 64>>>(d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  // expect D[b][i][j] = A[b][i] * B[b][j]; find which lanes supplied them
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int b = l / 4, j = l % 4;
      const float want = (100.f + 4 * b + i) * (1000.f + 4 * b + j);   // lane 4b+i supplies A[b][i], lane 4b+j supplies B[b][j]
      if (h[l * 4 + i] != want) { ok = 0; if (l < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, i, h[l * 4 + i], want); }
    }
  printf("layout D[block=l/4][i=reg][j=l%%4] = A(lane 4b+i) * B(lane 4b+j): %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 4;
  float ms;
  rate4<<<blocks, 256>>>(d, 10);
  hipEventRecord(e0); rate4<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 1 wave... blocks*4 waves over 1024 SIMDs = 4 waves per SIMD; MFMAs per SIMD = 4 * iters * 5
  printf("4x4x1_16b : %.3f ms  -> %.1f cycles per MFMA per SIMD at 2.4 GHz, %.1f TFLOP/s\n", ms, ms * 1e-3 * 2.4e9 / (4.0 * iters * 5),
         (double)blocks * 4 * iters * 5 * 512 / (ms * 1e-3) / 1e12);
  rate32<<<blocks, 256>>>(d, 10);
  hipEventRecord(e0); rate32<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("32x32x1_2b: %.3f ms  -> %.1f cycles per MFMA per SIMD, %.1f TFLOP/s (dependent chain, 4 waves per SIMD)\n", ms,
         ms * 1e-3 * 2.4e9 / (4.0 * iters), (double)blocks * 4 * iters * 4096 / (ms * 1e-3) / 1e12);
  return 0;
}
