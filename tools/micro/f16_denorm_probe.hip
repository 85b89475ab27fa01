// Does v_mfma_f32_32x32x16_f16 keep SUBNORMAL f16 inputs (|x| < 2^-14) or flush them to zero?  And what do the VALU conversions
// do (f32 -> f16 of a value in the subnormal range; FP16_OVFL saturation)?
//   hipcc --offload-arch=gfx950 -O2 -o f16_denorm_probe tools/micro/f16_denorm_probe.hip && ./f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

__global__ void k(float a_val, float b_val, float* out, int denorm_mode) {
  if (denorm_mode >= 0) __builtin_amdgcn_s_setreg((3 << 11) | (4 << 6) | 1, (unsigned)denorm_mode);   // MODE.FP_DENORM[3:0] at bits 7:4
  f2 av = {a_val, a_val};
  h2 ah = __builtin_convertvector(av, h2);
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ah[0]; b[i] = (_Float16)b_val; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)ah[0]; }
}
__global__ void kb(float a_val, float b_val, float* out) {     // bf16 with a subnormal bf16 input (2^-130)
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)a_val; b[i] = (__bf16)b_val; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void ksat(float v, float* out) {
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1u);
  f2 x = {v, -v};
  h2 h = __builtin_convertvector(x, h2);
  if (threadIdx.x == 0) { out[0] = (float)h[0]; out[1] = (float)h[1]; }
}
int main() {
  float* d; hipMalloc(&d, 64); float h[4];
  const float subs[] = {ldexpf(1.f, -15), ldexpf(1.f, -20), ldexpf(3.f, -24), ldexpf(1.f, -24)};
  for (int mode = -1; mode <= 15; mode += (mode < 0 ? 1 : 15)) {
    for (float a : subs) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.f, d, mode);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("FP_DENORM=%2d  a=%.6e (as f16: %.6e)  mfma_f16 sum_16 a*1024 = %.6e   expected %.6e\n", mode, a, h[1], h[0], 16.0 * a * 1024.0);
    }
  }
  hipLaunchKernelGGL(kb, dim3(1), dim3(64), 0, 0, ldexpf(1.f, -130), ldexpf(1.f, 100), d);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("bf16 subnormal 2^-130 * 2^100 x16 = %.6e   expected %.6e\n", h[0], 16.0 * ldexp(1.0, -30));
  hipLaunchKernelGGL(ksat, dim3(1), dim3(64), 0, 0, 1e8f, d);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("FP16_OVFL=1: cvt(1e8) = %.1f, cvt(-1e8) = %.1f\n", h[0], h[1]);
  return 0;
}
