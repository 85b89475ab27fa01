// Sustained v_mfma_f32_32x32x16_bf16 rate on the whole chip (no memory traffic): what "100 % of the matrix core"
// means for the lattice GEMMs in practice (clock under load included).  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with operands that change every instruction (pseudo-random bit patterns: realistic datapath toggling)
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_rand(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf8 a[6], b[6];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 6; ++r)
    for (int j = 0; j < 8; ++j) {
      h = h * 1664525u + 1013904223u;
      a[r][j] = (__bf16)((float)((int)(h >> 20) - 2048) * 1e-3f);
      h = h * 1664525u + 1013904223u;
      b[r][j] = (__bf16)((float)((int)(h >> 20) - 2048) * 1e-3f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(r + i) % 6], b[(r * 5 + i) % 6], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run_rand(const char* name, int wgs, int iters) {
  float* out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_rand<NACC><<<wgs, 256>>>(out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k_rand<NACC><<<wgs, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 6 * NACC * 32768.0;
    printf("%s wgs=%d iters=%d: %.3f ms  %.0f TFLOP/s (%.1f%% of 2500)\n", name, wgs, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
  }
  hipFree(out);
}

template <int NACC>
void run(const char* name, int wgs, int iters) {
  float* out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<wgs, 256>>>(out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<NACC><<<wgs, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 6 * NACC * 32768.0;
    printf("%s wgs=%d iters=%d: %.3f ms  %.0f TFLOP/s (%.1f%% of 2500)\n", name, wgs, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
  }
  hipFree(out);
}

int main_r02() {
  run<8>("8 accumulators, 2 WG/CU", 512, 2000);
  run<8>("8 accumulators, 2 WG/CU, long", 512, 20000);
  run<8>("8 accumulators, 1 WG/CU", 256, 2000);
  run<4>("4 accumulators, 2 WG/CU", 512, 4000);
  run_rand<8>("random operands, 8 accumulators, 2 WG/CU", 512, 2000);
  run_rand<8>("random operands, 8 accumulators, 2 WG/CU, long", 512, 20000);
  run_rand<8>("random operands, 8 accumulators, 2 WG/CU, short (0.3 ms)", 512, 200);
  return 0;
}

// ---- round 3: the int8 matrix core (v_mfma_i32_32x32x32_i8, twice the k of the bf16 form per instruction) with operands that
// change every instruction, and the effective shader clock of every run (s_memtime / s_memrealtime): is the "sustained" rate a
// clock (power) limit, and does int8 keep its 2x under it?
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));
struct Clk { unsigned long long cyc, rt; };

template <int NACC, bool I8>
__global__ __launch_bounds__(256, 2) void k_clk(float* out, Clk* clk, int iters) {
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float s = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (I8) {
    i16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    i4v a[6], b[6];
    for (int r = 0; r < 6; ++r)
      for (int j = 0; j < 4; ++j) {
        h = h * 1664525u + 1013904223u; a[r][j] = (int)h;                    // limbs: all 8 bits random
        h = h * 1664525u + 1013904223u; b[r][j] = (int)(h & 0x3f3f3f3fu) - 0x20202020;   // lattice integers in [-32, 31]
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(r + i) % 6], b[(r * 5 + i) % 6], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) s += (float)acc[i][j];
  } else {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf8 a[6], b[6];
    for (int r = 0; r < 6; ++r)
      for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u; a[r][j] = (__bf16)((float)((int)(h >> 20) - 2048) * 1e-3f);
        h = h * 1664525u + 1013904223u; b[r][j] = (__bf16)(float)((int)(h >> 26) - 32);
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(r + i) % 6], b[(r * 5 + i) % 6], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x].cyc = c1 - c0; clk[blockIdx.x].rt = r1 - r0; }
}

template <int NACC, bool I8>
void run_clk(const char* name, int wgs, int iters) {
  float* out; Clk* clk;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipMalloc(&clk, sizeof(Clk) * wgs);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_clk<NACC, I8><<<wgs, 256>>>(out, clk, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k_clk<NACC, I8><<<wgs, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    Clk* h = (Clk*)malloc(sizeof(Clk) * wgs);
    hipMemcpy(h, clk, sizeof(Clk) * wgs, hipMemcpyDeviceToHost);
    double ghz = 0, cyc = 0;
    for (int i = 0; i < wgs; ++i) { ghz += (double)h[i].cyc / ((double)h[i].rt * 10.0); cyc += (double)h[i].cyc; }
    free(h);
    const double ops = (double)wgs * 4 * iters * 6 * NACC * (I8 ? 65536.0 : 32768.0);
    printf("%s wgs=%d iters=%d: %.3f ms  %.0f T%s/s  effective clock %.2f GHz  %.1f cycles per MFMA per wavefront\n", name, wgs, iters, ms,
           ops / ms / 1e9, I8 ? "OP" : "FLOP", ghz / wgs, cyc / wgs / ((double)iters * 6 * NACC));
  }
  hipFree(out); hipFree(clk);
}

int main() {
  main_r02();
  run_clk<8, false>("bf16 32x32x16, random pieces x lattice ints, 2 WG/CU", 512, 4000);
  run_clk<8, true>("int8 32x32x32, random limbs x lattice ints, 2 WG/CU", 512, 4000);
  run_clk<8, false>("bf16 32x32x16, random pieces x lattice ints, 1 WG/CU", 256, 4000);
  run_clk<8, true>("int8 32x32x32, random limbs x lattice ints, 1 WG/CU", 256, 4000);
  run_clk<8, false>("bf16 32x32x16, ONE workgroup (one CU)", 1, 4000);
  run_clk<8, true>("int8 32x32x32, ONE workgroup (one CU)", 1, 4000);
  return 0;
}
