// Matrix-core / VALU overlap on one SIMD, measured in SHADER CYCLES (s_memtime) per wavefront, with the effective clock
// (cycles / s_memrealtime) beside it -- the round-2 measurement (pipe_overlap.hip) was wall-clock only and could not tell an
// issue-port limit from a clock/power limit.
//
//   part A  two wavefronts per SIMD with different roles: wavefronts 0-3 a chain of MFMAs (f32 32x32x2 or bf16 32x32x16,
//           1/2/4 independent accumulators), wavefronts 4-7 independent v_fma_f32 at 100/50/25 % of a wavefront's issue
//           rate (s_nop fill).  Each role alone, then together; loop counts calibrated so both run ~1 M cycles alone.
//   part B  ONE instruction stream per wavefront: every MFMA followed by KV v_fma_f32 (KV = 0..16), one and two
//           wavefronts per SIMD -- what a kernel like k_mid_fit_v5 does.
//   every line on 1 workgroup (one CU) and on 256 / 1024 workgroups (all CUs).
//
// hipcc --offload-arch=gfx950 -O3 -o pipe_overlap_cycles pipe_overlap_cycles.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

struct Rec { unsigned long long cyc, rt; };

__device__ __forceinline__ unsigned long long t_cyc() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long t_real() { return __builtin_amdgcn_s_memrealtime(); }

#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))

template <int N> __device__ __forceinline__ void nops() {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("s_nop 3");
}

// KIND 1: v_mfma_f32_32x32x2_f32, 2: v_mfma_f32_32x32x16_bf16.  One "iteration" = NACC MFMAs, each followed by KV v_fma.
template <int KIND, int NACC, int KV>
__device__ float role_mfma(int n, unsigned tid) {
  f16v acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;
  float x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = tid * 1e-3f + j;
  const float m = 1.0001f, c = 1e-5f;
  // operands differ per lane and are perturbed every iteration through the VALU values when KV > 0 (no constant folding)
  float fa = tid * 1e-3f + 0.5f, fb = 1.0f + tid * 1e-4f;
  bf8 ba, bb;
  for (int j = 0; j < 8; ++j) { ba[j] = (__bf16)(float)((tid * 7 + j) & 63); bb[j] = (__bf16)(float)((tid * 3 + j) & 31); }
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      if (KIND == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
      else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[a], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < KV; ++v) VFMA(x[(a * KV + v) & 15]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
    for (int j = 0; j < 16; ++j) s += acc[a][j];
#pragma unroll
  for (int j = 0; j < 16; ++j) s += x[j];
  return s;
}

// 16 independent v_fma_f32 chains; NOPS x "s_nop 3" after every instruction (0: full single-wavefront issue rate)
template <int NOPS>
__device__ float role_valu(int n, unsigned tid) {
  float x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = tid * 1e-3f + j;
  const float m = 1.0001f, c = 1e-5f;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { VFMA(x[j]); nops<NOPS>(); }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += x[j];
  return s;
}

// roles of the two wavefront halves (0-3 / 4-7): bit 0 = half 0 runs the MFMA role, bit 1 = half 1 runs the VALU role,
// bit 2 = half 1 ALSO runs the MFMA role (part B with two wavefronts per SIMD)
template <int KIND, int NACC, int KV, int NOPS>
__global__ __launch_bounds__(512) void k(float* out, Rec* rec, int n_mfma, int n_valu, int roles) {
  const unsigned tid = threadIdx.x, wave = tid >> 6;
  const bool lo = wave < 4;
  const bool do_mfma = (lo && (roles & 1)) || (!lo && (roles & 4));
  const bool do_valu = !lo && (roles & 2);
  float s = 0.f;
  __syncthreads();
  const unsigned long long c0 = t_cyc(), r0 = t_real();
  if (do_mfma) s += role_mfma<KIND, NACC, KV>(n_mfma, tid);
  if (do_valu) s += role_valu<NOPS>(n_valu, tid);
  const unsigned long long c1 = t_cyc(), r1 = t_real();
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) { rec[(size_t)blockIdx.x * 8 + wave].cyc = c1 - c0; rec[(size_t)blockIdx.x * 8 + wave].rt = r1 - r0; }
}

struct Res { double cyc_lo, cyc_hi, ghz, ms; };
typedef void (*kern_t)(float*, Rec*, int, int, int);

static float* d_out; static Rec* d_rec; static hipEvent_t e0, e1;

static Res run(kern_t fn, int grid, int threads, int n_mfma, int n_valu, int roles) {
  std::vector<Rec> h((size_t)grid * 8);
  hipMemset(d_rec, 0, sizeof(Rec) * h.size());
  hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), 0, 0, d_out, d_rec, n_mfma, n_valu, roles);   // warm
  hipDeviceSynchronize();
  hipMemset(d_rec, 0, sizeof(Rec) * h.size());
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), 0, 0, d_out, d_rec, n_mfma, n_valu, roles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d_rec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
  double cl = 0, ch = 0, ghz = 0; int nl = 0, nh = 0, ng = 0;
  const int waves = threads / 64;
  for (int b = 0; b < grid; ++b)
    for (int w = 0; w < waves; ++w) {
      const Rec& r = h[(size_t)b * 8 + w];
      const bool active = (w < 4) ? (roles & 1) : (roles & 6);
      if (!active) continue;
      if (w < 4) { cl += r.cyc; ++nl; } else { ch += r.cyc; ++nh; }
      if (r.rt) { ghz += (double)r.cyc / ((double)r.rt * 10.0); ++ng; }      // s_memrealtime ticks at 100 MHz = 10 ns
    }
  return Res{nl ? cl / nl : 0, nh ? ch / nh : 0, ng ? ghz / ng : 0, ms};
}

template <int KIND, int NACC, int NOPS>
static void part_a(const char* name, int grid) {
  kern_t fn = k<KIND, NACC, 0, NOPS>;
  // calibrate: cycles per iteration of each role alone, then ~1 M cycles each
  Res m0 = run(fn, grid, 512, 2000, 0, 1), v0 = run(fn, grid, 512, 0, 2000, 2);
  const int n_mfma = (int)(1.0e6 / (m0.cyc_lo / 2000.0)), n_valu = (int)(1.0e6 / (v0.cyc_hi / 2000.0));
  Res m = run(fn, grid, 512, n_mfma, 0, 1), v = run(fn, grid, 512, 0, n_valu, 2), t = run(fn, grid, 512, n_mfma, n_valu, 3);
  const double cpm = m.cyc_lo / ((double)n_mfma * NACC), cpv = v.cyc_hi / ((double)n_valu * 16);
  printf("A grid=%-4d %-5s acc=%d valu=%3d%%  alone: mfma %8.0f cyc (%5.1f/mfma, %.2f GHz, %.3f ms)  valu %8.0f cyc (%4.2f/fma, %.2f GHz, %.3f ms)"
         "  | together: mfma %8.0f (x%.2f)  valu %8.0f (x%.2f)  %.2f GHz  %.3f ms  (sum of alone %.3f, max %.3f)\n",
         grid, name, NACC, NOPS == 0 ? 100 : NOPS == 1 ? 50 : 25, m.cyc_lo, cpm, m.ghz, m.ms, v.cyc_hi, cpv, v.ghz, v.ms,
         t.cyc_lo, t.cyc_lo / m.cyc_lo, t.cyc_hi, t.cyc_hi / v.cyc_hi, t.ghz, t.ms, m.ms + v.ms, m.ms > v.ms ? m.ms : v.ms);
}

template <int KIND, int NACC, int KV>
static void part_b(const char* name, int grid) {
  kern_t fn = k<KIND, NACC, KV, 0>;
  const int n = 6000;
  Res one = run(fn, grid, 256, n, 0, 1), two = run(fn, grid, 512, n, 0, 5);
  printf("B grid=%-4d %-5s acc=%d  %2d v_fma per MFMA  1 wave/SIMD: %6.1f cyc/mfma %.2f GHz %.3f ms   2 waves/SIMD: %6.1f cyc/mfma per wave"
         " (%5.1f per SIMD-mfma) %.2f GHz %.3f ms\n",
         grid, name, NACC, KV, one.cyc_lo / ((double)n * NACC), one.ghz, one.ms, 0.5 * (two.cyc_lo + two.cyc_hi) / ((double)n * NACC),
         0.25 * (two.cyc_lo + two.cyc_hi) / ((double)n * NACC), two.ghz, two.ms);
}

int main() {
  hipMalloc(&d_out, (size_t)1024 * 512 * 4);
  hipMalloc(&d_rec, sizeof(Rec) * 1024 * 8);
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# cycles = s_memtime per wavefront (mean over the wavefronts of a role); GHz = cycles / s_memrealtime (100 MHz);\n"
         "# 'x' = cycles together / cycles alone for the same wavefronts (1.00 = perfect overlap; mfma x + valu x ~ 2 means the two serialise)\n");
  const int grids[] = {1, 256};
  for (int grid : grids) {
    part_a<1, 1, 0>("f32", grid); part_a<1, 2, 0>("f32", grid); part_a<1, 4, 0>("f32", grid);
    part_a<1, 2, 1>("f32", grid); part_a<1, 2, 3>("f32", grid);
    part_a<2, 1, 0>("bf16", grid); part_a<2, 2, 0>("bf16", grid); part_a<2, 4, 0>("bf16", grid);
    part_a<2, 4, 1>("bf16", grid); part_a<2, 4, 3>("bf16", grid);
  }
  for (int grid : grids) {
    part_b<1, 2, 0>("f32", grid); part_b<1, 2, 2>("f32", grid); part_b<1, 2, 4>("f32", grid); part_b<1, 2, 8>("f32", grid);
    part_b<1, 2, 12>("f32", grid); part_b<1, 2, 16>("f32", grid);
    part_b<2, 4, 0>("bf16", grid); part_b<2, 4, 2>("bf16", grid); part_b<2, 4, 4>("bf16", grid); part_b<2, 4, 6>("bf16", grid);
    part_b<2, 4, 8>("bf16", grid);
  }
  // the whole chip with FOUR workgroups per CU is not needed: 512-thread workgroups at one per CU fill every SIMD twice
  return 0;
}
