// hipemu -- a tiny CPU stand-in for the slice of the HIP runtime the rcmarl
// kernels use.  TEST INFRASTRUCTURE ONLY: it lets the *same* kernel sources be
// compiled with g++ and executed on the CPU (one workgroup at a time, one
// ucontext fiber per work-item, 64-lane wavefronts, emulated f32 MFMA) so
// indexing / LDS / barrier logic is exercised by the `-m "not gpu"` tests in a
// container without a GPU.  It is never linked into the product library and
// never loaded by the product package.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace hipemu {
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  bool done = false;
  unsigned long long wait_gen = 0;   // barrier generation the fiber waits to pass
  int wait_kind = 0;                 // 0 none, 1 block barrier, 2 wave barrier
};
struct State {
  dim3 grid, block, bidx, tidx;
  int cur = 0, nthreads = 0;
  std::vector<Fiber> fibers;
  ucontext_t sched;
  std::function<void()> body;
  // barriers
  int blk_arrived = 0; unsigned long long blk_gen = 0;
  std::vector<int> wave_arrived; std::vector<unsigned long long> wave_gen;
  // wave exchange scratch: [wave][lane][slot]
  std::vector<float> xch_f; std::vector<unsigned long long> xch_u;
  std::vector<unsigned> xch_m;        // [wave][lane][8 dwords]: operands of the bf16 MFMA emulation
  std::vector<float> xch_q;           // [wave][lane][2]: A operands of the 4x4x1 MFMA, two alternating slots
  std::vector<unsigned char> par_q;   // [wave][lane]: slot the lane's next 4x4x1 MFMA uses
  std::vector<char> dyn_smem;
};
State& st();
void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body);
void block_barrier();
void wave_barrier();
inline int lane() { return st().cur & 63; }
inline int wave() { return st().cur >> 6; }
inline int wave_size_here() {   // lanes present in this (possibly partial) wave
  int base = (st().cur >> 6) << 6;
  int n = st().nthreads - base;
  return n > 64 ? 64 : n;
}
}  // namespace hipemu

#define threadIdx (hipemu::st().tidx)
#define blockIdx (hipemu::st().bidx)
#define blockDim (hipemu::st().block)
#define gridDim (hipemu::st().grid)
#define warpSize 64

inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}

// ---- cross-lane ----------------------------------------------------------------
template <typename T> inline T __hipemu_xch(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "xch");
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  unsigned long long bits = 0; memcpy(&bits, &v, sizeof(T));
  s.xch_u[(size_t)w * 64 + l] = bits;
  hipemu::wave_barrier();
  unsigned long long got = s.xch_u[(size_t)w * 64 + (src_lane & 63)];
  hipemu::wave_barrier();
  T r; memcpy(&r, &got, sizeof(T));
  return r;
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return __hipemu_xch(v, hipemu::lane() ^ mask); }
template <typename T> inline T __shfl(T v, int src, int width = 64) { (void)width; return __hipemu_xch(v, src); }
template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
  (void)width; int l = hipemu::lane(); int src = l + (int)delta; if (src > 63) src = l; return __hipemu_xch(v, src);
}
inline void __hipemu_gather64(float v, float (&out)[64]) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  s.xch_f[((size_t)w * 64 + l) * 2] = v;
  hipemu::wave_barrier();
  int n = hipemu::wave_size_here();
  for (int i = 0; i < 64; ++i) out[i] = i < n ? s.xch_f[((size_t)w * 64 + i) * 2] : 0.f;
  hipemu::wave_barrier();
}
inline int __builtin_amdgcn_readfirstlane(int v) { return __hipemu_xch(v, 0); }

inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}

// ---- f32 MFMA (layout: cdna_hip_programming.md section 3) -----------------------
struct floatx16 { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct floatx4 { float v[4]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
inline floatx16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, floatx16 c, int, int, int) {
  // lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
  // holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] in register r.
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  s.xch_f[((size_t)w * 64 + l) * 2 + 0] = a;
  s.xch_f[((size_t)w * 64 + l) * 2 + 1] = b;
  hipemu::wave_barrier();
  floatx16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 2; ++k) {
      float av = s.xch_f[((size_t)w * 64 + row + 32 * k) * 2 + 0];
      float bv = s.xch_f[((size_t)w * 64 + col + 32 * k) * 2 + 1];
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  hipemu::wave_barrier();
  return d;
}
// two-block form: block b = lane>>5 on the input side; lane l supplies A_b[i=l&31][k=0], B_b[k=0][j=l&31];
// D_b[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] in register 16*b + r  (r = 0..15) -- every lane holds both blocks.
struct floatx32 { float v[32]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
inline floatx32 __builtin_amdgcn_mfma_f32_32x32x1f32(float a, float b, floatx32 c, int, int, int) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  s.xch_f[((size_t)w * 64 + l) * 2 + 0] = a;
  s.xch_f[((size_t)w * 64 + l) * 2 + 1] = b;
  hipemu::wave_barrier();
  floatx32 d = c;
  int col = l & 31;
  for (int blk = 0; blk < 2; ++blk)
    for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float av = s.xch_f[((size_t)w * 64 + row + 32 * blk) * 2 + 0];
      float bv = s.xch_f[((size_t)w * 64 + col + 32 * blk) * 2 + 1];
      d[16 * blk + r] = fmaf(av, bv, d[16 * blk + r]);
    }
  hipemu::wave_barrier();
  return d;
}
// v_permlane32_swap: lanes 32-63 of `vdst` swap with lanes 0-31 of `src` (the other two halves stay); returns {vdst', src'}
struct __hipemu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline __hipemu_u2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  s.xch_u[(size_t)w * 64 + l] = ((unsigned long long)vdst << 32) | src;
  hipemu::wave_barrier();
  unsigned long long other = s.xch_u[(size_t)w * 64 + (l ^ 32)];
  hipemu::wave_barrier();
  __hipemu_u2 r;
  if (l < 32) { r.v[0] = vdst; r.v[1] = (unsigned)(other >> 32); }           // lower: keeps vdst, src <- upper's vdst
  else { r.v[0] = (unsigned)(other & 0xffffffffu); r.v[1] = src; }              // upper: vdst <- lower's src, keeps src
  return r;
}

inline floatx4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, floatx4 c, int, int, int) {
  // sixteen independent 4x4x1 products: block = l>>2; lane 4*blk+i supplies A[blk][i], lane 4*blk+j supplies B[blk][j];
  // lane l = 4*blk+j holds D[blk][i = register][j]     (checked on gfx950: tools/micro/mfma4x4.hip)
  // ONE barrier per call (kernels issue hundreds of these): the A operands go through two alternating slots of their own --
  // a lane can only rewrite slot p two calls later, i.e. after the barrier of the call in between, which every lane
  // reaches only when it has finished reading slot p.
  auto& s = hipemu::st();
  const int w = hipemu::wave(), l = hipemu::lane();
  const size_t me = (size_t)w * 64 + l;
  const int p = s.par_q[me];
  s.par_q[me] = (unsigned char)(p ^ 1);
  s.xch_q[me * 2 + p] = a;
  hipemu::wave_barrier();
  floatx4 d = c;
  const size_t blk0 = (size_t)w * 64 + 4 * (l >> 2);
  for (int i = 0; i < 4; ++i) d[i] = fmaf(s.xch_q[(blk0 + i) * 2 + p], b, d[i]);
  return d;
}

inline floatx4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, floatx4 c, int, int, int) {
  // lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D[row=(l>>4)*4+r][col=l&15]
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  s.xch_f[((size_t)w * 64 + l) * 2 + 0] = a;
  s.xch_f[((size_t)w * 64 + l) * 2 + 1] = b;
  hipemu::wave_barrier();
  floatx4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = d[r];
    for (int k = 0; k < 4; ++k) {
      float av = s.xch_f[((size_t)w * 64 + row + 16 * k) * 2 + 0];
      float bv = s.xch_f[((size_t)w * 64 + col + 16 * k) * 2 + 1];
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  hipemu::wave_barrier();
  return d;
}

// ---- bf16 MFMA 32x32x16 (8 bf16 per lane per operand, packed in a uint4) ---------
// lane l supplies A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31] (e = 0..7, element e in
// bits 16*(e&1) of dword e>>1); D layout as the f32 forms.  Products of two bf16 are exact in
// fp32; the 16 of them are added to the accumulator in k order.
inline float __hipemu_bf16(unsigned dword, int hi) {
  unsigned bits = (hi ? (dword >> 16) : (dword & 0xffffu)) << 16;
  float f; memcpy(&f, &bits, 4); return f;
}
inline floatx16 __hipemu_mfma_f32_32x32x16_bf16(uint4 a, uint4 b, floatx16 c) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  unsigned* m = &s.xch_m[((size_t)w * 64 + l) * 8];
  m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  hipemu::wave_barrier();
  floatx16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 16; ++k) {
      const unsigned* ma = &s.xch_m[((size_t)w * 64 + row + 32 * (k >> 3)) * 8];
      const unsigned* mb = &s.xch_m[((size_t)w * 64 + col + 32 * (k >> 3)) * 8 + 4];
      int e = k & 7;
      acc += __hipemu_bf16(ma[e >> 1], e & 1) * __hipemu_bf16(mb[e >> 1], e & 1);
    }
    d[r] = acc;
  }
  hipemu::wave_barrier();
  return d;
}
// ---- f16 MFMA 32x32x16: operand layout of the bf16 form; products of two f16 are exact in fp32 ----------
inline float __hipemu_f16(unsigned dword, int hi) {
  unsigned h = hi ? (dword >> 16) : (dword & 0xffffu);
  unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu, bits;
  if (e == 31u) bits = sign | 0x7f800000u | (m << 13);
  else if (e == 0u) { float v = (float)m * 5.9604644775390625e-8f; return sign ? -v : v; }
  else bits = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &bits, 4); return f;
}
inline floatx16 __hipemu_mfma_f32_32x32x16_f16(uint4 a, uint4 b, floatx16 c) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  unsigned* m = &s.xch_m[((size_t)w * 64 + l) * 8];
  m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  hipemu::wave_barrier();
  floatx16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 16; ++k) {
      const unsigned* ma = &s.xch_m[((size_t)w * 64 + row + 32 * (k >> 3)) * 8];
      const unsigned* mb = &s.xch_m[((size_t)w * 64 + col + 32 * (k >> 3)) * 8 + 4];
      int e = k & 7;
      acc += __hipemu_f16(ma[e >> 1], e & 1) * __hipemu_f16(mb[e >> 1], e & 1);
    }
    d[r] = acc;
  }
  hipemu::wave_barrier();
  return d;
}
// ---- int8 MFMA 32x32x32 (16 int8 per lane per operand, packed in a uint4), i32 accumulate: exact ----------
// lane l supplies A[i=l&31][k=16*(l>>5)+e], B[k=16*(l>>5)+e][j=l&31] (e = 0..15, byte e&3 of dword e>>2); D layout as the f32 forms.
struct intx16 { int v[16]; int& operator[](int i) { return v[i]; } const int& operator[](int i) const { return v[i]; } };
inline intx16 __hipemu_mfma_i32_32x32x32_i8(uint4 a, uint4 b, intx16 c) {
  auto& s = hipemu::st();
  int w = hipemu::wave(), l = hipemu::lane();
  unsigned* m = &s.xch_m[((size_t)w * 64 + l) * 8];
  m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  hipemu::wave_barrier();
  intx16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int acc = d[r];
    for (int k = 0; k < 32; ++k) {
      const unsigned* ma = &s.xch_m[((size_t)w * 64 + row + 32 * (k >> 4)) * 8];
      const unsigned* mb = &s.xch_m[((size_t)w * 64 + col + 32 * (k >> 4)) * 8 + 4];
      int e = k & 15;
      acc += (int)(signed char)((ma[e >> 2] >> (8 * (e & 3))) & 0xff) * (int)(signed char)((mb[e >> 2] >> (8 * (e & 3))) & 0xff);
    }
    d[r] = acc;
  }
  hipemu::wave_barrier();
  return d;
}
// global_load_lds_dwordx4: LDS destination = wave-uniform base + lane*16 (the hardware takes the
// base from M0, i.e. from the first lane), global source per lane.
inline void __hipemu_glds16(const void* gsrc, void* lds_base) {
  unsigned long long base0 = __hipemu_xch((unsigned long long)(uintptr_t)lds_base, 0);
  memcpy(reinterpret_cast<char*>((uintptr_t)base0) + 16 * hipemu::lane(), gsrc, 16);
}

// ds_read_b64_tr_b16 (gfx950; semantics probed on the chip: tools/micro/tr_b16_probe.hip, profiles/r03b_ds_read_tr_b16_probe.txt):
// every lane FETCHES 8 bytes (four u16) at its own address; inside each group of 16 lanes the 16 x 4 elements are
// transposed: lane ll (0..15) of a group receives, as element j = 0..3, element (ll & 3) of what lane 4*j + (ll >> 2) fetched.
// (With lane addresses row (ll>>2), column 4*(ll&3) of a [4][16] block that is column ll of the block, rows 0..3.)
struct __hipemu_s4 { short v[4]; short& operator[](int i) { return v[i]; } const short& operator[](int i) const { return v[i]; } };
inline __hipemu_s4 __hipemu_ds_read_tr16_b64(const void* p) {
  unsigned long long mine;
  memcpy(&mine, p, 8);
  const int l = hipemu::lane(), ll = l & 15, base = l & ~15;
  __hipemu_s4 r;
  for (int j = 0; j < 4; ++j) {
    const unsigned long long o = __hipemu_xch(mine, base + 4 * j + (ll >> 2));
    r.v[j] = (short)((o >> (16 * (ll & 3))) & 0xffffull);
  }
  return r;
}

// dynamic LDS
#define HIPEMU_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(hipemu::st().dyn_smem.data())

inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __fdividef(float a, float b) { return a / b; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
