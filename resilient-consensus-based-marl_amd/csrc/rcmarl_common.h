// Shared device/host helpers for the rcmarl HIP kernels (gfx950 / MI355X).
//
// Data layout conventions (see DESIGN.md "Data layout in HBM"):
//   * every network family (actor / critic / team-reward) of every agent of every
//     seed is one row of a stacked parameter matrix  theta[S][N][ldp]  (fp32),
//     row = [W1(in x hid) | b1 | W2(hid x hid) | b2 | W3(hid x out) | b3 | pad]
//     in Keras order (reference main.py:59-82, SURVEY.md 8b "weight exchange
//     format"); ldp = P rounded up to 64 floats so every row is 256-B aligned.
//   * activations of the batched layer-1 GEMM are FEATURE-MAJOR:
//     a1t[S][N*hid][ldb]  (row = one hidden unit of one agent, contiguous over
//     the replay batch index b) so that a wavefront's lanes map to consecutive b.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RCMARL_OK 0
#define RCMARL_ERR_ARG 1
#define RCMARL_ERR_LAUNCH 2
#define RCMARL_ERR_UNSUPPORTED 3
// one job of rcmarl_minibatch_fit_multi.  THREE definitions must agree: this one, include/rcmarl.h and capi.MbJob (ctypes); the
// static_asserts below pin this one, rcmarl_mb_job_layout() (abi.hip) hands the numbers to tests/test_capi_symbols.py, which holds
// the ctypes structure and the public header to them.
typedef struct rcmarl_mb_job {
  const float* x; long x_seed_stride;      /* replay tensor of this network's input family and its seed stride (floats) */
  float* theta;                            /* [S][N][ldp], rows `agents` fitted in place */
  const int* agents; int n_adv;            /* the fitted agents of every seed */
  int in_dim, ldp, reserved_;
  const float* y;                          /* targets [S][N][ldb] */
  const int* perm;                         /* int32 [S][n_adv][epochs][B], or NULL (natural order) */
  float* loss_out;                         /* [S][N] first-epoch loss, or NULL */
  int* ovf_flags;                          /* int32[S * n_adv], zero before first use, one buffer per job and call site */
} rcmarl_mb_job;
#include <stddef.h>
static_assert(sizeof(rcmarl_mb_job) == 80 && offsetof(rcmarl_mb_job, x_seed_stride) == 8 && offsetof(rcmarl_mb_job, theta) == 16 &&
              offsetof(rcmarl_mb_job, agents) == 24 && offsetof(rcmarl_mb_job, n_adv) == 32 && offsetof(rcmarl_mb_job, in_dim) == 36 &&
              offsetof(rcmarl_mb_job, ldp) == 40 && offsetof(rcmarl_mb_job, y) == 48 && offsetof(rcmarl_mb_job, perm) == 56 &&
              offsetof(rcmarl_mb_job, loss_out) == 64 && offsetof(rcmarl_mb_job, ovf_flags) == 72,
              "rcmarl_mb_job: the layout include/rcmarl.h and capi.MbJob declare");

#ifdef RCMARL_EMU
#define RCMARL_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define RCMARL_DYN_SMEM(type, name) HIPEMU_DYN_SMEM(type, name)
#define RCMARL_EXPORT extern "C"
typedef floatx16 rc_f32x16;
typedef floatx4 rc_f32x4;
typedef floatx32 rc_f32x32;
#else
// hipGetLastError() is sticky across the whole runtime (PyTorch's own calls included):
// drop any stale error first so rcmarl_check_launch() reports THIS launch only.
#define RCMARL_LAUNCH(kernel, grid, block, smem, stream, ...) \
  do { (void)hipGetLastError();                              \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__); } while (0)
#define RCMARL_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
  type* name = reinterpret_cast<type*>(name##_raw)
#define RCMARL_EXPORT extern "C" __attribute__((visibility("default")))
typedef float rc_f32x16 __attribute__((ext_vector_type(16)));
typedef float rc_f32x4 __attribute__((ext_vector_type(4)));
typedef float rc_f32x32 __attribute__((ext_vector_type(32)));
#endif

static inline int rcmarl_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RCMARL_OK : RCMARL_ERR_LAUNCH;
}

// Offsets of the six Keras arrays inside one parameter row.
struct NetGeom {
  int in_dim, hid, out_dim;
  int o_b1, o_W2, o_b2, o_W3, o_b3, P, P_hid;
};
__host__ __device__ static inline NetGeom make_geom(int in_dim, int hid, int out_dim) {
  NetGeom g;
  g.in_dim = in_dim; g.hid = hid; g.out_dim = out_dim;
  g.o_b1 = in_dim * hid;
  g.o_W2 = g.o_b1 + hid;
  g.o_b2 = g.o_W2 + hid * hid;
  g.o_W3 = g.o_b2 + hid;
  g.o_b3 = g.o_W3 + hid * out_dim;
  g.P = g.o_b3 + out_dim;
  g.P_hid = g.o_W3;
  return g;
}

#define RC_LEAK 0.1f
__device__ __forceinline__ float rc_lrelu(float z) { return z > 0.f ? z : RC_LEAK * z; }
// LeakyReLU keeps the sign, so the derivative can be recovered from the activation.
__device__ __forceinline__ float rc_lrelu_grad_from_act(float a) { return a > 0.f ? 1.f : RC_LEAK; }

__device__ __forceinline__ float rc_wave_sum(float v) {
#ifdef RCMARL_EMU
  // same xor-butterfly association order, evaluated locally after ONE lane exchange
  // (keeps the CPU emulation of shuffle-heavy kernels fast)
  float all[64], nxt[64];
  __hipemu_gather64(v, all);
  for (int m = 32; m >= 1; m >>= 1) {
    for (int l = 0; l < 64; ++l) nxt[l] = all[l] + all[l ^ m];
    for (int l = 0; l < 64; ++l) all[l] = nxt[l];
  }
  return all[hipemu::lane()];
#else
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
}

// Wave-wide sum whose result is only needed in ONE lane (lane 63): six v_add_f32 with DPP
// modifiers (quad_perm x2, row_half_mirror, row_mirror, row_bcast15, row_bcast31) instead of
// six ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float rc_wave_sum_lane63(float v) {
#ifdef RCMARL_EMU
  return rc_wave_sum(v);
#else
#define RC_DPP_ADD(ctrl, rmask) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
  RC_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
  RC_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
  RC_DPP_ADD(0x141, 0xf);   // row_half_mirror
  RC_DPP_ADD(0x140, 0xf);   // row_mirror        -> every lane holds its 16-lane row sum
  RC_DPP_ADD(0x142, 0xa);   // row_bcast15 into rows 1,3
  RC_DPP_ADD(0x143, 0xc);   // row_bcast31 into rows 2,3 -> lane 63 holds the wave sum
#undef RC_DPP_ADD
  return v;
#endif
}

// Three independent wave sums at once (results in lane 63 of each): 18 v_add_f32 with the DPP modifier FUSED
// (hipcc lowers the update_dpp form above to v_mov + v_mov_dpp + v_add, three issue slots per step).  The three
// chains are interleaved so that every DPP read is two instructions behind the write it depends on (the
// VALU-write -> DPP-read hazard needs 2 wait states, and nothing pads inside an asm statement); the leading
// s_nop covers the producers of a, b, c.
__device__ __forceinline__ void rc_wave_sum3_lane63(float& a, float& b, float& c) {
#ifdef RCMARL_EMU
  a = rc_wave_sum(a); b = rc_wave_sum(b); c = rc_wave_sum(c);
#else
#define RC_DPP3(mod)                          \
  "v_add_f32_dpp %0, %0, %0 " mod "\n\t"      \
  "v_add_f32_dpp %1, %1, %1 " mod "\n\t"      \
  "v_add_f32_dpp %2, %2, %2 " mod "\n\t"
  asm volatile("s_nop 1\n\t"
               RC_DPP3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               RC_DPP3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_half_mirror row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_mirror row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_bcast:15 row_mask:0xa bank_mask:0xf")
               RC_DPP3("row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c));
#undef RC_DPP3
#endif
}

static inline int rc_ceil_div(int a, int b) { return (a + b - 1) / b; }

// streaming store: data written once and not re-read by this kernel (keeps the operand panels in L2)
#ifdef RCMARL_EMU
#define RC_NT_STORE(ptr, val) (*(ptr) = (val))
__device__ __forceinline__ void rc_nt_store4(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
#else
#define RC_NT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
// four consecutive floats at a 16-byte aligned address: one global_store_dwordx4 nt
__device__ __forceinline__ void rc_nt_store4(float* p, float a, float b, float c, float d) {
  typedef float rc_v4f __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((rc_v4f){a, b, c, d}, reinterpret_cast<rc_v4f*>(p));
}
#endif

// Two fp32 lanes per register pair: v_pk_fma_f32 / v_pk_mul_f32 run 128 FMAs per wavefront instruction in the
// 4-cycle issue slot a plain v_fma_f32 spends on 64 (the 157 TFLOP/s fp32 vector peak is the PACKED rate).
#ifdef RCMARL_EMU
struct rc_f2 { float x, y; };
__device__ __forceinline__ rc_f2 rc_fma2(rc_f2 a, rc_f2 b, rc_f2 c) { return rc_f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
typedef float rc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rc_f2 rc_fma2(rc_f2 a, rc_f2 b, rc_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
__device__ __forceinline__ rc_f2 rc_bcast2(float v) { return rc_f2{v, v}; }
#ifdef RCMARL_EMU
__device__ __forceinline__ rc_f2 rc_add2(rc_f2 a, rc_f2 b) { return rc_f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ rc_f2 rc_sub2(rc_f2 a, rc_f2 b) { return rc_f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ rc_f2 rc_mul2(rc_f2 a, rc_f2 b) { return rc_f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ rc_f2 rc_max2(rc_f2 a, rc_f2 b) { return rc_f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
#else
__device__ __forceinline__ rc_f2 rc_add2(rc_f2 a, rc_f2 b) { return a + b; }
__device__ __forceinline__ rc_f2 rc_sub2(rc_f2 a, rc_f2 b) { return a - b; }
__device__ __forceinline__ rc_f2 rc_mul2(rc_f2 a, rc_f2 b) { return a * b; }
__device__ __forceinline__ rc_f2 rc_max2(rc_f2 a, rc_f2 b) { return __builtin_elementwise_max(a, b); }
#endif
// LeakyReLU of two values at once: max(z, leak*z) == rc_lrelu(z) bit for bit (0 < leak < 1)
__device__ __forceinline__ rc_f2 rc_lrelu2(rc_f2 z) { return rc_max2(z, rc_mul2(z, rc_bcast2(RC_LEAK))); }

// Sums over the lanes of each 32-lane HALF of the wavefront, three values at once; results in lanes 31 and 63
// (five fused-DPP adds per value: the wave-wide rc_wave_sum3_lane63 without its last step).
__device__ __forceinline__ void rc_half_sum3_lane31(float& a, float& b, float& c) {
#ifdef RCMARL_EMU
  float all[64], *vals[3] = {&a, &b, &c};
  for (int q = 0; q < 3; ++q) {
    __hipemu_gather64(*vals[q], all);
    const int base = hipemu::lane() & 32;
    float nxt[32], cur[32];
    for (int l = 0; l < 32; ++l) cur[l] = all[base + l];
    for (int m = 1; m < 32; m <<= 1) {                 // xor butterfly, lowest distance first (as the DPP sequence)
      for (int l = 0; l < 32; ++l) nxt[l] = cur[l] + cur[l ^ m];
      for (int l = 0; l < 32; ++l) cur[l] = nxt[l];
    }
    *vals[q] = cur[hipemu::lane() & 31];
  }
#else
#define RC_DPP3(mod)                          \
  "v_add_f32_dpp %0, %0, %0 " mod "\n\t"      \
  "v_add_f32_dpp %1, %1, %1 " mod "\n\t"      \
  "v_add_f32_dpp %2, %2, %2 " mod "\n\t"
  asm volatile("s_nop 1\n\t"
               RC_DPP3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               RC_DPP3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_half_mirror row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_mirror row_mask:0xf bank_mask:0xf")
               RC_DPP3("row_bcast:15 row_mask:0xa bank_mask:0xf")
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c));
#undef RC_DPP3
#endif
}

// ---- the few places where the host emulation of the test suite (RCMARL_EMU, tests/hipemu) and the gfx950 build differ in
// kernel-independent plumbing live HERE, so that kernel sources carry no build switches for them -------------------------
#ifdef RCMARL_EMU
#define RC_WAIT_VMEM() ((void)0)                      // (the emulated LDS-DMA is a synchronous copy)
#define RC_WAIT_VMEM_N(n) ((void)0)
__device__ __forceinline__ void rc_sleep(int) {}
__device__ __forceinline__ void rc_sleep_short(int) {}
__device__ __forceinline__ void rc_setprio1() {}
__device__ __forceinline__ void rc_setprio(int) {}
#else
// outstanding vector-memory operations of this wavefront (incl. LDS-DMA issued through inline asm, which hipcc does not see)
#define RC_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define RC_WAIT_VMEM_N(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void rc_sleep(int n) {     // n x ~3.4 us (scheduling aid only)
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
__device__ __forceinline__ void rc_sleep_short(int n) {   // n x ~0.2 us (64 x 8 cycles)
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}
__device__ __forceinline__ void rc_setprio1() { __builtin_amdgcn_s_setprio(1); }
#define rc_setprio(n) __builtin_amdgcn_s_setprio(n)
#endif

// host side: opt a kernel into more dynamic LDS than the default limit; number of workgroups of a persistent launch
template <class K>
static inline bool rc_want_lds(K kernel, size_t smem, size_t above = 0) {
#ifndef RCMARL_EMU
  if (smem > above)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
#endif
  return true;
}
static inline int rc_persistent_grid(int resident) {
#ifdef RCMARL_EMU
  return 3;                                           // makes the CPU emulation exercise the strided tile walk
#else
  return resident;
#endif
}

// True if `v` holds for any active lane of the wavefront (one v_cmp + one scalar test).  Used to put a rare slow path
// on a wavefront-uniform branch; both sides of such a branch must give the same result for a lane whose own `v` is false
// (the host emulation decides per lane).
__device__ __forceinline__ bool rc_any(bool v) {
#ifdef RCMARL_EMU
  return v;
#else
  return __builtin_amdgcn_ballot_w64(v) != 0ull;
#endif
}

// 64-bit mask of `v` over the lanes of the wavefront (bit l = lane l; every lane must call it)
__device__ __forceinline__ unsigned long long rc_ballot(bool v) {
#ifdef RCMARL_EMU
  float all[64];
  __hipemu_gather64(v ? 1.f : 0.f, all);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) if (all[l] != 0.f) m |= 1ull << l;
  return m;
#else
  return __builtin_amdgcn_ballot_w64(v);
#endif
}

// True if `v` holds for every active lane (for a fast path without per-lane predicates; the other path must be correct
// for any lane, and the host emulation decides per lane)
__device__ __forceinline__ bool rc_all(bool v) {
#ifdef RCMARL_EMU
  return v;
#else
  return __builtin_amdgcn_ballot_w64(!v) == 0ull;
#endif
}

// Placed first in a rarely taken block: the compiler may not hoist ("speculate") the block's arithmetic above the branch
#ifdef RCMARL_EMU
#define RC_NO_SPECULATE() ((void)0)
#else
#define RC_NO_SPECULATE() asm volatile("" ::: "memory")
#endif

// Scheduling fence: hipcc's scheduler otherwise hoists every independent LDS read of a fully unrolled loop to its top
// (100 ds_read_b128 of weights = 400 live registers -> spills); nothing moves across this point.
#ifdef RCMARL_EMU
#define RC_SCHED_FENCE() ((void)0)
#else
#define RC_SCHED_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// A value the optimiser must treat as unknown: address arithmetic that depends on it stays where it is written (loop-invariant
// code motion otherwise hoists every per-lane address of a big unrolled epilogue out of the enclosing loops and keeps
// hundreds of registers alive across them).
#ifdef RCMARL_EMU
__device__ __forceinline__ int rc_opaque_v(int v) { return v; }
__device__ __forceinline__ int rc_opaque_s(int v) { return v; }
__device__ __forceinline__ void rc_touch(float&) {}
__device__ __forceinline__ void rc_swap_halves(unsigned& a, unsigned& b) {      // (every lane of the wavefront must call it)
  const unsigned pa = __shfl_xor(a, 32, 64), pb = __shfl_xor(b, 32, 64);
  if (threadIdx.x & 32) a = pb; else b = pa;
}
#else
__device__ __forceinline__ int rc_opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int rc_opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
// Exchange between the two lanes of a pair (lane, lane ^ 32): lanes 0-31 receive the partner's `a` in `b`, lanes 32-63 the partner's
// `b` in `a` (v_permlane32_swap: lanes 32-63 of its first operand <-> lanes 0-31 of its second; one instruction, no LDS).
__device__ __forceinline__ void rc_swap_halves(unsigned& a, unsigned& b) {
  const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = sw[0];
  b = sw[1];
}
// a use of a loaded value that costs nothing: pins the compiler's s_waitcnt for it to this point of the program
__device__ __forceinline__ void rc_touch(float& v) { asm volatile("" : "+v"(v)); }
#endif

// v_permlane32_swap on two floats: lanes 32-63 of `a` swap with lanes 0-31 of `b`
__device__ __forceinline__ void rc_swap32(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// Ordering point for data exchanged through LDS among the lanes of ONE wavefront (no workgroup barrier): the LDS
// executes a wavefront's instructions in order, so this only has to stop the compiler from moving accesses across it.
#ifdef RCMARL_EMU
#define RC_WAVE_SYNC() hipemu::wave_barrier()
#else
#define RC_WAVE_SYNC()                                         \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)
#endif
