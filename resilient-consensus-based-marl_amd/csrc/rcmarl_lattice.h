// Lattice ("exact bf16x3") form of the two layer-1 GEMMs -- shared definitions.
//
// Why.  The network input of every agent is the global grid-world state (reference
// environments/grid_world.py:66-72): column c of a replay row is  x = (pos - mean_c)/std_c  with
// integer pos, mean_c = (n-1)/2, or a raw action index 0..4 (training/train_agents.py:91).  So
//     x[b][c] = alpha_c * K[b][c],   K[b][c] = 2*pos - (n-1)  (or the action)  -- a SMALL INTEGER,
// exactly representable in bf16 (|K| <= 256).  A fp32 weight w splits EXACTLY into three bf16
// pieces w = h + m + l (8+8+8 significand bits, round-to-nearest residuals), and every product
// K*piece has <= 16 significant bits, i.e. is exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  Hence
//     sum_c x_c w_c  =  sum_c K_c * fl(alpha_c w_c)            (forward,  3 bf16 MFMA passes)
//     sum_b x_bc dz_b = alpha_c * sum_b K_bc * (h+m+l)(dz_b)   (backward, 3 bf16 MFMA passes)
// carry the same fp32 accumulation error class as an fmaf chain (one extra rounding of
// alpha_c*w), at 16/3 = 5.3x the f32-MFMA issue rate.  rcmarl_lattice_encode VERIFIES the lattice
// property of the actual replay tensor (sets a flag otherwise; the engine then stays on the f32
// MFMA path of layer1_gemm.hip).
//
// Packed operand format ("PK"): a matrix Q[R rows][K reduction] of NP bf16 pieces is stored as
// 8-KiB blocks  [R/128][KT][NP][128 rows][32 k]  (KT = allocated k-tiles), and inside a block the
// four 16-byte chunks of a 64-byte row are XOR-swizzled:
//     byte(r, k, p) = (((r>>7)*KT + (k>>5))*NP + p)*8192 + (r&127)*64 + ((((k&31)>>3) ^ ((r>>2)&3))<<4) + (k&7)*2
// so that (1) one (row-tile, k-tile) of all pieces is ONE contiguous run -> a stage of the GEMM is
// filled by 1-KiB global_load_lds_dwordx4 bursts with a linear LDS image, and (2) the MFMA fragment
// read  ds_read_b128(row = lane&31, chunk = 2*kstep + lane>>5)  is bank-conflict free.
#pragma once
#include "rcmarl_common.h"
#include <stdlib.h>

#define RC_PK_BLOCK 8192

__host__ __device__ static inline long rc_pk_offset(int r, int k, int p, int KT, int NP) {
  return (((long)(r >> 7) * KT + (k >> 5)) * NP + p) * RC_PK_BLOCK + (r & 127) * 64 +
         ((((k & 31) >> 3) ^ ((r >> 2) & 3)) << 4) + (k & 7) * 2;
}

// fp32 -> bf16 bits, round to nearest even (finite inputs)
__device__ __forceinline__ unsigned rc_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float rc_bf16_to_f32(unsigned h) { return __uint_as_float(h << 16); }

// w == h + m + l exactly (each residual is exactly representable, |w| < 2^127, no fp32 underflow)
__device__ __forceinline__ void rc_split3(float w, unsigned& h, unsigned& m, unsigned& l) {
  h = rc_bf16_rne(w);
  const float r1 = w - rc_bf16_to_f32(h);
  m = rc_bf16_rne(r1);
  const float r2 = r1 - rc_bf16_to_f32(m);
  l = rc_bf16_rne(r2);
}

// the same split for two values at once; piece of w0 in bits 0-15, of w1 in bits 16-31 of each output
// (device: v_cvt_pk_bf16_f32 rounds both to nearest even in one instruction, v_pk_add_f32 forms both residuals)
#ifdef RCMARL_EMU
__device__ __forceinline__ void rc_split3_pair(float w0, float w1, unsigned& h, unsigned& m, unsigned& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  rc_split3(w0, h0, m0, l0);
  rc_split3(w1, h1, m1, l1);
  h = h0 | (h1 << 16); m = m0 | (m1 << 16); l = l0 | (l1 << 16);
}
#else
typedef __bf16 rc_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rc_split3_pair(float w0, float w1, unsigned& h, unsigned& m, unsigned& l) {
  rc_f2 v = {w0, w1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rc_bf16x2));
  const rc_f2 r1 = v - rc_f2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, rc_bf16x2));
  const rc_f2 r2 = r1 - rc_f2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, rc_bf16x2));
}
#endif

// ---------------------------------------------------------------------------------------------
// Two-piece f16 form ("f16x2", round 3).  A fp32 value v, scaled by a fixed power of two, is carried as h = rn_f16(v),
// l = rn_f16(v - h): 11 + 11 significand bits and the sign of the residual, i.e. v up to ONE unit in the last place of its
// fp32 significand (exact for 3 values in 4; the rest are off by exactly one fp32 ulp), or up to 2^-25 of the scaled unit
// where the residual falls into f16's subnormals.  Products K * piece (|K| <= 256) have <= 20 significant bits: exact in
// the fp32 accumulator of v_mfma_f32_32x32x16_f16.  Two matrix passes and 4 bytes per value instead of three and 6 -- the
// GEMMs' matrix time AND their LDS traffic, which bound them.  f16 has 5 exponent bits, hence the
// fixed scales:  forward  W'' = 2^10 * alpha_k * W1  (finite while |alpha_k W1| < 64, full precision above 2.4e-4,
// absolute 3e-11 below), backward dz'' = 2^8 * dz1 (finite while |dz1| < 256; absolute 1.2e-10, far below what lr * dz
// contributes to one ulp of a weight); the epilogues multiply the scale out (exact).  RCMARL_LAT_F16: bit 0 = forward
// operand, bit 1 = backward operand in this form; 0 = the exact three-piece bf16 form everywhere.
#define RC_F16_W_SCALE 1024.f
#define RC_F16_W_UNSCALE 0.0009765625f
#ifndef RC_F16_DZ_SCALE                 // (A/B builds: -DRC_F16_DZ_SCALE=65536.f -DRC_F16_DZ_UNSCALE=1.52587890625e-05f in every source)
#define RC_F16_DZ_SCALE 256.f
#define RC_F16_DZ_UNSCALE 0.00390625f
#endif
// activations of a wide network as two f16 pieces (wide_kernels.hip, dense_pk.hip): full precision from |a| >= 2.0e-3, finite to 1015
#define RC_F16_ACT_SCALE 64.f
#define RC_F16_ACT_UNSCALE 0.015625f
#ifndef RC_LAT_F16_DEFAULT
#define RC_LAT_F16_DEFAULT 3
#endif
// (csrc/abi.hip) the operand form -- RCMARL_LAT_F16 read once, then rcmarl_lattice_set_f16_mode -- and the form each packed
// buffer was last WRITTEN in: producers record it, consumers refuse a buffer written in the other form (RCMARL_ERR_ARG)
int rc_lat_f16_mode();
void rc_form_set(const void* buf, int f16_form);
bool rc_form_ok(const void* buf, int f16_form);

// Saturation.  Kernels that form f16 pieces call rc_f16_saturate() first: MODE.FP16_OVFL (bit 23) makes a f32 -> f16 conversion
// of a FINITE value beyond +-65504 return +-65504 instead of infinity (true infinities and NaNs pass).  A value beyond the form's
// range is then carried as h + l = +-131008 / scale at most (|alpha W1| <= 127.9, |dz1| <= 511.8) -- a clipped but FINITE operand:
// a fit that has blown up (the reference's fast_lr leaves a few of 4096 TR nets at 1e8 in the BASELINE configs[3] workload, finite
// in fp32) keeps producing finite numbers that the trimmed mean of its neighbours discards, instead of NaNs that spread.
__device__ __forceinline__ void rc_f16_saturate() {
#ifndef RCMARL_EMU
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1u);          // hwreg(HW_REG_MODE, offset 23, size 1) = 1
#endif
}

// fp32 -> f16 bits, round to nearest even, subnormals kept, finite overflow saturates (the emulation's twin of FP16_OVFL = 1)
__device__ __forceinline__ unsigned rc_f16_rne(float f) {
  unsigned x = __float_as_uint(f);
  const unsigned sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u);
  if (x >= 0x38800000u) {                            // >= 2^-14: a normal f16 (or overflow)
    x += 0x00000fffu + ((x >> 13) & 1u);
    x -= 0x38000000u;
    const unsigned h = x >> 13;
    return sign | (h >= 0x7c00u ? 0x7bffu : h);
  }
  return sign | (unsigned)rintf(__uint_as_float(x) * 16777216.f);     // multiples of 2^-24 (0x400 = 2^-14 falls out right)
}
__device__ __forceinline__ float rc_f16_to_f32(unsigned h) {
  const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  if (e == 31u) return __uint_as_float(sign | 0x7f800000u | (m << 13));
  if (e == 0u) { const float v = (float)m * 5.9604644775390625e-8f; return sign ? -v : v; }
  return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
// two values at once: h, l pieces of (w0, w1) -- already scaled by the caller -- packed low/high as rc_split3_pair does
#ifdef RCMARL_EMU
__device__ __forceinline__ void rc_split2h_pair(float w0, float w1, unsigned& h, unsigned& l) {
  const unsigned h0 = rc_f16_rne(w0), h1 = rc_f16_rne(w1);
  const unsigned l0 = rc_f16_rne(w0 - rc_f16_to_f32(h0)), l1 = rc_f16_rne(w1 - rc_f16_to_f32(h1));
  h = h0 | (h1 << 16); l = l0 | (l1 << 16);
}
__device__ __forceinline__ void rc_split2h_x8(const float (&x)[8], uint4& h, uint4& l) {
  rc_split2h_pair(x[0], x[1], h.x, l.x);
  rc_split2h_pair(x[2], x[3], h.y, l.y);
  rc_split2h_pair(x[4], x[5], h.z, l.z);
  rc_split2h_pair(x[6], x[7], h.w, l.w);
}
#else
typedef _Float16 rc_h2 __attribute__((ext_vector_type(2)));
// Three instructions per pair: one packed conversion for h, then l = rn_f16(v - h) as ONE mixed-precision FMA per value
// (v_fma_mix{lo,hi}_f16: -h read as f16, times 1.0, plus v in fp32; v - h is exact in fp32, so the only rounding is the one to
// f16) -- the plain form (convert h back, packed subtract, packed convert) is five, and hipcc folds any C++ spelling of the FMA
// back into it.  Hence inline asm, which the compiler's hazard recognizer does not look into: a write to the HIGH half of a
// register (v_fma_mixhi) must be one wait state away from a vector instruction that reads the register, and two from an MFMA
// -- the s_nop closing each block (without it the fused prototypes read stale pieces: tools/prototypes/fused_fit.hip, round 4).
#define RC_MIXLO(d, h, v) "v_fma_mixlo_f16 " d ", -" h ", 1.0, " v " op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
#define RC_MIXHI(d, h, v) "v_fma_mixhi_f16 " d ", -" h ", 1.0, " v " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
__device__ __forceinline__ void rc_split2h_pair(float w0, float w1, unsigned& h, unsigned& l) {
  const rc_f2 v = {w0, w1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rc_h2));
  asm(RC_MIXLO("%0", "%1", "%2") RC_MIXHI("%0", "%1", "%3") "s_nop 1" : "=&v"(l) : "v"(h), "v"(w0), "v"(w1));
}
// eight values: the pieces of (x0, x1), (x2, x3), ... in the four words of h and l; one closing s_nop for all
__device__ __forceinline__ void rc_split2h_x8(const float (&x)[8], uint4& h, uint4& l) {
  h.x = __builtin_bit_cast(unsigned, __builtin_convertvector((rc_f2{x[0], x[1]}), rc_h2));
  h.y = __builtin_bit_cast(unsigned, __builtin_convertvector((rc_f2{x[2], x[3]}), rc_h2));
  h.z = __builtin_bit_cast(unsigned, __builtin_convertvector((rc_f2{x[4], x[5]}), rc_h2));
  h.w = __builtin_bit_cast(unsigned, __builtin_convertvector((rc_f2{x[6], x[7]}), rc_h2));
  asm(RC_MIXLO("%0", "%4", "%8") RC_MIXHI("%0", "%4", "%9") RC_MIXLO("%1", "%5", "%10") RC_MIXHI("%1", "%5", "%11")
      RC_MIXLO("%2", "%6", "%12") RC_MIXHI("%2", "%6", "%13") RC_MIXLO("%3", "%7", "%14") RC_MIXHI("%3", "%7", "%15") "s_nop 1"
      : "=&v"(l.x), "=&v"(l.y), "=&v"(l.z), "=&v"(l.w)
      : "v"(h.x), "v"(h.y), "v"(h.z), "v"(h.w), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}
#undef RC_MIXLO
#undef RC_MIXHI
#endif
// acc = max(acc, |a|, |b|) in one instruction (hipcc spends three on fmaxf(acc, fmaxf(fabsf(a), fabsf(b))))
__device__ __forceinline__ float rc_amax3(float acc, float a, float b) {
#ifdef RCMARL_EMU
  return fmaxf(acc, fmaxf(fabsf(a), fabsf(b)));
#else
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
  return r;
#endif
}

#ifdef RCMARL_EMU
__device__ __forceinline__ rc_f32x16 rc_mfma_bf16(uint4 a, uint4 b, rc_f32x16 c) {
  return __hipemu_mfma_f32_32x32x16_bf16(a, b, c);
}
__device__ __forceinline__ rc_f32x16 rc_mfma_f16(uint4 a, uint4 b, rc_f32x16 c) {
  return __hipemu_mfma_f32_32x32x16_f16(a, b, c);
}
#define RC_GLDS16(gsrc, lds_base) __hipemu_glds16((gsrc), (lds_base))
#define RC_GLDS16S(sbase, voff, lds_base) __hipemu_glds16((sbase) + (voff), (lds_base))
typedef unsigned char* rc_lds_t;                               // an LDS location handed to RC_GLDS16S
__device__ __forceinline__ rc_lds_t rc_lds_addr(unsigned char* p) { return p; }
#else
typedef __bf16 rc_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ rc_f32x16 rc_mfma_bf16(uint4 a, uint4 b, rc_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rc_bf16x8, a), __builtin_bit_cast(rc_bf16x8, b), c,
                                                 0, 0, 0);
}
typedef _Float16 rc_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ rc_f32x16 rc_mfma_f16(uint4 a, uint4 b, rc_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rc_f16x8, a), __builtin_bit_cast(rc_f16x8, b), c, 0, 0, 0);
}
// global_load_lds_dwordx4: LDS destination = wave-uniform base (M0) + lane*16, global source per lane.
// Issued through inline asm ON PURPOSE: with the builtin, hipcc cannot tell the LDS-DMA destination
// (the other stage) from the fragment reads of the current stage and drains vmcnt(0) before the first
// ds_read of every k-tile, serialising load and compute.  The asm form is invisible to that pass; the
// k-loop orders the DMA explicitly (RC_WAIT_VMEM + barrier before a stage is read or refilled).
__device__ __forceinline__ void rc_glds16(const void* gsrc, const void* lds_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(lds_base));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
}
#define RC_GLDS16(gsrc, lds_base) rc_glds16((gsrc), (lds_base))
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the per-k-tile address
// update is two SALU adds instead of 64-bit VALU arithmetic in every lane
typedef unsigned rc_lds_t;                                     // LDS byte address (M0 value), plain integer arithmetic
__device__ __forceinline__ rc_lds_t rc_lds_addr(unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(p);
}
__device__ __forceinline__ void rc_glds16s(const unsigned char* sbase, unsigned voff, rc_lds_t lds_addr) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v)
               : "memory", "m0");
}
#define RC_GLDS16S(sbase, voff, lds_base) rc_glds16s((sbase), (voff), (lds_base))
#endif

// ds_read_b64_tr_b16: LDS transpose read for 16-bit elements.  Every lane supplies the address of 8 bytes; within each group
// of 16 lanes the fetched [16 lanes][4 elements] are transposed (lane ll gets element (ll&3) of lanes (ll>>2), 4+(ll>>2),
// 8+(ll>>2), 12+(ll>>2)).  With lane ll fetching row (ll>>2), columns 4*(ll&3).. of a row-major [4][16] block, lane ll ends up
// with COLUMN ll of the block (rows 0..3): two reads give a lane 8 consecutive rows of its column = one MFMA operand.
// Returns the four 16-bit elements as two packed dwords (element 0 in the low half of .x).
__device__ __forceinline__ uint2 rc_lds_read_tr16(const unsigned short* p) {
#ifdef RCMARL_EMU
  const __hipemu_s4 v = __hipemu_ds_read_tr16_b64(p);
  uint2 r;
  r.x = (unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
  r.y = (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
  return r;
#else
  typedef short rc_s4 __attribute__((ext_vector_type(4)));
  const rc_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rc_s4*)(p));
  return __builtin_bit_cast(uint2, v);
#endif
}

// ---------------------------------------------------------------------------------------------
// The "ten units per lane" form of a 20-unit layer on the 32x32x16 matrix core, shared by k_mid_fit_v8 (mid_kernels.hip) and the
// adversaries' mini-batch fit (minibatch_fit.hip); the layout argument is in k_mid_fit_v8's header.  Lane (row j = lane&31,
// half h = lane>>5) holds local units u = 0..9 of its row = global units v8_unit(h, u).
__device__ __forceinline__ int v8_unit(int h, int u) { return u < 8 ? u + 8 * h : 16 + 2 * h + (u - 8); }
// unit of accumulator / A-operand row i (0..31), or -1 (padding row)
__device__ __forceinline__ int v8_row_unit(int i) {
  const int h = (i >> 2) & 1, q = i >> 3, e = i & 3;
  if (q < 2) return v8_unit(h, 4 * q + e);
  return (q == 2 && e < 2) ? v8_unit(h, 8 + e) : -1;
}
// unit of contraction slot k (0..31), or -1
__device__ __forceinline__ int v8_slot_unit(int k) {
  if (k < 16) return v8_unit(k >> 3, k & 7);
  const int h = (k - 16) >> 3, u = 8 + ((k - 16) & 7);
  return u < 10 ? v8_unit(h, u) : -1;
}

struct V8Pieces { uint4 h, l; };
// RC_V8_DROP_LL=1 (set by mid_kernels.hip for its kernels): the low x low pass left out.
#ifndef RC_V8_DROP_LL
#define RC_V8_DROP_LL 0
#endif
__device__ __forceinline__ rc_f32x16 v8_mfma4(const V8Pieces& a, const V8Pieces& b, rc_f32x16 c) {
  if (!RC_V8_DROP_LL) c = rc_mfma_f16(a.l, b.l, c);
  c = rc_mfma_f16(a.l, b.h, c);
  c = rc_mfma_f16(a.h, b.l, c);
  c = rc_mfma_f16(a.h, b.h, c);
  return c;
}
template <bool SCALED>
__device__ __forceinline__ V8Pieces v8_split8(const float (&x)[8], float sc) {
  V8Pieces p;
  if (SCALED) {
    const float xs[8] = {x[0] * sc, x[1] * sc, x[2] * sc, x[3] * sc, x[4] * sc, x[5] * sc, x[6] * sc, x[7] * sc};
    rc_split2h_x8(xs, p.h, p.l);
  } else {
    rc_split2h_x8(x, p.h, p.l);
  }
  return p;
}
