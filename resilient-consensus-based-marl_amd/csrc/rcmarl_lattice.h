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

#ifdef RCMARL_EMU
__device__ __forceinline__ rc_f32x16 rc_mfma_bf16(uint4 a, uint4 b, rc_f32x16 c) {
  return __hipemu_mfma_f32_32x32x16_bf16(a, b, c);
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
