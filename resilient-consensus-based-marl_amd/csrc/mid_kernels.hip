// K4/K5 (layers 2-3), K2+K3, K6, K7 -- everything between the two layer-1 GEMMs.
//
// After rcmarl_layer1_forward the first-layer activations of every agent are
// FEATURE-MAJOR: a1t[S][N*HID][ldb].  In all kernels below one lane owns one
// replay row b of one (seed, agent); a workgroup = 256 consecutive rows, so the
// agent's weights W2,b2,W3,b3 are workgroup-uniform (scalar loads, SGPR
// operands) and every a1t access is a fully coalesced 256-B wave load.
//
//   k_mid_fit         one full-batch SGD step of critic.fit / TR.fit, layers 2-3
//                     forward + MSE + backward      (agents/resilient_CAC_agents.py:118,136)
//   k_mid_value       V(s) / r(s,a) forward, optional TD target r + gamma*V(ns)   (:114-115)
//   k_consensus_head  estimate consensus + projection residual                    (:168-206, :60-84)
//   k_mid_actor       softmax / TD-weighted sparse CE forward+backward            (:86-101)
//   k_small_sgd / k_small_adam / k_head_apply   reduce partials, apply updates
// Round 6 (visit l): the 20-unit layers of this file run THREE of the four piece products (h x h, h x l, l x h); the low x low one is
// at most 2^-22 of a term -- a quarter of an fp32 rounding step of it -- and leaving it out moves k_mid_fit_v8's gradient records by
// 0.7-1.5e-7 of their scale, LESS than the fp32 vector-ALU form of the same kernel differs from the four-pass form through its
// summation order alone (2.2e-7; profiles/r06l_mid_ab_drop_ll.txt), for 5 % of the kernel's time (434-440 -> 410-419 us).
// -DRC_V8_DROP_LL=0 restores the fourth pass; the adversaries' mini-batch chain (minibatch_fit.hip) keeps all four.
#ifndef RC_V8_DROP_LL
#define RC_V8_DROP_LL 1
#endif
#include "rcmarl_lattice.h"
// A/B switch: raise the wavefront's issue priority around its matrix-core groups (so that a group is issued as early as possible and the
// other wavefronts' vector work runs under it).
#ifndef RC_MID_SETPRIO
#define RC_MID_SETPRIO 0
#endif
#define RC_MX_BEGIN() do { if (RC_MID_SETPRIO) rc_setprio(RC_MID_SETPRIO); } while (0)
#define RC_MX_END() do { if (RC_MID_SETPRIO) rc_setprio(0); } while (0)
#include <stdlib.h>
#include "selnet_generated.inc"

namespace {

constexpr int ROWS = 256;          // replay rows per workgroup
constexpr int LDR = ROWS + 1;      // LDS row stride: (k*LDR + r) % 32 distinct over k

template <int HID> struct FitPart {  // layout of one partial-gradient record
  static constexpr int gW2 = 0, gb2 = HID * HID, gW3 = gb2 + HID, gb3 = gW3 + HID, gb1 = gb3 + 1, loss = gb1 + HID,
                       SIZE = loss + 1, NSMALL = SIZE - gb2;
};
template <int HID, int A> struct ActorPart {
  static constexpr int gW2 = 0, gW3 = HID * HID, gb2 = gW3 + HID * A, gb3 = gb2 + HID, gb1 = gb3 + A,
                       loss = gb1 + HID, SIZE = loss + 1, NSMALL = SIZE - gb2;
};

template <int HID>
__device__ __forceinline__ void load_a1(const float* __restrict__ a1t, long row0, int ldb, int b, bool valid,
                                        float (&a1)[HID]) {
#pragma unroll
  for (int j = 0; j < HID; ++j) a1[j] = valid ? a1t[(row0 + j) * ldb + b] : 0.f;
}

// z2 = a1 @ W2 + b2 ; a2 = lrelu(z2)
template <int HID>
__device__ __forceinline__ void layer2(const float* __restrict__ th, const NetGeom& g, const float (&a1)[HID],
                                       float (&a2)[HID]) {
  // two output units per instruction (v_pk_fma_f32; the weight pair is two consecutive floats of a W2 row, a scalar
  // operand): every unit keeps its own fmaf chain over j, so the values are those of the scalar form bit for bit
  static_assert(HID % 2 == 0, "units are processed in pairs");
  rc_f2 acc[HID / 2];
#pragma unroll
  for (int q = 0; q < HID / 2; ++q) acc[q] = rc_bcast2(0.f);
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    const rc_f2 aj = rc_bcast2(a1[j]);
#pragma unroll
    for (int q = 0; q < HID / 2; ++q)
      acc[q] = rc_fma2(aj, rc_f2{th[g.o_W2 + j * HID + 2 * q], th[g.o_W2 + j * HID + 2 * q + 1]}, acc[q]);
  }
#pragma unroll
  for (int q = 0; q < HID / 2; ++q) {
    a2[2 * q] = rc_lrelu(acc[q].x + th[g.o_b2 + 2 * q]);
    a2[2 * q + 1] = rc_lrelu(acc[q].y + th[g.o_b2 + 2 * q + 1]);
  }
}

template <int HID>
__device__ __forceinline__ float head1(const float* __restrict__ w3, float b3, const float (&a2)[HID]) {
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) v = fmaf(a2[k], w3[k], v);
  return v + b3;
}

// Sum K per-lane values over the 256 lanes (4 wavefronts) of the workgroup and store the K
// totals to out[0..K).  red: >= 4*K floats of LDS.  Two barriers per call.
template <int K>
__device__ __forceinline__ void block_reduce_store(float (&vals)[K], float* red, float* __restrict__ out) {
#pragma unroll
  for (int k = 0; k < K; ++k) vals[k] = rc_wave_sum_lane63(vals[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) red[w * K + k] = vals[k];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += ROWS) out[k] = (red[k] + red[K + k]) + (red[2 * K + k] + red[3 * K + k]);
}

// ---------------------------------------------------------------------------------------------
// k_mid_fit ("v5", the form that survived rounds 1-2: a DPP-tree form, two forms with VALU layer products and one with
// 32x32 transposed layer products were built, measured slower and removed -- DESIGN.md section 5 keeps the numbers):
// layers 2-3 forward + MSE + backward for one replay row per lane, every product on the f32-input MFMA forms.
//   * the two 20x20 layer products as v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer products per instruction,
//     block b = lanes 4b..4b+3: D_b[i][j] += A(lane 4b+i) * B(lane 4b+j), lane 4b+j holding column j in its four registers.
//     With B = the lane's OWN value (its replay row's a1[m] / dz2[k]) and A = W2[m][4g+i] (four units of group g, the same
//     for every block), register i of lane l accumulates unit 4g+i of ROW l: the result is born row-per-lane, 20 units are
//     five groups exactly, m ascending = one fmaf chain per (unit, row);
//   * everything summed over rows except gb1 comes out of ONE 32x32 reduction product per 32 rows,
//       A = [a1 | 1 | a2[0..10]*dv]^T,  B = [dz2 | 1 | dv | diff^2 | a2[11..19]*dv]
//     (G[i<20][j<20] = gW2, G[20][j<20] = gb2, G[21+i][20] = gW3[i], G[20][21] = gb3, G[20][22] = loss,
//     G[20][23+i] = gW3[11+i]); gb1 = sum dz1 by six fused-DPP adds per value.
// EMIT: instead of overwriting a1t with dz1 (fp32, feature-major), write dz1 as three exact bf16 pieces in the packed layout
// the lattice backward GEMM reads (rcmarl_lattice.h: rows = (agent,unit) column, reduction = replay row); rows b >= B of the
// last chunk are written as zeros.
// NOTE (round 3, profiles/r03_pipe_overlap_cycles.txt): the f32-input MFMA runs on the vector ALUs -- its cycles ADD to
// the VALU's; this kernel's time is the sum of its 232 f32 MFMAs, ~500 VALU and ~190 LDS instructions per 64 rows.
template <int HID, bool EMIT, bool DZ16 = false>
__global__ __launch_bounds__(256, EMIT ? 3 : 4) void k_mid_fit_v5(float* __restrict__ a1t, const float* __restrict__ theta,
                                                       const float* __restrict__ y, float* __restrict__ partials, int N,
                                                       int B, int in_dim, int ldp, int ldb, int nchunk, int cpw,
                                                       unsigned char* __restrict__ dzp, int dzp_rt, int dzp_kt,
                                                       const int* __restrict__ fix_flags, const int* __restrict__ fix_gen) {
  static_assert(HID == 20, "panel layout of the reduction product is written for 20 units");
  // fix-up launch behind k_mid_fit_v8 (rcmarl_mid_fit_lattice): only the agents whose operands left the f16 range there
  if (fix_flags != nullptr && fix_flags[blockIdx.z * N + blockIdx.y] != *fix_gen) return;
  typedef FitPart<HID> PT;
// (rows per pass of the reduction product: 32; 16 halves the panels -- 22.7 instead of 39 KiB of LDS per workgroup --
// and measured slower, 877 / 1071 against 857 / 985 us: occupancy is not limited by the LDS here)
#ifndef RC_V5_PLD
#define RC_V5_PLD 33
#endif
  constexpr int WLD = 32, PLD = RC_V5_PLD, GA = 11;    // padded weight rows; panel row stride (rows per pass + 1); gW3 values in the A panel
  constexpr int PANEL = (2 * 32 * PLD > 528 ? 2 * 32 * PLD : 528);   // floats per wavefront: A panel | B panel (later: its record)
  __shared__ __attribute__((aligned(16))) float sW2[HID * WLD];        // W2[m][i]   (i >= HID: zeros)
  __shared__ __attribute__((aligned(16))) float sW2T[HID * WLD];       // W2[j][k] stored as [k][j]
  __shared__ __attribute__((aligned(16))) float sV[2 * HID + 4];         // b2 | W3 | b3
  __shared__ __attribute__((aligned(16))) float sPn[4 * PANEL];
  static_assert(!EMIT || (PANEL * 4 >= HID * 3 * 64 * 2 && PANEL % 4 == 0), "the dz transpose (HID x 3 pieces x 64 rows of bf16) fits the panel area");
  static_assert(PANEL >= PT::SIZE + 64, "a staged record (+ one dump word per lane) fits the panel area");
  const int s = blockIdx.z, i = blockIdx.y;
  const int c_begin = blockIdx.x * cpw, c_end = min(nchunk, c_begin + cpw);
  const int r = threadIdx.x;
  const int lane = r & 63, wave = r >> 6, l31 = lane & 31, half = lane >> 5;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  const long row0 = ((long)s * N + i) * HID;
  if (EMIT && DZ16) rc_f16_saturate();
  for (int e = r; e < HID * WLD; e += ROWS) {
    const int m = e / WLD, c = e - m * WLD;
    sW2[e] = c < HID ? th[g.o_W2 + m * HID + c] : 0.f;
    sW2T[e] = c < HID ? th[g.o_W2 + c * HID + m] : 0.f;
  }
  if (r < 2 * HID + 1) sV[r] = th[g.o_b2 + r];
  float* sA = sPn + wave * PANEL;
  float* sB = sA + 32 * PLD;
  // where element q of the lane's result tile D[(q&3) + 8*(q>>2) + 4*half][l31] goes inside a record (else: dump word)
  int rec_idx[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int ii = (q & 3) + 8 * (q >> 2) + 4 * half, jj = l31;
    int idx = -1;
    if (ii < HID) idx = jj < HID ? ii * HID + jj : -1;
    else if (ii == HID) idx = jj < HID ? PT::gb2 + jj : (jj == HID + 1 ? PT::gb3 : (jj == HID + 2 ? PT::loss : (jj >= HID + 3 ? PT::gW3 + GA + (jj - HID - 3) : -1)));
    else idx = jj == HID ? PT::gW3 + (ii - HID - 1) : -1;
    rec_idx[q] = idx >= 0 ? idx : PT::SIZE + lane;
  }
  const float* yrow = y + ((long)s * N + i) * ldb;
  __syncthreads();
  static_assert(PT::SIZE <= 2 * ROWS, "a thread carries at most two elements of the workgroup's record");
  float racc0 = 0.f, racc1 = 0.f;
  // (requesting the activations of chunk c+1 while chunk c is processed costs 20 registers = one wavefront
  // per SIMD of occupancy here and measured slower: 919 vs 857 us)
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int b = chunk * ROWS + r;
    const bool valid = b < B;
    float a1[HID];
    load_a1<HID>(a1t, row0, ldb, b, valid, a1);
    const float ycur = valid ? yrow[b] : 0.f;
    // ---- layer 2 forward
    float z2[HID];
    {
      static_assert(HID % 4 == 0, "units in groups of four");
      rc_f32x4 zq[HID / 4];
#pragma unroll
      for (int g4 = 0; g4 < HID / 4; ++g4) zq[g4] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mg = 0; mg < HID / 4; ++mg)
#pragma unroll
        for (int g4 = 0; g4 < HID / 4; ++g4) {       // sW2T[unit][m] = W2[m][unit]: four consecutive m in one read
          const float4 w = *reinterpret_cast<const float4*>(&sW2T[(4 * g4 + (lane & 3)) * WLD + 4 * mg]);
          zq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, a1[4 * mg + 0], zq[g4], 0, 0, 0);
          zq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, a1[4 * mg + 1], zq[g4], 0, 0, 0);
          zq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, a1[4 * mg + 2], zq[g4], 0, 0, 0);
          zq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, a1[4 * mg + 3], zq[g4], 0, 0, 0);
        }
#pragma unroll
      for (int g4 = 0; g4 < HID / 4; ++g4)
#pragma unroll
        for (int e = 0; e < 4; ++e) z2[4 * g4 + e] = zq[g4][e];
    }
    float a2[HID];
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < HID; ++k) a2[k] = rc_lrelu(z2[k] + sV[k]);
#pragma unroll
    for (int k = 0; k < HID; ++k) v = fmaf(a2[k], sV[HID + k], v);
    v += sV[2 * HID];
    const float diff = valid ? v - ycur : 0.f;
    const float dv = (2.0f * diff) / (float)B;
    float dz2[HID], g3[HID];
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      g3[k] = a2[k] * dv;
      dz2[k] = dv * sV[HID + k] * rc_lrelu_grad_from_act(a2[k]);
    }
    // ---- the reduction product over this wavefront's 64 rows, QR rows at a time through its panels
    rc_f32x16 g1;
#pragma unroll
    for (int q = 0; q < 16; ++q) g1[q] = 0.f;
    constexpr int QR = PLD - 1;                        // rows per pass (16 or 32)
#pragma unroll
    for (int pp = 0; pp < 64 / QR; ++pp) {
      RC_WAVE_SYNC();                                  // earlier fragment reads / record reads of this area are done
      if (lane / QR == pp) {
        const int c = lane % QR;
#pragma unroll
        for (int k = 0; k < HID; ++k) {
          sA[k * PLD + c] = a1[k];
          sB[k * PLD + c] = dz2[k];
        }
        sA[HID * PLD + c] = 1.f;
        sB[HID * PLD + c] = 1.f;
#pragma unroll
        for (int q = 0; q < GA; ++q) sA[(HID + 1 + q) * PLD + c] = g3[q];
        sB[(HID + 1) * PLD + c] = dv;
        sB[(HID + 2) * PLD + c] = diff * diff;
#pragma unroll
        for (int q = 0; q < HID - GA; ++q) sB[(HID + 3 + q) * PLD + c] = g3[GA + q];
      }
      RC_WAVE_SYNC();
      const int ia = l31 * PLD + half;
#pragma unroll
      for (int m = 0; m < QR / 2; ++m) g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[ia + 2 * m], sB[ia + 2 * m], g1, 0, 0, 0);
    }
    // ---- layer 2 backward
    float dz1[HID];
    {
      rc_f32x4 dq[HID / 4];
#pragma unroll
      for (int g4 = 0; g4 < HID / 4; ++g4) dq[g4] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kg = 0; kg < HID / 4; ++kg)
#pragma unroll
        for (int g4 = 0; g4 < HID / 4; ++g4) {       // sW2[j][k] = W2[j][k]: da1[j] = sum_k dz2[k] W2[j][k], k ascending
          const float4 w = *reinterpret_cast<const float4*>(&sW2[(4 * g4 + (lane & 3)) * WLD + 4 * kg]);
          dq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, dz2[4 * kg + 0], dq[g4], 0, 0, 0);
          dq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, dz2[4 * kg + 1], dq[g4], 0, 0, 0);
          dq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, dz2[4 * kg + 2], dq[g4], 0, 0, 0);
          dq[g4] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, dz2[4 * kg + 3], dq[g4], 0, 0, 0);
        }
#pragma unroll
      for (int g4 = 0; g4 < HID / 4; ++g4)
#pragma unroll
        for (int e = 0; e < 4; ++e) dz1[4 * g4 + e] = dq[g4][e] * rc_lrelu_grad_from_act(a1[4 * g4 + e]);
    }
    if (EMIT) {
      // packed bf16 pieces (rcmarl_lattice.h): row = i*HID + j, k = replay row.  A lane owns ONE replay row, i.e. 2 bytes of
      // every (unit, piece) row of the packed image: stored directly that is 120 two-byte store instructions per
      // wavefront.  Instead the wavefront transposes its 64 rows x 60 (unit, piece) values through its (now idle) panel
      // area and writes 16-byte chunks = 8 consecutive replay rows of one (unit, piece): 8 store instructions.
      // DZ16: two f16 pieces of 2^8 dz1 instead of three bf16 pieces of dz1 (rcmarl_lattice.h).
      constexpr int NP = DZ16 ? 2 : 3;
      unsigned short* stg = reinterpret_cast<unsigned short*>(sA);          // [HID * NP (unit, piece)][64 rows] 16-bit = 7680 / 5120 B
      RC_WAVE_SYNC();                                  // the reduction product's fragment reads are done
#pragma unroll
      for (int q = 0; q < HID / 2; ++q) {
        if constexpr (DZ16) {
          unsigned h, l;
          rc_split2h_pair(dz1[2 * q] * RC_F16_DZ_SCALE, dz1[2 * q + 1] * RC_F16_DZ_SCALE, h, l);
          stg[((2 * q) * 2 + 0) * 64 + lane] = (unsigned short)h;
          stg[((2 * q) * 2 + 1) * 64 + lane] = (unsigned short)l;
          stg[((2 * q + 1) * 2 + 0) * 64 + lane] = (unsigned short)(h >> 16);
          stg[((2 * q + 1) * 2 + 1) * 64 + lane] = (unsigned short)(l >> 16);
        } else {
          unsigned h, m, l;
          rc_split3_pair(dz1[2 * q], dz1[2 * q + 1], h, m, l);              // bits 0-15: unit 2q, bits 16-31: unit 2q+1
          stg[((2 * q) * 3 + 0) * 64 + lane] = (unsigned short)h;
          stg[((2 * q) * 3 + 1) * 64 + lane] = (unsigned short)m;
          stg[((2 * q) * 3 + 2) * 64 + lane] = (unsigned short)l;
          stg[((2 * q + 1) * 3 + 0) * 64 + lane] = (unsigned short)(h >> 16);
          stg[((2 * q + 1) * 3 + 1) * 64 + lane] = (unsigned short)(m >> 16);
          stg[((2 * q + 1) * 3 + 2) * 64 + lane] = (unsigned short)(l >> 16);
        }
      }
      RC_WAVE_SYNC();
      unsigned char* base = dzp + (long)s * dzp_rt * dzp_kt * (NP * RC_PK_BLOCK);
      const int k0 = chunk * ROWS + wave * 64;         // first replay row of this wavefront (a multiple of 64: two k-tiles)
#pragma unroll
      for (int it = 0; it < (HID * NP * 8 + 63) / 64; ++it) {
        const int c = it * 64 + lane;                  // chunk index: (unit, piece) = c >> 3, rows 8*(c&7) .. +7
        if (c < HID * NP * 8) {
          const int up = c >> 3, c8 = c & 7;
          const int unit = up / NP, piece = up - NP * unit;
          const int R = i * HID + unit;
          const int kt = (k0 >> 5) + (c8 >> 2), c4 = c8 & 3;
          if (kt < dzp_kt) {
            const uint4 v4 = *reinterpret_cast<const uint4*>(stg + up * 64 + 8 * c8);
            const unsigned off = (unsigned)(((R >> 7) * dzp_kt + kt) * NP + piece) * RC_PK_BLOCK + (unsigned)(R & 127) * 64 +
                                 (unsigned)((c4 ^ ((R >> 2) & 3)) << 4);
            *reinterpret_cast<uint4*>(base + off) = v4;
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < HID; ++j)
        if (valid) a1t[(row0 + j) * ldb + b] = dz1[j];
    }
    // ---- gb1[j] = sum over the 64 rows of dz1[j]
    {
      float sm[HID + 1];
#pragma unroll
      for (int j = 0; j < HID; ++j) sm[j] = dz1[j];
      sm[HID] = 0.f;
      static_assert((HID + 1) % 3 == 0, "sums are reduced three at a time");
#pragma unroll
      for (int q = 0; q < (HID + 1) / 3; ++q) rc_wave_sum3_lane63(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
      RC_WAVE_SYNC();                                  // this wavefront's panel reads are done: the area takes its record
#pragma unroll
      for (int q = 0; q < 16; ++q) sA[rec_idx[q]] = g1[q];
      if (lane == 63) {
#pragma unroll
        for (int j = 0; j < HID; ++j) sA[PT::gb1 + j] = sm[j];
      }
    }
    __syncthreads();
    // the four wavefronts' records of this chunk, added to the workgroup's running record (ONE record per workgroup leaves the
    // kernel: a twelfth to a half of the per-chunk records rcmarl_small_sgd used to re-read)
    racc0 += (sPn[r] + sPn[PANEL + r]) + (sPn[2 * PANEL + r] + sPn[3 * PANEL + r]);
    if (r + ROWS < PT::SIZE) racc1 += (sPn[r + ROWS] + sPn[PANEL + r + ROWS]) + (sPn[2 * PANEL + r + ROWS] + sPn[3 * PANEL + r + ROWS]);
    __syncthreads();                                   // records read out before the next chunk's panels land
  }
  float* out = partials + (((long)s * N + i) * gridDim.x + blockIdx.x) * PT::SIZE;
  out[r] = racc0;
  if (r + ROWS < PT::SIZE) out[r + ROWS] = racc1;
}

// ---------------------------------------------------------------------------------------------
// k_mid_fit, f16 matrix-core form ("v8", round 3) -- the default behind rcmarl_mid_fit_lattice when the backward operand is
// carried as f16 pieces (RCMARL_LAT_F16 bit 1).
// Why: v_mfma_f32_* (every product of v5) executes on the vector ALUs -- measured in shader cycles, its time ADDS to the VALU's
// (profiles/r03_pipe_overlap_cycles.txt) -- while the 16-bit matrix core is a separate pipe that VALU work hides behind and is 16x
// faster per flop.  Here all three products of the step run as v_mfma_f32_32x32x16_f16 on TWO f16 pieces of their (scaled) fp32
// operands (rcmarl_lattice.h: x = h + l to one unit in the last place of x): a product x*y is hh + hl + lh + ll, four exact
// f16 x f16 products, fp32 accumulate, smallest first; a split costs 5 VALU per value pair.  Scales: W2'' = 2^10 W2,
// dz2'' = 2^10 dz2, a1 unscaled; the results are multiplied back (exact).  Results differ from v5's fmaf chains in the last bits
// (same bars vs the oracle: tests), NOT bit-identical to v5.
// Range: an agent whose a1, 2^10 dz2 or 2^10 W2 leaves the f16 range (a fit that blew up: clipped operands would feed the blow-up
// instead of letting fp32 arithmetic saturate it -- measured, profiles/r03r_*) is FLAGGED (ovf_flags[seed][agent] = the call's
// generation number) and recomputed by k_mid_fit_v5 in a second launch that exits at once for every other agent.
//
// Layout.  A wavefront owns 64 replay rows as two blocks of 32.  In a block, lane (j = lane&31, h = lane>>5) works for row
// j and holds TEN of its 20 units: U_0 = {0..7, 16, 17}, U_1 = {8..15, 18, 19} (local index u = 0..9).  That is exactly the
// B-operand shape of the 32x32x16 instruction for a contraction over units -- lane (row j, k-group h) supplies 8 contraction
// slots per k-step: slots 8h..8h+7 of step 0 are local units 0..7, of step 1 local units 8, 9 and six zeros -- and, with
// the rows of the weight operand (A) permuted the same way, also its OUTPUT shape: accumulator register r of lane (j, h) is
// local unit r of row j (r < 10).  So a1 -> z2 -> a2 -> dz2 -> da1 -> dz1 never leave their lanes: no swaps, no LDS.
//   z2[row][unit]  = sum_m a1[row][m] W2[m][unit]      A = W2^T (permuted rows/slots), B = a1 pieces
//   da1[row][m]    = sum_j dz2[row][j] W2[m][j]        A = W2   (permuted),            B = dz2 pieces
// The row REDUCTION gW2 = a1^T dz2 (+ gb2 as its ones row) contracts over ROWS, i.e. over the lane axis: the pieces the layer
// products already made are written row-major into per-wavefront LDS planes ([piece][row][unit], one 16-byte + one 4-byte
// store per piece) and read back TRANSPOSED with ds_read_b64_tr_b16 (a lane = one unit, eight consecutive rows = one MFMA
// operand): no second split.  The plain row sums (gb1, gW3, gb3, loss: 22 values) are fused-DPP half-wave sums.
// History: the same data flow on three exact bf16 pieces (six products, 9 VALU per split pair; "v7") ran
// 827 us against v5's 797 at BASELINE configs[3] -- the splits, plane traffic and dependent MFMA chains gave back what the
// matrix core saved; with two f16 pieces it runs ~600 us against v5's ~715 (profiles/r03n_mid_ab.txt).
#ifndef RC_V8_WAVES
#define RC_V8_WAVES 3                    // wavefronts per SIMD the register allocation aims at
#endif
#define RC_V8_S 1024.f                   // scale of W2 and of dz2
#define RC_V8_US 0.0009765625f
#define RC_V8_RANGE 65000.f              // a (scaled) operand beyond this would saturate: the agent is flagged and redone by k_mid_fit_v5
template <int HID, bool EMIT>
__global__ __launch_bounds__(256, RC_V8_WAVES) void k_mid_fit_v8(float* __restrict__ a1t, const float* __restrict__ theta,
                                                    const float* __restrict__ y, float* __restrict__ partials, int N,
                                                    int B, int in_dim, int ldp, int ldb, int nchunk, int cpw,
                                                    unsigned char* __restrict__ dzp, int dzp_rt, int dzp_kt,
                                                    int* __restrict__ ovf_flags, const int* __restrict__ ovf_gen_p) {
  static_assert(HID == 20, "unit sets and panel layout are written for 20 units");
  typedef FitPart<HID> PT;
  constexpr int LU = 10;                               // units per lane
  // A piece plane = two windows (columns 0-15 | 16-31) of [32 rows][16 columns] f16, 32 bytes per row: the 16-lane group of a
  // transpose read then fetches 128 CONTIGUOUS bytes (four rows of one window) -- the one layout the LDS serves without bank
  // conflicts (cdna guide, T10).  Rows 4-7 of every eight hold their two 16-byte halves swapped: the eight consecutive lanes a
  // 16-byte write is served by then cover all 32 banks.
  // Columns: 20 units | 1.0 (A planes) | unused; what the unused columns hold only feeds elements of G nobody reads.
  constexpr int PLANE = 2 * 32 * 16;                   // f16 elements of one piece plane
  // Window 1 stores rows 4-7 of every eight BEFORE rows 0-3 (the two 16-lane groups a transpose read serves together -- one per
  // window -- then sit in opposite halves of the 256-byte bank row) and rotates its four 8-byte column groups by row >> 2
  // (the 4-byte writes of units 16-19: 2-way instead of 4-way).
  auto pl_addr = [](int row, int col) {
    const int win = col >> 4, c = col & 15, x = (row >> 2) & 1;
    if (win == 0) return row * 16 + (c ^ (8 * x));
    return 512 + (row ^ 4) * 16 + ((((c >> 2) + (row >> 2)) & 3) << 2) + (c & 3);
  };
  constexpr int PANEL_B = 2 * 2 * PLANE * 2;           // bytes per wavefront: A planes (a1 | 1) then B planes (dz2; later dz transpose, record)
  __shared__ __attribute__((aligned(16))) uint4 sWf[2 * 2 * 2 * 2 * 32];   // [product][k-step][piece][k-group][row i]: 16-byte A fragments
  __shared__ __attribute__((aligned(16))) float sV[2 * HID + 4];           // b2 | W3 | b3
  __shared__ __attribute__((aligned(16))) unsigned char sPn[4 * PANEL_B];
  static_assert(2 * PLANE * 2 >= HID * 2 * 32 * 2, "the dz transpose (HID x 2 pieces x 32 rows of f16) fits the B planes");
  static_assert(2 * PLANE * 2 >= (PT::SIZE + 64) * 4, "a staged record (+ one dump word per lane) fits the B planes");
  const int s = blockIdx.z, i = blockIdx.y;
  const int c_begin = blockIdx.x * cpw, c_end = min(nchunk, c_begin + cpw);
  const int r = threadIdx.x;
  const int lane = r & 63, wave = __builtin_amdgcn_readfirstlane(r >> 6), l31 = lane & 31, half = lane >> 5;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  const long row0 = ((long)s * N + i) * HID;
  rc_f16_saturate();
  const int ovf_gen = *ovf_gen_p;                       // this launch pair's generation number (bumped on the device: see mid_flags)
  // ---- the agent's 2^10 W2 as f16 pieces in A-fragment order, both orientations (once per workgroup)
  {
    unsigned short* wf16 = reinterpret_cast<unsigned short*>(sWf);
    for (int e = r; e < 2 * 32 * 32; e += ROWS) {
      const int prod = e >> 10, ri = (e >> 5) & 31, k = e & 31;
      const int ui = v8_row_unit(ri), uk = v8_slot_unit(k);
      float w = 0.f;
      if (ui >= 0 && uk >= 0) w = prod == 0 ? th[g.o_W2 + uk * HID + ui] : th[g.o_W2 + ui * HID + uk];
      unsigned ph, pl;
      rc_split2h_pair(w * RC_V8_S, 0.f, ph, pl);
      if (fabsf(w) * RC_V8_S > RC_V8_RANGE) ovf_flags[s * N + i] = ovf_gen;
      const int ks = k >> 4, kg = (k >> 3) & 1;
      const int base = ((((prod * 2 + ks) * 2 + 0) * 2 + kg) * 32 + ri) * 8 + (k & 7);     // in f16 elements; piece stride 2*32*8
      wf16[base] = (unsigned short)ph;
      wf16[base + 2 * 32 * 8] = (unsigned short)pl;
    }
  }
  if (r < 2 * HID + 1) sV[r] = th[g.o_b2 + r];
  unsigned short* pA = reinterpret_cast<unsigned short*>(sPn + wave * PANEL_B);     // [piece][row][column]: a1 units 0..19 | 1.0 | unused
  unsigned short* pB = pA + 2 * PLANE;                                              // [piece][row][column]: dz2 units 0..19 | unused
  float* sRec = reinterpret_cast<float*>(pB);                                       // the wavefront's record, staged over the B planes
  // column 20 of the A planes is the constant 1 (-> gb2 = sum dz2): pieces (1.0, 0); written once, nothing else touches it.
  // The spare columns of the A planes are zeroed once; columns 20.. of the B planes (and what a transpose read picks up
  // beyond column 23) only feed elements of G nobody reads.
  for (int e = lane; e < 2 * 32 * 12; e += 64) {
    const int pc = e / (32 * 12), rw = (e / 12) & 31, cl = 20 + e % 12;
    pA[pc * PLANE + pl_addr(rw, cl)] = (pc == 0 && cl == 20) ? (unsigned short)0x3C00 : (unsigned short)0;
  }
  const float* yrow = y + ((long)s * N + i) * ldb;
  __syncthreads();
  const float b3 = sV[2 * HID];
  const uint4* wfA = sWf + half * 32 + l31;             // + ((prod*2 + ks)*2 + piece) * 64
  auto loadA = [&](int prod, int ks) {
    V8Pieces a;
    a.h = wfA[((prod * 2 + ks) * 2 + 0) * 64];
    a.l = wfA[((prod * 2 + ks) * 2 + 1) * 64];
    return a;
  };
  // transpose-read address of this lane inside a plane (rcmarl_lattice.h: rc_lds_read_tr16): as MFMA operand lane (i = l31,
  // k-group = half) it wants rows 8*half .. 8*half+7 of column l31 -> two reads of four rows; its 16-lane group fetches the
  // [4 rows][16 columns] block of columns 16*((lane>>4)&1).., this lane row (lane&15)>>2 of it, columns 4*(lane&3)..
  int tr_off[2];                                        // rows 0-3 / 4-7 of the lane's eight (+ 16 rows per k-step)
#pragma unroll
  for (int t = 0; t < 2; ++t) tr_off[t] = pl_addr(8 * half + 4 * t + ((lane & 15) >> 2), 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  // this lane's own slots in a plane: row l31, columns 8*half..8*half+7 (local units 0..7) and 16+2*half, 17+2*half (8, 9)
  const int wr8 = pl_addr(l31, 8 * half), wr2 = pl_addr(l31, 16 + 2 * half);
  uint4 z4;
  z4.x = z4.y = z4.z = z4.w = 0u;
  float amax = 0.f;                                    // largest |operand| this lane split (a1, 2^10 dz2)
  // Everything summed over rows accumulates per wavefront over ALL the workgroup's chunks -- the row reduction in the accumulator
  // tile g1, the plain sums per lane -- and is reduced ONCE at the end (round 4: the per-chunk reduction, 120 fused-DPP adds + the
  // record staging + two workgroup barriers per 64 rows, was a fifth of this kernel's instructions).
  rc_f32x16 g1;
#pragma unroll
  for (int q = 0; q < 16; ++q) g1[q] = 0.f;
  float gb1l[LU], gw3l[LU], gb3a = 0.f, lossa = 0.f;    // per-lane partial sums over this wavefront's blocks
#pragma unroll
  for (int u = 0; u < LU; ++u) gb1l[u] = gw3l[u] = 0.f;
  // The lane's ten layer-1 activations + target of a block are loaded ONE BLOCK AHEAD (round 4: a block is one dependent chain --
  // loads, split, matrix core, vector ALU, matrix core, ... -- and with three wavefronts per SIMD nothing else covered the ~2 us
  // an HBM load takes under load: the wavefronts sat 43 % of their cycles waiting).  Unconditional loads from a clamped row
  // (ten in flight, no exec-mask branches).  Addresses: one wave-uniform base
  // per local unit (scalar registers) + ONE 32-bit byte offset per lane and unit group -- no 64-bit vector address arithmetic.
  const unsigned char* ubase[LU];
#pragma unroll
  for (int u = 0; u < LU; ++u) ubase[u] = reinterpret_cast<const unsigned char*>(a1t + (row0 + (u < 8 ? u : u + 8)) * ldb);
  const unsigned hoff8 = (unsigned)(8 * half) * (unsigned)ldb, hoff2 = (unsigned)(2 * half) * (unsigned)ldb;   // v8_unit(half, u) - v8_unit(0, u) rows
  float setA[LU], yA;                                  // the prefetched block
  auto fetch = [&](float (&nxa)[LU], float& nxy, int bfirst) {
    const unsigned bc = (unsigned)min(bfirst + l31, B - 1);
    const unsigned o8 = (hoff8 + bc) * 4u, o2 = (hoff2 + bc) * 4u;
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      nxa[u] = *reinterpret_cast<const float*>(ubase[u] + (u < 8 ? o8 : o2));
      RC_SCHED_FENCE();                                  // (the same issue order at both call sites)
    }
    nxy = yrow[bc];
    RC_SCHED_FENCE();
  };
  // A block's packed dz1 chunks leave the staging planes at the TOP of the next block, behind that block's wait for its prefetched
  // loads: the memory counter is in order, so stores issued at the end of a block would be waited for -- a round trip to L2 -- at
  // the top of the next one; issued there, everything outstanding at a wait is one block old.
  // The 16-byte chunks of the packed image (8 consecutive replay rows of one (unit, piece)) come back from the staging through the
  // TRANSPOSE read: pass p, the 16-lane group g of a wavefront reads window p of rows 8g..8g+7, lane j of it receives column j.
  // Staging = three windows of [32 rows][32 bytes], the layout of the piece planes (same swap of the 16-byte halves in rows 4-7
  // of every eight): window m, row = replay row, bytes 0-15 = the lower half-wave's m-th piece register group, bytes 16-31 the
  // upper's.  Piece groups of a lane: h of its local units 0-7 | l of units 0-7 | h, l of units 8, 9: 16, 16 and 8 bytes.
  // (and rows 4-7 before rows 0-3 in every second group of eight: the two 16-lane groups a transpose read serves together)
  auto st_addr = [&](int k, int m, int w) { return m * 1024 + 32 * (k ^ (4 * ((k >> 3) & 1))) + (w ^ (16 * ((k >> 2) & 1))); };
  unsigned st_wr[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) st_wr[m] = (unsigned)st_addr(l31, m, 16 * half);
  unsigned st_rd[2];                                       // transpose-read address of the lane for rows 0-3 / 4-7 of its group, window 0
#pragma unroll
  for (int t = 0; t < 2; ++t) st_rd[t] = (unsigned)st_addr(8 * (lane >> 4) + 4 * t + ((lane & 15) >> 2), 0, 8 * (lane & 3));
  auto st_pass = [&](int ps, int t) { (void)t; return (unsigned)ps * 1024u; };
  unsigned st_off[3];                                   // byte offset of the lane's chunk inside the k-tile of the packed image, per pass
  bool st_on[3];
#pragma unroll
  for (int ps = 0; ps < 3; ++ps) {
    // (lanes j, j+16, j+32, j+48 -- not four neighbours -- write the four chunks of a 64-byte segment; a probe build with the
    // neighbour mapping (and wrong placement) ran in the same time: 471 against 466 us, so the stores are left as the transpose delivers them)
    const int j = lane & 15, hf = j >> 3, cc = j & 7;    // column j of window ps -> (half, pair q, element e)
    const int c4 = lane >> 4;                            // c4: the 8-row group
    const int lu = ps < 2 ? cc : 8 + (cc & 1), piece = ps < 2 ? ps : (cc >> 1) & 1;      // local unit, piece of this column
    const int unit = v8_unit(hf, lu);
    const int R = i * HID + unit;
    st_on[ps] = ps < 2 || cc < 4;
    st_off[ps] = (unsigned)(((R >> 7) * dzp_kt) * 2 + piece) * RC_PK_BLOCK + (unsigned)(R & 127) * 64 + (unsigned)((c4 ^ ((R >> 2) & 3)) << 4);
  }
  unsigned char* dz_base = dzp + (long)s * dzp_rt * dzp_kt * (2 * RC_PK_BLOCK);
  int kt_staged = -1;
  auto store_staged = [&](int kt) {
    const unsigned short* stg = pB;
    if (kt < dzp_kt) {
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        const uint2 t0 = rc_lds_read_tr16(stg + ((st_rd[0] + st_pass(ps, 0)) >> 1)), t1 = rc_lds_read_tr16(stg + ((st_rd[1] + st_pass(ps, 1)) >> 1));
        uint4 v4;
        v4.x = t0.x; v4.y = t0.y; v4.z = t1.x; v4.w = t1.y;
        if (st_on[ps]) *reinterpret_cast<uint4*>(dz_base + (st_off[ps] + (unsigned)kt * (2 * RC_PK_BLOCK))) = v4;
      }
    }
  };
  const float Bf = (float)B, rB = 1.0f / Bf;
  const rc_f2 leak2 = rc_bcast2(RC_LEAK);
  auto do_block = [&](float (&a1l)[LU], const float ycur, float (&nxa)[LU], float& nxy, const int chunk, const int blk) {
      const int bfirst = chunk * ROWS + wave * 64 + 32 * blk;                 // wave-uniform
      const int b = bfirst + l31;
      const bool valid = b < B;
      // (rows beyond B carry row B-1's activations, not zeros: their diff -- hence dz2, dz1 and every sum they enter -- is zero)
      fetch(nxa, nxy, bfirst + (blk ? ROWS - 32 : 32));    // the next block of this wavefront (past the workgroup's last: a clamped, unused read)
      if (EMIT && kt_staged >= 0) store_staged(kt_staged);
#pragma unroll
      for (int u = 0; u < LU; u += 2) amax = rc_amax3(amax, a1l[u], a1l[u + 1]);
      // ---- layer 2 forward on the f16 matrix core; the a1 pieces also go to the A planes (operand of the row reduction)
      V8Pieces pa0, pa1;
      {
        const float x0[8] = {a1l[0], a1l[1], a1l[2], a1l[3], a1l[4], a1l[5], a1l[6], a1l[7]};
        pa0 = v8_split8<false>(x0, 1.f);
        pa1.h = z4; pa1.l = z4;
        rc_split2h_pair(a1l[8], a1l[9], pa1.h.x, pa1.l.x);
      }
      RC_SCHED_FENCE();
      RC_WAVE_SYNC();                                    // the previous block's transpose reads of the planes are done
      *reinterpret_cast<uint4*>(pA + 0 * PLANE + wr8) = pa0.h;
      *reinterpret_cast<uint4*>(pA + 1 * PLANE + wr8) = pa0.l;
      *reinterpret_cast<unsigned*>(pA + 0 * PLANE + wr2) = pa1.h.x;
      *reinterpret_cast<unsigned*>(pA + 1 * PLANE + wr2) = pa1.l.x;
      rc_f32x16 zz;
#pragma unroll
      for (int q = 0; q < 16; ++q) zz[q] = 0.f;
      RC_MX_BEGIN();
      zz = v8_mfma4(loadA(0, 1), pa1, zz);
      zz = v8_mfma4(loadA(0, 0), pa0, zz);
      RC_MX_END();
      RC_SCHED_FENCE();
      // a2 = lrelu(z2 + b2) two units at a time (max(z, leak z): bit for bit the select form), v = a2 . W3 + b3
      rc_f2 a2p[LU / 2], w3p[LU / 2];
      float vp = 0.f;
#pragma unroll
      for (int q = 0; q < LU / 2; ++q) {
        const rc_f2 zq = {zz[2 * q], zz[2 * q + 1]};
        const rc_f2 bq = {sV[v8_unit(half, 2 * q)], sV[v8_unit(half, 2 * q + 1)]};
        w3p[q] = rc_f2{sV[HID + v8_unit(half, 2 * q)], sV[HID + v8_unit(half, 2 * q + 1)]};
        a2p[q] = rc_lrelu2(rc_fma2(zq, rc_bcast2(RC_V8_US), bq));
      }
#pragma unroll
      for (int q = 0; q < LU / 2; ++q) { vp = fmaf(a2p[q].x, w3p[q].x, vp); vp = fmaf(a2p[q].y, w3p[q].y, vp); }
      float va = vp, vb = vp;
      rc_swap32(va, vb);                                 // va: lanes 32-63 now hold the low half's partial; vb: lanes 0-31 the high half's
      const float v = (vp + (half ? va : vb)) + b3;
      const float diff = valid ? v - ycur : 0.f;
      // dv = (2 diff) / B, correctly rounded without the division sequence (q = x rB; r = x - q B exactly; q + r rB: Markstein --
      // the rounded quotient whenever it is a normal number; B's significand is not all ones)
      const float x2 = 2.0f * diff, q0 = x2 * rB;
      const float dv = fmaf(fmaf(-q0, Bf, x2), rB, q0);
      const float dvs = dv * RC_V8_S;
      if (half == 0) { gb3a += dv; lossa = fmaf(diff, diff, lossa); }   // (both lanes of a row hold the same v: count it once)
      // 2^10 dz2 = (dvs W3) * (a2 > 0 ? 1 : leak)  (power-of-two scale: same bits as scaling afterwards)
      float dz2l[LU];
#pragma unroll
      for (int q = 0; q < LU / 2; ++q) {
        const rc_f2 t = rc_mul2(rc_bcast2(dvs), w3p[q]), tl = rc_mul2(t, leak2);
        const rc_f2 gw = rc_fma2(a2p[q], rc_bcast2(dv), rc_f2{gw3l[2 * q], gw3l[2 * q + 1]});
        gw3l[2 * q] = gw.x; gw3l[2 * q + 1] = gw.y;
        dz2l[2 * q] = a2p[q].x > 0.f ? t.x : tl.x;
        dz2l[2 * q + 1] = a2p[q].y > 0.f ? t.y : tl.y;
      }
#pragma unroll
      for (int u = 0; u < LU; u += 2) amax = rc_amax3(amax, dz2l[u], dz2l[u + 1]);
      // ---- layer 2 backward; the dz2 pieces also go to the B planes
      V8Pieces pd0, pd1;
      {
        const float x0[8] = {dz2l[0], dz2l[1], dz2l[2], dz2l[3], dz2l[4], dz2l[5], dz2l[6], dz2l[7]};
        pd0 = v8_split8<false>(x0, 1.f);
        pd1.h = z4; pd1.l = z4;
        rc_split2h_pair(dz2l[8], dz2l[9], pd1.h.x, pd1.l.x);
      }
      *reinterpret_cast<uint4*>(pB + 0 * PLANE + wr8) = pd0.h;
      *reinterpret_cast<uint4*>(pB + 1 * PLANE + wr8) = pd0.l;
      *reinterpret_cast<unsigned*>(pB + 0 * PLANE + wr2) = pd1.h.x;
      *reinterpret_cast<unsigned*>(pB + 1 * PLANE + wr2) = pd1.l.x;
      rc_f32x16 dd;
#pragma unroll
      for (int q = 0; q < 16; ++q) dd[q] = 0.f;
      RC_MX_BEGIN();
      dd = v8_mfma4(loadA(1, 1), pd1, dd);
      dd = v8_mfma4(loadA(1, 0), pd0, dd);
      RC_MX_END();
      RC_SCHED_FENCE();
      // dz1 = (dd * scale) * (a1 > 0 ? 1 : leak)   (EMIT: 2^8 dz1, what the packed operand carries)
      float dz1l[LU];
      const rc_f2 sc2 = rc_bcast2(RC_V8_US * RC_V8_US * (EMIT ? RC_F16_DZ_SCALE : 1.f));
#pragma unroll
      for (int q = 0; q < LU / 2; ++q) {
        const rc_f2 t = rc_mul2(rc_f2{dd[2 * q], dd[2 * q + 1]}, sc2), tl = rc_mul2(t, leak2);
        const rc_f2 dq = {a1l[2 * q] > 0.f ? t.x : tl.x, a1l[2 * q + 1] > 0.f ? t.y : tl.y};
        const rc_f2 gb = rc_add2(rc_f2{gb1l[2 * q], gb1l[2 * q + 1]}, dq);
        gb1l[2 * q] = gb.x; gb1l[2 * q + 1] = gb.y;
        dz1l[2 * q] = dq.x; dz1l[2 * q + 1] = dq.y;
      }
      if (!EMIT) {
#pragma unroll
        for (int u = 0; u < LU; ++u)
          if (valid) a1t[(row0 + v8_unit(half, u)) * ldb + b] = dz1l[u];
      }
      // ---- the row reduction G = [a1 | 1]^T [dz2] over this block's 32 rows: operands read back TRANSPOSED from the planes
      // (a lane = one column, eight consecutive rows), straight into the MFMA: no second split, no fp32 panels
      RC_WAVE_SYNC();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        V8Pieces ra, rb;
        const int o = tr_off[0] + 16 * 16 * ks, o4 = tr_off[1] + 16 * 16 * ks;
        uint2 t0, t1;
        t0 = rc_lds_read_tr16(pA + 0 * PLANE + o); t1 = rc_lds_read_tr16(pA + 0 * PLANE + o4);
        ra.h.x = t0.x; ra.h.y = t0.y; ra.h.z = t1.x; ra.h.w = t1.y;
        t0 = rc_lds_read_tr16(pA + 1 * PLANE + o); t1 = rc_lds_read_tr16(pA + 1 * PLANE + o4);
        ra.l.x = t0.x; ra.l.y = t0.y; ra.l.z = t1.x; ra.l.w = t1.y;
        t0 = rc_lds_read_tr16(pB + 0 * PLANE + o); t1 = rc_lds_read_tr16(pB + 0 * PLANE + o4);
        rb.h.x = t0.x; rb.h.y = t0.y; rb.h.z = t1.x; rb.h.w = t1.y;
        t0 = rc_lds_read_tr16(pB + 1 * PLANE + o); t1 = rc_lds_read_tr16(pB + 1 * PLANE + o4);
        rb.l.x = t0.x; rb.l.y = t0.y; rb.l.z = t1.x; rb.l.w = t1.y;
        g1 = v8_mfma4(ra, rb, g1);
        RC_SCHED_FENCE();                              // (keeps the second k-step's eight reads from being hoisted: registers)
      }
      if (EMIT) {
        // the f16 pieces of 2^8 dz1 (rcmarl_lattice.h) go to the (now idle) B planes in staging order; a block is exactly one k-tile
        // of the packed image
        unsigned short* stg = pB;
        RC_WAVE_SYNC();                                // the reduction's transpose reads are done
        uint4 ph, pl;                                   // window 0: the h pieces of local units 0-7, window 1: their l pieces,
        uint2 w2;                                       // window 2: h and l of units 8, 9 -- whole registers as the split delivers them
        {
          const float x0[8] = {dz1l[0], dz1l[1], dz1l[2], dz1l[3], dz1l[4], dz1l[5], dz1l[6], dz1l[7]};
          rc_split2h_x8(x0, ph, pl);
          rc_split2h_pair(dz1l[8], dz1l[9], w2.x, w2.y);
        }
        // (whole register groups: a 16-byte write assembled from registers of different groups is emitted as dword pairs,
        // which conflict 4-way on these 32-byte rows -- knock-out counters in profiles/r04u_lds_conflict_knockouts.txt)
        *reinterpret_cast<uint4*>(stg + (st_wr[0] >> 1)) = ph;
        *reinterpret_cast<uint4*>(stg + (st_wr[1] >> 1)) = pl;
        *reinterpret_cast<uint2*>(stg + (st_wr[2] >> 1)) = w2;
        RC_WAVE_SYNC();
        kt_staged = bfirst >> 5;                       // this block's k-tile: stored at the top of the next block
      }
  };
  fetch(setA, yA, c_begin * ROWS + wave * 64);
  // (one register set + a copy per block; two sets used alternately by a twice-unrolled loop save the ten copies and measure 6 %
  // SLOWER -- 498 against 468 us: profiles/r04q_mid_ab.txt)
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
      float cur[LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) cur[u] = setA[u];
      const float ycur = yA;
      do_block(cur, ycur, setA, yA, chunk, blk);
    }
  }
  if (EMIT && kt_staged >= 0) store_staged(kt_staged);
  {
    // ---- what is summed over rows outside the matrix core: gb1, gW3 (per unit), gb3, loss -- all blocks were added above per
    // lane; now over the 32 lanes of each half (results in lanes 31 and 63)
    {
      float sm[2 * LU + 4];
#pragma unroll
      for (int u = 0; u < LU; ++u) { sm[u] = gb1l[u] * (EMIT ? RC_F16_DZ_UNSCALE : 1.f); sm[LU + u] = gw3l[u]; }
      sm[2 * LU] = gb3a; sm[2 * LU + 1] = lossa; sm[2 * LU + 2] = sm[2 * LU + 3] = 0.f;
#pragma unroll
      for (int q = 0; q < (2 * LU + 4) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
      RC_WAVE_SYNC();                                  // this wavefront's plane / transpose reads are done: the B planes take its record
#pragma unroll
      for (int q = 0; q < 16; ++q) {               // element q of the lane's tile G[(q&3) + 8*(q>>2) + 4*half][l31] -> its record slot
        const int ii = (q & 3) + 8 * (q >> 2) + 4 * half, jj = l31;
        int idx = -1;
        if (jj < HID) idx = ii < HID ? ii * HID + jj : (ii == HID ? PT::gb2 + jj : -1);
        sRec[idx >= 0 ? idx : PT::SIZE + lane] = g1[q] * RC_V8_US;             // (else: a dump word of the lane's own); dz2 was carried as 2^10 dz2
      }
      if (l31 == 31) {
#pragma unroll
        for (int u = 0; u < LU; ++u) {
          sRec[PT::gb1 + v8_unit(half, u)] = sm[u];
          sRec[PT::gW3 + v8_unit(half, u)] = sm[LU + u];
        }
        if (half == 0) { sRec[PT::gb3] = sm[2 * LU]; sRec[PT::loss] = sm[2 * LU + 1]; }
      }
    }
  }
  __syncthreads();
  if (amax > RC_V8_RANGE) ovf_flags[s * N + i] = ovf_gen;            // (same value from every lane and workgroup of the agent)
  {
    const float* r0 = reinterpret_cast<const float*>(sPn + 0 * PANEL_B + 2 * PLANE * 2);
    const float* r1 = reinterpret_cast<const float*>(sPn + 1 * PANEL_B + 2 * PLANE * 2);
    const float* r2 = reinterpret_cast<const float*>(sPn + 2 * PANEL_B + 2 * PLANE * 2);
    const float* r3 = reinterpret_cast<const float*>(sPn + 3 * PANEL_B + 2 * PLANE * 2);
    float* out = partials + (((long)s * N + i) * gridDim.x + blockIdx.x) * PT::SIZE;          // the workgroup's record
    out[r] = (r0[r] + r1[r]) + (r2[r] + r3[r]);
    if (r + ROWS < PT::SIZE) out[r + ROWS] = (r0[r + ROWS] + r1[r + ROWS]) + (r2[r + ROWS] + r3[r + ROWS]);
  }
}

// theta(small arrays) -= lr * sum_chunks partial; optional loss_out[s][n] = sum(diff^2)/B.
template <int HID>
__global__ __launch_bounds__(256) void k_small_sgd(const float* __restrict__ partials, float* __restrict__ theta,
                                                   const int* __restrict__ mask,
                                                   float* __restrict__ loss_out, int N, int B, int in_dim, int ldp,
                                                   int nchunk, float lr) {
  typedef FitPart<HID> PT;
  const int s = blockIdx.y, i = blockIdx.x;
  if (mask && !mask[i]) return;
  const NetGeom g = make_geom(in_dim, HID, 1);
  float* th = theta + ((long)s * N + i) * ldp;
  const float* pp = partials + ((long)s * N + i) * nchunk * PT::SIZE;
  for (int e = threadIdx.x; e < PT::SIZE; e += blockDim.x) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += pp[(long)c * PT::SIZE + e];
    if (e == PT::loss) {
      if (loss_out) loss_out[(long)s * N + i] = sum / (float)B;
      continue;
    }
    int o;
    if (e < PT::gb2) o = g.o_W2 + e;
    else if (e < PT::gW3) o = g.o_b2 + (e - PT::gb2);
    else if (e < PT::gb3) o = g.o_W3 + (e - PT::gW3);
    else if (e < PT::gb1) o = g.o_b3;
    else o = g.o_b1 + (e - PT::gb1);
    th[o] = th[o] - lr * sum;
  }
}

// ---------------------------------------------------------------------------------------------
// out[s][n][b] = head(a1)                    (r_applied == nullptr)
//             = r_applied[s][n][b] + gamma*v (TD target, agents/resilient_CAC_agents.py:115)
template <int HID>
__global__ __launch_bounds__(256) void k_mid_value(const float* __restrict__ a1t, const float* __restrict__ theta,
                                                   const float* __restrict__ r_applied, float gamma,
                                                   float* __restrict__ out, int N, int B, int in_dim, int ldp,
                                                   int ldb) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int b = blockIdx.x * ROWS + threadIdx.x;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a1[HID], a2[HID];
  load_a1<HID>(a1t, ((long)s * N + i) * HID, ldb, b, valid, a1);
  layer2<HID>(th, g, a1, a2);
  const float v = head1<HID>(th + g.o_W3, th[g.o_b3], a2);
  if (valid) {
    const long o = ((long)s * N + i) * ldb + b;
    out[o] = r_applied ? r_applied[o] + gamma * v : v;
  }
}

// ---------------------------------------------------------------------------------------------
// K2+K3.  For cooperative agent i: phi = features_{theta_i}(x); V_k = phi . W3(msg[nbr[i][k]]) + b3;
// agg = resilient aggregate over k; residual e = (agg - V_live)/(|phi|^2+1);
// partial[chunk] = [sum_b e*phi (HID) | sum_b e]
template <int HID, int D, int H>
__device__ __forceinline__ float select_agg(const float (&v)[D]) {
  float lo, hi;
  SelNet<D, H>::run(v, lo, hi);
  const float lower = fminf(lo, v[0]), upper = fmaxf(hi, v[0]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) sum += __builtin_amdgcn_fmed3f(v[k], lower, upper);
  return sum / (float)D;
}

template <int HID, int D, int H>
__global__ __launch_bounds__(256) void k_consensus_head(const float* __restrict__ a1t, const float* __restrict__ theta,
                                                        const float* __restrict__ msg, const int* __restrict__ nbr,
                                                        const int* __restrict__ coop,
                                                        float* __restrict__ partials, float* __restrict__ agg_out,
                                                        int N, int B, int in_dim, int ldp, int ldb, int nchunk) {
  __shared__ float red[4 * (HID + 1)];
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  if (!coop[i]) return;                       // workgroup-uniform
  const int b = chunk * ROWS + threadIdx.x;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a1[HID], phi[HID];
  load_a1<HID>(a1t, ((long)s * N + i) * HID, ldb, b, valid, a1);
  layer2<HID>(th, g, a1, phi);
  float nrm = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) nrm = fmaf(phi[k], phi[k], nrm);
  nrm += 1.0f;
  float v[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float* mh = msg + ((long)s * N + nbr[i * D + k]) * ldp;
    v[k] = head1<HID>(mh + g.o_W3, mh[g.o_b3], phi);
  }
  const float agg = select_agg<HID, D, H>(v);
  const float v_live = head1<HID>(th + g.o_W3, th[g.o_b3], phi);
  const float e = valid ? (agg - v_live) / nrm : 0.f;
  if (agg_out && valid) agg_out[((long)s * N + i) * ldb + b] = agg;
  float* out = partials + (((long)s * N + i) * nchunk + chunk) * (HID + 1);
  float proj[HID + 1];
#pragma unroll
  for (int k = 0; k < HID; ++k) proj[k] = e * phi[k];
  proj[HID] = e;
  block_reduce_store<HID + 1>(proj, red, out);
}

// ---------------------------------------------------------------------------------------------
// K2+K3 with layer 2 AND the d + 1 heads on the f16 matrix core (round 6; 20 units, d + 1 <= 32 heads).
// k_consensus_head above spends ~1270 vector instructions per 64 replay rows, 60 % of them the d + 1 length-20 dot products of the
// heads and a third the 20 x 20 layer (profiles/r05_sq_k_consensus_head_18_8.json: VALU 100 % busy, SQ_INSTS_MFMA = 0).  Here both
// run as v_mfma_f32_32x32x16_f16 on two-piece f16 operands in the lane layout of k_mid_fit_v8 (rcmarl_lattice.h: lane (row j, half h)
// holds ten of the row's 20 units -- the B-operand shape of a contraction over units):
//   z2[unit][row]  = sum_m W2[m][unit] a1[row][m]     A = 2^10 W2^T (rows / slots permuted as in v8),  B = a1 pieces
//   est[head][row] = sum_u W3_head[u] phi[row][u]     A = 2^10 [W3 of msg[nbr[i][0..d-1]] | W3 live],  B = phi pieces
// The estimates come out with the 32 heads of a row spread over its two lanes (lane (j, h): heads 8q + 4h + e).  A wavefront works on
// TWO blocks of 32 rows; one v_permlane32_swap per accumulator register pair hands lanes 0-31 the other half of block 0's heads and
// lanes 32-63 the other half of block 1's: afterwards lane l holds ALL heads of replay row l of the wavefront's 64, in registers, and
// the selection network, the clamp, the mean and the projection residual run one row per lane as before.
// Range: weights beyond the f16 range of 2^10 w (|w| > 63) or activations beyond 65000 send the wavefront to the fp32 lane code of
// k_consensus_head for its 64 rows (same launch).
// Scales (round 6, third form): the layer-2 table is 2^6 W2^T and the head table 2^10 W3, the biases ride in two spare contraction
// slots of the second k-step (units 10 / 11 of half 0: two pieces of the scaled bias + two of what those left; the B operand carries a
// constant there: 1.0 for layer 2, 2^6 for the heads), so the accumulators ARE 2^6 z2 and 2^16 est and no instruction rescales them:
// LeakyReLU, the order statistics, the clamp and the mean commute with a power of two (every fp32 operation on 2^k x returns 2^k times
// its result on x), |phi|^2 is rescaled once per row (2^-12) and the residual twice (2^-16, 2^-6 for the projection sums).
#define RC_K2MX_RANGE 65000.f
#define RC_K2MX_S2 64.f
#define RC_K2MX_S3 1024.f
#define RC_K2MX_ONE_H2 0x3C003C00u       // f16 (1.0, 1.0): the two bias slots of layer 2
#define RC_K2MX_ONE_H3 0x54005400u       // f16 (64.0, 64.0): the heads' bias slots (phi arrives as 2^6 phi)
#ifndef RC_K2MX_WAVES
#define RC_K2MX_WAVES 3                  // wavefronts per SIMD the register allocation aims at (161 registers; 4 spills, 2 leaves 192 unused)
#endif
// The product of the two LOW pieces (<= 2^-22 of a term: a quarter of an fp32 rounding step of it) is left out, as in the packed-operand
// GEMMs of the wide critic: three matrix-core passes per k-step instead of four (RC_K2MX_DROP_LL=0 restores it).
#ifndef RC_K2MX_DROP_LL
#define RC_K2MX_DROP_LL RC_V8_DROP_LL
#endif
__device__ __forceinline__ rc_f32x16 k2_mfma(const V8Pieces& a, const V8Pieces& b, rc_f32x16 c) {
  if (!RC_K2MX_DROP_LL) c = rc_mfma_f16(a.l, b.l, c);
  c = rc_mfma_f16(a.l, b.h, c);
  c = rc_mfma_f16(a.h, b.l, c);
  c = rc_mfma_f16(a.h, b.h, c);
  return c;
}
template <int D, int H>
__global__ __launch_bounds__(256, RC_K2MX_WAVES) void k_consensus_head_mx(const float* __restrict__ a1t, const float* __restrict__ theta,
                                                           const float* __restrict__ msg, const int* __restrict__ nbr,
                                                           const int* __restrict__ coop, float* __restrict__ partials,
                                                           float* __restrict__ agg_out, int N, int B, int in_dim, int ldp, int ldb,
                                                           int nchunk, int cpw) {
  constexpr int HID = 20, LU = 10, NH = D + 1, REC = HID + 2;
  static_assert(NH <= 32, "the heads of an agent are the 32 rows of one matrix-core operand");
  __shared__ __attribute__((aligned(16))) uint4 sWf[2 * 2 * 2 * 32];      // layer 2: [k-step][piece][k-group][row i] 16-byte A fragments
  __shared__ __attribute__((aligned(16))) uint4 sHf[2 * 2 * 2 * 32];      // heads:   [k-step][piece][k-group][head]
  __shared__ float red[4 * REC];                                          // per wavefront: sum_b e phi (20) | sum_b e (two halves)
  __shared__ int s_ovf;
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  if (!coop[i]) return;                                                   // workgroup-uniform
  const int r = threadIdx.x, lane = r & 63, wave = __builtin_amdgcn_readfirstlane(r >> 6), l31 = lane & 31, half = lane >> 5;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const long row0 = ((long)s * N + i) * HID;
  rc_f16_saturate();
  if (r == 0) s_ovf = 0;
  __syncthreads();
  {
    unsigned short* wf16 = reinterpret_cast<unsigned short*>(sWf);
    unsigned short* hf16 = reinterpret_cast<unsigned short*>(sHf);
    bool bad = false;
    for (int e = r; e < 2 * 32 * 32; e += ROWS) {
      const int which = e >> 10, ri = (e >> 5) & 31, k = e & 31;
      const int uk = v8_slot_unit(k);
      const bool bias = k == 18 || k == 19;                               // slots of local units 10, 11 of half 0 (second k-step)
      const float sc = which ? RC_K2MX_S3 : RC_K2MX_S2;
      float w = 0.f;
      if (which == 0) {
        const int ui = v8_row_unit(ri);
        if (ui >= 0 && uk >= 0) w = th[g.o_W2 + uk * HID + ui];
        if (ui >= 0 && bias) w = th[g.o_b2 + ui];
      } else if (ri < NH && (uk >= 0 || bias)) {
        const float* src = ri < D ? msg + ((long)s * N + nbr[i * D + ri]) * ldp : th;
        w = bias ? src[g.o_b3] : src[g.o_W3 + uk];
      }
      unsigned ph, pl;
      rc_split2h_pair(w * sc, 0.f, ph, pl);
      if (!(fabsf(w) * sc <= RC_K2MX_RANGE)) bad = true;                  // (NaN weights take the fp32 path too)
      if (bias) {                                                         // slot 18: the two pieces of sc*b; slot 19: the pieces of what they left
        if (k == 19) {
          const float rest = (w * sc - rc_f16_to_f32(ph & 0xffffu)) - rc_f16_to_f32(pl & 0xffffu);
          rc_split2h_pair(rest, 0.f, ph, pl);
        }
      }
      const int ks = k >> 4, kg = (k >> 3) & 1;
      const int base = ((((ks * 2 + 0) * 2 + kg) * 32 + ri) * 8) + (k & 7);     // in f16 elements; piece stride 2*32*8
      unsigned short* dst = which ? hf16 : wf16;
      dst[base] = (unsigned short)ph;
      dst[base + 2 * 32 * 8] = (unsigned short)pl;
    }
    if (bad) s_ovf = 1;
  }
  __syncthreads();
  // The workgroup walks `cpw` chunks of 256 rows (the fragment tables above cost ~300 vector instructions per thread: per chunk
  // they were a third of the kernel, profiles/r06f_sq_k_consensus_head_mx_per_chunk.json); a wavefront keeps its sums over all its
  // rows in registers and reduces them ONCE.  The ten activations of the next chunk's two blocks are requested a chunk ahead.
  const uint4* wfA = sWf + half * 32 + l31;
  const uint4* hfA = sHf + half * 32 + l31;
  auto loadA = [&](const uint4* base, int ks) {
    V8Pieces a;
    a.h = base[((ks * 2 + 0) * 2) * 32];
    a.l = base[((ks * 2 + 1) * 2) * 32];
    return a;
  };
  uint4 z4;
  z4.x = z4.y = z4.z = z4.w = 0u;
  const bool wg_slow = s_ovf != 0;
  float usum[LU], es = 0.f;                                               // sums over this lane's rows: e * phi[unit v8_unit(half, u)], e
#pragma unroll
  for (int u = 0; u < LU; ++u) usum[u] = 0.f;
  unsigned slow_mask = 0u;                                                // wave-uniform: bit c - c_begin = that chunk's 64 rows take the fp32 lane code
  const int c_begin = chunk * cpw, c_end = min(nchunk, c_begin + cpw);
  // The twenty activations of a wavefront's two blocks are requested a chunk ahead, straight into the registers the split of the
  // current chunk has just read (no second register set, no copies).  Addresses as in k_mid_fit_v8: one wave-uniform base per local
  // unit in scalar registers + one 32-bit byte offset per lane, block and unit group.
  const unsigned char* ubase[LU];
#pragma unroll
  for (int u = 0; u < LU; ++u) ubase[u] = reinterpret_cast<const unsigned char*>(a1t + (row0 + (u < 8 ? u : u + 8)) * ldb);
  const unsigned hoff8 = (unsigned)(8 * half) * (unsigned)ldb, hoff2 = (unsigned)(2 * half) * (unsigned)ldb;   // v8_unit(half, u) - v8_unit(0, u) rows
  float a1n[2][LU];
  auto fetch = [&](int c) {
    const int bw = c * ROWS + wave * 64;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const unsigned bc = (unsigned)min(bw + 32 * blk + l31, B - 1);      // (rows beyond B: a clamped read, their e is zero)
      const unsigned o8 = (hoff8 + bc) * 4u, o2 = (hoff2 + bc) * 4u;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        a1n[blk][u] = *reinterpret_cast<const float*>(ubase[u] + (u < 8 ? o8 : o2));
        RC_SCHED_FENCE();
      }
    }
  };
  if (!wg_slow) fetch(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    const int bw = c * ROWS + wave * 64;                                  // the wavefront's first replay row of this chunk
    const int b = bw + lane;                                              // the row this lane finishes (block `half`, row l31 of it)
    const bool valid = b < B;
    float agg = 0.f;
    if (wg_slow) {
      slow_mask |= 1u << (c - c_begin);
      continue;
    }
    {
      float amax = 0.f;
      V8Pieces pa0[2], pa1[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int u = 0; u < LU; u += 2) amax = rc_amax3(amax, a1n[blk][u], a1n[blk][u + 1]);
        const float x0[8] = {a1n[blk][0], a1n[blk][1], a1n[blk][2], a1n[blk][3], a1n[blk][4], a1n[blk][5], a1n[blk][6], a1n[blk][7]};
        pa0[blk] = v8_split8<false>(x0, 1.f);
        pa1[blk].h = z4; pa1[blk].l = z4;
        rc_split2h_pair(a1n[blk][8], a1n[blk][9], pa1[blk].h.x, pa1[blk].l.x);
        pa1[blk].h.y = RC_K2MX_ONE_H2;                                    // the bias slots: b2 sits in the table
      }
      RC_SCHED_FENCE();
      fetch(min(c + 1, c_end - 1));                                       // (the last chunk once more rather than a branch: lesson (ii) of round 4)
      float phi[2][LU], nrmb[2];                                          // phi: 2^6 phi
      float est[2][16];                                                   // 2^16 est
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        rc_f32x16 zz;
#pragma unroll
        for (int q = 0; q < 16; ++q) zz[q] = 0.f;
        RC_MX_BEGIN();
        zz = k2_mfma(loadA(wfA, 0), pa0[blk], zz);
        zz = k2_mfma(loadA(wfA, 1), pa1[blk], zz);                        // (the k-step with the bias last: it joins a finished sum)
        RC_MX_END();
        float np = 0.f;
#pragma unroll
        for (int u = 0; u < LU; ++u) {
          phi[blk][u] = fmaxf(zz[u], RC_LEAK * zz[u]);
          np = fmaf(phi[blk][u], phi[blk][u], np);
        }
#pragma unroll
        for (int u = 0; u < LU; u += 2) amax = rc_amax3(amax, phi[blk][u], phi[blk][u + 1]);
        float na = np, nb = np;
        rc_swap32(na, nb);                                                // na: lanes 32-63 hold the low half's part; nb: lanes 0-31 the high half's
        nrmb[blk] = fmaf(np + (half ? na : nb), 1.0f / (RC_K2MX_S2 * RC_K2MX_S2), 1.0f);
        V8Pieces p0, p1;
        {
          const float x0[8] = {phi[blk][0], phi[blk][1], phi[blk][2], phi[blk][3], phi[blk][4], phi[blk][5], phi[blk][6], phi[blk][7]};
          p0 = v8_split8<false>(x0, 1.f);
          p1.h = z4; p1.l = z4;
          rc_split2h_pair(phi[blk][8], phi[blk][9], p1.h.x, p1.l.x);
          p1.h.y = RC_K2MX_ONE_H3;
        }
        rc_f32x16 ee;
#pragma unroll
        for (int q = 0; q < 16; ++q) ee[q] = 0.f;
        RC_MX_BEGIN();
        ee = k2_mfma(loadA(hfA, 0), p0, ee);
        ee = k2_mfma(loadA(hfA, 1), p1, ee);
        RC_MX_END();
#pragma unroll
        for (int q = 0; q < 16; ++q) est[blk][q] = ee[q];
      }
      if (rc_ballot(!(amax <= RC_K2MX_RANGE)) != 0ull) {           // (a ballot, not rc_any: lane collectives follow, the emulation must decide per WAVE)
        slow_mask |= 1u << (c - c_begin);
        continue;
      }
      {
        // register q of est[0] / est[1]: head 8 (q >> 2) + 4 half + (q & 3) of block 0 / 1.  After the swaps est[0][q] is head
        // 8 (q >> 2) + (q & 3) and est[1][q] head 8 (q >> 2) + 4 + (q & 3) of THIS lane's row (block `half`, row l31).
        constexpr int NQ = 4 * ((NH + 7) / 8) < 16 ? 4 * ((NH + 7) / 8) : 16;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rc_swap32(est[0][q], est[1][q]);
        auto head = [&](int k) { return (k & 4) ? est[1][4 * (k >> 3) + (k & 3)] : est[0][4 * (k >> 3) + (k & 3)]; };
        float v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = head(k);
        constexpr float US23 = 1.0f / (RC_K2MX_S2 * RC_K2MX_S3);
        const float aggs = select_agg<HID, D, H>(v);                      // 2^16 times the aggregate (every step commutes with the scale)
        agg = aggs * US23;
        const float nrm = half ? nrmb[1] : nrmb[0];
        const float e = valid ? ((aggs - head(D)) * US23) / nrm : 0.f;
        const float es6 = e * (1.0f / RC_K2MX_S2);                       // phi is carried as 2^6 phi
        const float eo = __shfl_xor(es6, 32, 64);                         // the residual of the row the partner lane finishes
        const float e0 = half ? eo : es6, e1 = half ? es6 : eo;           // of block 0 / block 1, row l31
#pragma unroll
        for (int u = 0; u < LU; ++u) usum[u] = fmaf(e1, phi[1][u], fmaf(e0, phi[0][u], usum[u]));
        es += e;
      }
    }
    if (agg_out && valid) agg_out[((long)s * N + i) * ldb + b] = agg;
  }
  // ---- one reduction per workgroup: the half-wave sums of the matrix-core rows + what the fp32 lane code collected
  {
    float dummy = 0.f;
    static_assert(LU == 10, "ten unit sums + the residual sum, reduced three at a time");
    rc_half_sum3_lane31(usum[0], usum[1], usum[2]);
    rc_half_sum3_lane31(usum[3], usum[4], usum[5]);
    rc_half_sum3_lane31(usum[6], usum[7], usum[8]);
    rc_half_sum3_lane31(usum[9], es, dummy);
    if (lane < REC) red[wave * REC + lane] = 0.f;
    RC_WAVE_SYNC();
    if (l31 == 31) {
#pragma unroll
      for (int u = 0; u < LU; ++u) red[wave * REC + v8_unit(half, u)] = usum[u];
      red[wave * REC + HID + half] = es;
    }
    RC_WAVE_SYNC();
  }
  // ---- the chunks whose operands left the f16 range (rare; wave-uniform): the fp32 lane code of k_consensus_head for those 64 rows,
  // after the matrix-core loop so that the two never hold their registers at the same time
  if (slow_mask != 0u) {
    RC_NO_SPECULATE();
    for (int c = c_begin; c < c_end; ++c) {
      if (!((slow_mask >> (c - c_begin)) & 1u)) continue;
      const int b = c * ROWS + wave * 64 + lane;
      const bool valid = b < B;
      float a1[HID], phi[HID];
      load_a1<HID>(a1t, row0, ldb, b, valid, a1);
      layer2<HID>(th, g, a1, phi);
      float nrm = 0.f;
#pragma unroll
      for (int k = 0; k < HID; ++k) nrm = fmaf(phi[k], phi[k], nrm);
      nrm += 1.0f;
      float v[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float* mh = msg + ((long)s * N + nbr[i * D + k]) * ldp;
        v[k] = head1<HID>(mh + g.o_W3, mh[g.o_b3], phi);
      }
      const float agg = select_agg<HID, D, H>(v);
      const float e = valid ? (agg - head1<HID>(th + g.o_W3, th[g.o_b3], phi)) / nrm : 0.f;
      if (agg_out && valid) agg_out[((long)s * N + i) * ldb + b] = agg;
#pragma unroll
      for (int k = 0; k <= HID; ++k) {
        const float t = rc_wave_sum_lane63(k < HID ? e * phi[k] : e);
        if (lane == 63) red[wave * REC + k] += t;
      }
    }
  }
  __syncthreads();
  // ONE record for the workgroup's chunks (rcmarl_head_apply sums the records of an agent): the first chunk's slot takes it, the
  // others are zero
  float* out = partials + (((long)s * N + i) * nchunk + c_begin) * (HID + 1);
  if (r < HID) out[r] = (red[r] + red[REC + r]) + (red[2 * REC + r] + red[3 * REC + r]);
  if (r == HID) out[HID] = ((red[HID] + red[HID + 1]) + (red[REC + HID] + red[REC + HID + 1])) +
                           ((red[2 * REC + HID] + red[2 * REC + HID + 1]) + (red[3 * REC + HID] + red[3 * REC + HID + 1]));
  for (int e2 = r; e2 < (c_end - c_begin - 1) * (HID + 1); e2 += ROWS) out[(HID + 1) + e2] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// k_mid_value with layer 2 on the f16 matrix core (round 6, visit t): the value head of a 20-unit net on cached layer-1 activations --
// the TD target of every epoch and the actor phase's three value rows.  k_mid_value spends ~500 vector instructions per 64 rows on the
// 20 x 20 layer; here the layer is k_consensus_head_mx's first half (same table, same bias slots, 2^6 z2 in the accumulators), the head
// is ten FMAs per lane on the scaled activations (W3 pre-divided by 2^6) and one v_permlane32_swap + add joins the two halves of a row.
// Out-of-range weights / activations: the fp32 lane code for those rows, in the same launch (bit-identical to k_mid_value).
template <int HID_>
__global__ __launch_bounds__(256, 4) void k_mid_value_mx(const float* __restrict__ a1t, const float* __restrict__ theta,
                                                         const float* __restrict__ r_applied, float gamma, float* __restrict__ out,
                                                         int N, int B, int in_dim, int ldp, int ldb, int nchunk, int cpw) {
  constexpr int HID = 20, LU = 10;
  static_assert(HID_ == HID, "compiled for 20 units");
  __shared__ __attribute__((aligned(16))) uint4 sWf[2 * 2 * 2 * 32];      // layer 2: [k-step][piece][k-group][row i] 16-byte A fragments
  __shared__ int s_ovf;
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  const int r = threadIdx.x, lane = r & 63, wave = __builtin_amdgcn_readfirstlane(r >> 6), l31 = lane & 31, half = lane >> 5;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const long row0 = ((long)s * N + i) * HID;
  rc_f16_saturate();
  if (r == 0) s_ovf = 0;
  __syncthreads();
  {
    unsigned short* wf16 = reinterpret_cast<unsigned short*>(sWf);
    bool bad = false;
    for (int e = r; e < 32 * 32; e += ROWS) {
      const int ri = (e >> 5) & 31, k = e & 31;
      const int uk = v8_slot_unit(k), ui = v8_row_unit(ri);
      const bool bias = k == 18 || k == 19;
      float w = 0.f;
      if (ui >= 0 && uk >= 0) w = th[g.o_W2 + uk * HID + ui];
      if (ui >= 0 && bias) w = th[g.o_b2 + ui];
      unsigned ph, pl;
      rc_split2h_pair(w * RC_K2MX_S2, 0.f, ph, pl);
      if (!(fabsf(w) * RC_K2MX_S2 <= RC_K2MX_RANGE)) bad = true;
      if (k == 19) {
        const float rest = (w * RC_K2MX_S2 - rc_f16_to_f32(ph & 0xffffu)) - rc_f16_to_f32(pl & 0xffffu);
        rc_split2h_pair(rest, 0.f, ph, pl);
      }
      const int ks = k >> 4, kg = (k >> 3) & 1;
      const int base = ((((ks * 2 + 0) * 2 + kg) * 32 + ri) * 8) + (k & 7);
      wf16[base] = (unsigned short)ph;
      wf16[base + 2 * 32 * 8] = (unsigned short)pl;
    }
    if (bad) s_ovf = 1;
  }
  __syncthreads();
  const uint4* wfA = sWf + half * 32 + l31;
  auto loadA = [&](int ks) {
    V8Pieces a;
    a.h = wfA[((ks * 2 + 0) * 2) * 32];
    a.l = wfA[((ks * 2 + 1) * 2) * 32];
    return a;
  };
  uint4 z4;
  z4.x = z4.y = z4.z = z4.w = 0u;
  const bool wg_slow = s_ovf != 0;
  float w3[LU];                                                            // the lane's ten head weights, for activations carried as 2^6 phi
#pragma unroll
  for (int u = 0; u < LU; ++u) w3[u] = th[g.o_W3 + v8_unit(half, u)] * (1.0f / RC_K2MX_S2);
  const float b3 = th[g.o_b3];
  const int c_begin = chunk * cpw, c_end = min(nchunk, c_begin + cpw);
  const unsigned char* ubase[LU];
#pragma unroll
  for (int u = 0; u < LU; ++u) ubase[u] = reinterpret_cast<const unsigned char*>(a1t + (row0 + (u < 8 ? u : u + 8)) * ldb);
  const unsigned hoff8 = (unsigned)(8 * half) * (unsigned)ldb, hoff2 = (unsigned)(2 * half) * (unsigned)ldb;
  float a1n[2][LU];
  auto fetch = [&](int c) {
    const int bw = c * ROWS + wave * 64;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const unsigned bc = (unsigned)min(bw + 32 * blk + l31, B - 1);
      const unsigned o8 = (hoff8 + bc) * 4u, o2 = (hoff2 + bc) * 4u;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        a1n[blk][u] = *reinterpret_cast<const float*>(ubase[u] + (u < 8 ? o8 : o2));
        RC_SCHED_FENCE();
      }
    }
  };
  unsigned slow_mask = 0u;
  if (!wg_slow) fetch(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = c * ROWS + wave * 64 + lane;                             // the row this lane finishes (block `half`, row l31 of it)
    if (wg_slow) {
      slow_mask |= 1u << (c - c_begin);
      continue;
    }
    float amax = 0.f;
    V8Pieces pa0[2], pa1[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int u = 0; u < LU; u += 2) amax = rc_amax3(amax, a1n[blk][u], a1n[blk][u + 1]);
      const float x0[8] = {a1n[blk][0], a1n[blk][1], a1n[blk][2], a1n[blk][3], a1n[blk][4], a1n[blk][5], a1n[blk][6], a1n[blk][7]};
      pa0[blk] = v8_split8<false>(x0, 1.f);
      pa1[blk].h = z4; pa1[blk].l = z4;
      rc_split2h_pair(a1n[blk][8], a1n[blk][9], pa1[blk].h.x, pa1[blk].l.x);
      pa1[blk].h.y = RC_K2MX_ONE_H2;
    }
    RC_SCHED_FENCE();
    fetch(min(c + 1, c_end - 1));
    float part[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      rc_f32x16 zz;
#pragma unroll
      for (int q = 0; q < 16; ++q) zz[q] = 0.f;
      zz = k2_mfma(loadA(0), pa0[blk], zz);
      zz = k2_mfma(loadA(1), pa1[blk], zz);
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const float phi = fmaxf(zz[u], RC_LEAK * zz[u]);
        acc = fmaf(phi, w3[u], acc);
      }
      part[blk] = acc;
    }
    if (rc_ballot(!(amax <= RC_K2MX_RANGE)) != 0ull) {           // (a ballot, not rc_any: lane collectives follow, the emulation must decide per WAVE)                                // (activations beyond the f16 range; |phi| cannot exceed what the
      slow_mask |= 1u << (c - c_begin);                                    // saturated pieces of a1 and the in-range table allow)
      continue;
    }
    rc_swap32(part[0], part[1]);          // lanes 0-31: both halves' parts of block 0's row l31; lanes 32-63: of block 1's row l31
    const float v = (part[0] + part[1]) + b3;
    if (b < B) {
      const long o = ((long)s * N + i) * ldb + b;
      out[o] = r_applied ? r_applied[o] + gamma * v : v;
    }
  }
  if (slow_mask != 0u) {
    RC_NO_SPECULATE();
    for (int c = c_begin; c < c_end; ++c) {
      if (!((slow_mask >> (c - c_begin)) & 1u)) continue;
      const int b = c * ROWS + wave * 64 + lane;
      const bool valid = b < B;
      float a1[HID], a2[HID];
      load_a1<HID>(a1t, row0, ldb, b, valid, a1);
      layer2<HID>(th, g, a1, a2);
      const float v = head1<HID>(th + g.o_W3, th[g.o_b3], a2);
      if (valid) {
        const long o = ((long)s * N + i) * ldb + b;
        out[o] = r_applied ? r_applied[o] + gamma * v : v;
      }
    }
  }
}

// runtime (d, H) fallback: neighbour estimates staged in LDS, order statistics by rank counting
template <int HID>
__global__ __launch_bounds__(256) void k_consensus_head_generic(
    const float* __restrict__ a1t, const float* __restrict__ theta, const float* __restrict__ msg,
    const int* __restrict__ nbr, const int* __restrict__ coop, float* __restrict__ partials,
    float* __restrict__ agg_out, int N, int B, int in_dim, int ldp, int ldb, int nchunk, int d, int H) {
  __shared__ float red[4 * (HID + 1)];
  RCMARL_DYN_SMEM(float, est);                // [d][ROWS]
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  if (!coop[i]) return;
  const int r = threadIdx.x, b = chunk * ROWS + r;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a1[HID], phi[HID];
  load_a1<HID>(a1t, ((long)s * N + i) * HID, ldb, b, valid, a1);
  layer2<HID>(th, g, a1, phi);
  float nrm = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) nrm = fmaf(phi[k], phi[k], nrm);
  nrm += 1.0f;
  for (int k = 0; k < d; ++k) {
    const float* mh = msg + ((long)s * N + nbr[i * d + k]) * ldp;
    est[k * ROWS + r] = head1<HID>(mh + g.o_W3, mh[g.o_b3], phi);
  }
  const float own = est[r];
  float lo = own, hi = own;
  for (int k = 0; k < d; ++k) {
    const float x = est[k * ROWS + r];
    int rank = 0;
    for (int m = 0; m < d; ++m) {
      const float yv = est[m * ROWS + r];
      rank += (yv < x || (yv == x && m < k)) ? 1 : 0;
    }
    if (rank == H) lo = x;
    if (rank == d - H - 1) hi = x;
  }
  const float lower = fminf(lo, own), upper = fmaxf(hi, own);
  float sum = 0.f;
  for (int k = 0; k < d; ++k) sum += __builtin_amdgcn_fmed3f(est[k * ROWS + r], lower, upper);
  const float agg = sum / (float)d;
  const float v_live = head1<HID>(th + g.o_W3, th[g.o_b3], phi);
  const float e = valid ? (agg - v_live) / nrm : 0.f;
  if (agg_out && valid) agg_out[((long)s * N + i) * ldb + b] = agg;
  float* out = partials + (((long)s * N + i) * nchunk + chunk) * (HID + 1);
  float proj[HID + 1];
#pragma unroll
  for (int k = 0; k < HID; ++k) proj[k] = e * phi[k];
  proj[HID] = e;
  block_reduce_store<HID + 1>(proj, red, out);
}

// K3 alone: projection residual toward a caller-supplied aggregate agg_in[S][N][ldb]
// (critic_update_team(s, agg) / TR_update_team(sa, agg), agents/resilient_CAC_agents.py:60-84);
// same partial records as k_consensus_head.
template <int HID>
__global__ __launch_bounds__(256) void k_projection(const float* __restrict__ a1t, const float* __restrict__ theta,
                                                    const float* __restrict__ agg_in, const int* __restrict__ coop,
                                                    float* __restrict__ partials, int N, int B, int in_dim, int ldp,
                                                    int ldb, int nchunk) {
  __shared__ float red[4 * (HID + 1)];
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  if (!coop[i]) return;
  const int b = chunk * ROWS + threadIdx.x;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a1[HID], phi[HID];
  load_a1<HID>(a1t, ((long)s * N + i) * HID, ldb, b, valid, a1);
  layer2<HID>(th, g, a1, phi);
  float nrm = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) nrm = fmaf(phi[k], phi[k], nrm);
  nrm += 1.0f;
  const float v_live = head1<HID>(th + g.o_W3, th[g.o_b3], phi);
  const float e = valid ? (agg_in[((long)s * N + i) * ldb + b] - v_live) / nrm : 0.f;
  float* out = partials + (((long)s * N + i) * nchunk + chunk) * (HID + 1);
  float proj[HID + 1];
#pragma unroll
  for (int k = 0; k < HID; ++k) proj[k] = e * phi[k];
  proj[HID] = e;
  block_reduce_store<HID + 1>(proj, red, out);
}

// W3 += (1/B) sum_chunks partial[0..HID) ; b3 += (1/B) sum partial[HID]     (cooperative agents)
template <int HID>
__global__ __launch_bounds__(64) void k_head_apply(const float* __restrict__ partials, float* __restrict__ theta,
                                                   const int* __restrict__ coop, int N, int B, int in_dim,
                                                   int ldp, int nchunk) {
  const int s = blockIdx.y, i = blockIdx.x;
  if (!coop[i]) return;
  const NetGeom g = make_geom(in_dim, HID, 1);
  float* th = theta + ((long)s * N + i) * ldp;
  const float* pp = partials + ((long)s * N + i) * nchunk * (HID + 1);
  const int e = threadIdx.x;
  if (e <= HID) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += pp[c * (HID + 1) + e];
    const int o = (e < HID) ? g.o_W3 + e : g.o_b3;
    th[o] = th[o] + sum / (float)B;
  }
}

// ---------------------------------------------------------------------------------------------
// K7 actor: softmax + TD-weighted sparse cross-entropy, backward through layers 3-2.
template <int HID, int A>
__global__ __launch_bounds__(256) void k_mid_actor(float* __restrict__ a1t, const float* __restrict__ theta,
                                                   const float* __restrict__ act_t /*[S][N][ldb] labels*/,
                                                   const float* __restrict__ delta /*[S][N][ldb] sample weights*/,
                                                   float* __restrict__ partials, int N, int B, int in_dim, int ldp,
                                                   int ldb, int nchunk) {
  typedef ActorPart<HID, A> PT;
  __shared__ float sA[HID * LDR];
  __shared__ float sD[HID * LDR];
  __shared__ float red[4 * PT::NSMALL];
  const int s = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
  const int r = threadIdx.x, b = chunk * ROWS + r;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, HID, A);
  const float* th = theta + ((long)s * N + i) * ldp;
  const long row0 = ((long)s * N + i) * HID;
  float a1[HID], a2[HID], dz2[HID], dl[A];
  load_a1<HID>(a1t, row0, ldb, b, valid, a1);
  layer2<HID>(th, g, a1, a2);
  float logit[A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < HID; ++k) acc = fmaf(a2[k], th[g.o_W3 + k * A + a], acc);
    logit[a] = acc + th[g.o_b3 + a];
  }
  float mx = logit[0];
#pragma unroll
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, logit[a]);
  float se = 0.f;
#pragma unroll
  for (int a = 0; a < A; ++a) se += expf(logit[a] - mx);
  const float lse = logf(se);
  const long o = ((long)s * N + i) * ldb + b;
  const int label = valid ? (int)act_t[o] : 0;
  const float w = valid ? delta[o] : 0.f;
  float nll = 0.f;
  const float wB = w / (float)B;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const float logp = (logit[a] - mx) - lse;
    if (a == label) nll = -logp;
    dl[a] = (expf(logp) - (a == label ? 1.f : 0.f)) * wB;
  }
  float* out = partials + (((long)s * N + i) * nchunk + chunk) * PT::SIZE;
  float small[PT::NSMALL];                     // [gb2 | gb3 | gb1 | loss] as in ActorPart
#pragma unroll
  for (int a = 0; a < A; ++a) small[PT::gb3 - PT::gb2 + a] = dl[a];
  small[PT::loss - PT::gb2] = nll * w;
#pragma unroll
  for (int k = 0; k < HID; ++k) {
    float da2 = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) da2 = fmaf(dl[a], th[g.o_W3 + k * A + a], da2);
    dz2[k] = da2 * rc_lrelu_grad_from_act(a2[k]);
    small[k] = dz2[k];
  }
  // phase 1: gW3[k][a] = sum_r a2[r][k]*dl[r][a]
#pragma unroll
  for (int k = 0; k < HID; ++k) sA[k * LDR + r] = a2[k];
#pragma unroll
  for (int a = 0; a < A; ++a) sD[a * LDR + r] = dl[a];
  __syncthreads();
  for (int e = r; e < HID * A; e += ROWS) {
    const int k = e / A, a = e - k * A;
    float acc = 0.f;
    for (int q = 0; q < ROWS; ++q) acc = fmaf(sA[k * LDR + q], sD[a * LDR + q], acc);
    out[PT::gW3 + e] = acc;
  }
  __syncthreads();
  // phase 2: gW2 and dz1
#pragma unroll
  for (int k = 0; k < HID; ++k) {
    sA[k * LDR + r] = a1[k];
    sD[k * LDR + r] = dz2[k];
  }
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    float da1 = 0.f;
#pragma unroll
    for (int k = 0; k < HID; ++k) da1 = fmaf(dz2[k], th[g.o_W2 + j * HID + k], da1);
    const float dz1 = da1 * rc_lrelu_grad_from_act(a1[j]);
    small[PT::gb1 - PT::gb2 + j] = dz1;
    if (valid) a1t[(row0 + j) * ldb + b] = dz1;
  }
  block_reduce_store<PT::NSMALL>(small, red, out + PT::gb2);
  for (int e = r; e < HID * HID; e += ROWS) {
    const int j = e / HID, k = e - j * HID;
    float acc = 0.f;
    for (int q = 0; q < ROWS; ++q) acc = fmaf(sA[j * LDR + q], sD[k * LDR + q], acc);
    out[PT::gW2 + e] = acc;
  }
}

template <int HID, int A>
__global__ __launch_bounds__(256) void k_small_adam(const float* __restrict__ partials, float* __restrict__ theta,
                                                    float* __restrict__ adam_m, float* __restrict__ adam_v,
                                                    const int* __restrict__ mask,
                                                    float* __restrict__ loss_out, int N, int B, int in_dim, int ldp,
                                                    int nchunk, float alpha, float one_m_b1, float one_m_b2,
                                                    float eps) {
  typedef ActorPart<HID, A> PT;
  const int s = blockIdx.y, i = blockIdx.x;
  if (mask && !mask[i]) return;
  const NetGeom g = make_geom(in_dim, HID, A);
  const long ro = ((long)s * N + i) * ldp;
  const float* pp = partials + ((long)s * N + i) * nchunk * PT::SIZE;
  for (int e = threadIdx.x; e < PT::SIZE; e += blockDim.x) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += pp[(long)c * PT::SIZE + e];
    if (e == PT::loss) {
      if (loss_out) loss_out[(long)s * N + i] = sum / (float)B;
      continue;
    }
    int o;
    if (e < PT::gW3) o = g.o_W2 + e;
    else if (e < PT::gb2) o = g.o_W3 + (e - PT::gW3);
    else if (e < PT::gb3) o = g.o_b2 + (e - PT::gb2);
    else if (e < PT::gb1) o = g.o_b3 + (e - PT::gb3);
    else o = g.o_b1 + (e - PT::gb1);
    float mm = adam_m[ro + o], vv = adam_v[ro + o];
    mm += (sum - mm) * one_m_b1;
    vv += (sum * sum - vv) * one_m_b2;
    adam_m[ro + o] = mm; adam_v[ro + o] = vv;
    theta[ro + o] = theta[ro + o] - (mm * alpha) / (sqrtf(vv) + eps);
  }
}

// ---------------------------------------------------------------------------------------------
// K6 helpers
// r_coop[s][b] = sum over cooperative agents (in index order) of r[s][b][n]/n_coop   (train_agents.py:96-98)
__global__ __launch_bounds__(256) void k_team_reward(const float* __restrict__ r, long seed_stride,
                                                     const int* __restrict__ coop, int n_coop,
                                                     float* __restrict__ rcoop, int N, int B, int ldb) {
  const int s = blockIdx.y, b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* row = r + (long)s * seed_stride + (long)b * N;
  float acc = 0.f;
  for (int n = 0; n < N; ++n)
    if (coop[n]) acc += row[n] / (float)n_coop;
  rcoop[(long)s * ldb + b] = acc;
}

// agent-major gather: out[s][n][b] = mode[n]==0 ? src[s][b][n] : (mode[n]==1 ? rcoop[s][b] : -rcoop[s][b])
__global__ __launch_bounds__(256) void k_gather_agent_major(const float* __restrict__ src, long seed_stride,
                                                            const float* __restrict__ rcoop,
                                                            const int* __restrict__ mode, float* __restrict__ out,
                                                            int N, int B, int ldb) {
  const int s = blockIdx.z, n = blockIdx.y, b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int m = mode ? mode[n] : 0;
  float v;
  if (m == 0) v = src[(long)s * seed_stride + (long)b * N + n];
  else v = (m == 1) ? rcoop[(long)s * ldb + b] : -rcoop[(long)s * ldb + b];
  out[((long)s * N + n) * ldb + b] = v;
}

// every `step`-th replay row, starting at `first`, gathered into a dense block: dst[s][k][c] = src[s][first + k*step][c]
// (the last next-state row of every episode: the only rows of a TD target that need a forward pass of their own)
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, long seed_stride, int first, int step,
                                                     int n_rows, int width, float* __restrict__ dst) {
  const int s = blockIdx.z, k = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < width) dst[((long)s * n_rows + k) * width + c] = src[(long)s * seed_stride + (long)(first + k * step) * width + c];
}

// ... and their values scattered back: out[s][n][first + k*step] = v[s][n][k]  or  r_applied[s][n][first + k*step] + gamma * v
// (multiply, then add: the two roundings of k_mid_value's TD target)
__global__ __launch_bounds__(256) void k_scatter_values(const float* __restrict__ v, const float* __restrict__ r_applied, float gamma,
                                                        float* __restrict__ out, int first, int step, int n_rows, int ldb) {
  const long row = blockIdx.y;                            // (seed, agent)
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rows) return;
  const int b = first + k * step;
  const float val = v[row * ldb + k];
  if (r_applied == nullptr) {
    out[row * ldb + b] = val;
  } else {
    const float gv = gamma * val;
    out[row * ldb + b] = r_applied[row * ldb + b] + gv;
  }
}

// delta = r_team + gamma*V(ns) - V(s)            (agents/resilient_CAC_agents.py:98)
__global__ __launch_bounds__(256) void k_td_error(const float* __restrict__ r_team, const float* __restrict__ v_next,
                                                  const float* __restrict__ v_cur, float gamma,
                                                  float* __restrict__ delta, long n_total) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_total) delta[t] = r_team[t] + gamma * v_next[t] - v_cur[t];
}

// chunks of 256 rows one mid-fit workgroup walks (the agent's weights are staged once per workgroup -- ~4 us of mostly latency:
// 3-, 4-, 6-, 12-chunk workgroups ran 517, 485, 458, 438 us at BASELINE configs[3], profiles/r04w_mid_cpw.txt): ALL of an agent's
// chunks (at most 16) when the S*N workgroup columns alone fill the chip eight times over; else half of them, at most 6; fewer
// -- down to one -- when even that leaves the chip short of workgroups (a single instance of the reference's 5-agent scenario is
// 5 columns: 10 workgroups of 6 chunks each were 37 us of latency)
int midfit_cpw(int nchunk, long columns) {
  if (columns >= 2048) return nchunk < 16 ? nchunk : 16;
  int c = (nchunk + 1) / 2;
  if (c > 6) c = 6;
  while (c > 1 && columns * ((nchunk + c - 1) / c) < 512) --c;
  return c < 1 ? 1 : c;
}

// rcmarl_mid_fit_lattice with the backward operand as f16 pieces (RCMARL_LAT_F16 bit 1, the default): k_mid_fit_v8 (f16 matrix-core
// form) followed by a fix-up launch of k_mid_fit_v5 for the agents v8 flagged as out of range; RCMARL_MIDFIT=5 forces v5 alone.
// Everything else -- rcmarl_mid_fit (fp32 dz1 in place: networks off the lattice path), three-piece bf16 operands -- is v5.
// Read at every call (tests switch it inside one process).
bool midfit_v8() {
  const char* e = getenv("RCMARL_MIDFIT");
  return (rc_lat_f16_mode() & 2) && !(e && atoi(e) == 5);
}

// Out-of-range flags of k_mid_fit_v8: one int per (seed, agent), owned by the library (the C-ABI hands no workspace over), zeroed at
// allocation.  A launch pair marks and reads them with a GENERATION number that lives on the device beside them and is bumped by a
// one-thread kernel in front of every pair: nothing is ever cleared, and a pair replayed from a captured hipGraph gets a fresh
// number like an eager one (a number passed as a kernel argument would be frozen into the graph).  Allocated at the first call
// (65536 entries cover every BASELINE shape; a larger S*N reallocates) -- calls are expected from one host thread and one stream at a
// time, as the engine issues them.
__global__ void k_bump_generation(int* g) { *g = *g >= (1 << 30) ? 1 : *g + 1; }
bool bad_mid(const void* a, const void* b, int S, int N, int B, int in_dim, int hid, int ldp, int ldb) {
  return !a || !b || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || hid <= 0 || (ldp & 63) || (ldb & 63) || ldb < B;
}

}  // namespace

#define RC_HID_SWITCH(hid, STMT)               \
  switch (hid) {                               \
    case 20: { constexpr int HID_ = 20; STMT; } break; \
    default: return RCMARL_ERR_UNSUPPORTED;    \
  }

RCMARL_EXPORT int rcmarl_fit_partial_size(int hid) { return hid * hid + 3 * hid + 2; }
RCMARL_EXPORT int rcmarl_actor_partial_size(int hid, int n_actions) {
  return hid * hid + 2 * hid + hid * n_actions + n_actions + 1;
}
RCMARL_EXPORT int rcmarl_rows_per_chunk(void) { return ROWS; }

RCMARL_EXPORT int rcmarl_mid_fit(float* a1t, const float* theta, const float* y, float* partials, int S, int N, int B,
                                 int in_dim, int hid, int ldp, int ldb, void* stream) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !y || !partials) return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS), cpw = midfit_cpw(nchunk, (long)S * N);
  const dim3 grid(rc_ceil_div(nchunk, cpw), N, S), block(ROWS);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_fit_v5<HID_, false>), grid, block, 0, stream, a1t, theta, y, partials, N, B, in_dim,
                                   ldp, ldb, nchunk, cpw, (unsigned char*)nullptr, 0, 0, (const int*)nullptr, (const int*)nullptr));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_mid_fit_lattice(const float* a1t, const float* theta, const float* y, float* partials,
                                         void* dzp, int dzp_rt, int dzp_kt, int S, int N, int B, int in_dim, int hid,
                                         int ldp, int ldb, int* ovf_flags, void* stream) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !y || !partials || !dzp) return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS), cpw = midfit_cpw(nchunk, (long)S * N);
  if (dzp_rt * 128 < N * hid || dzp_kt * 32 < nchunk * ROWS) return RCMARL_ERR_ARG;   // every lane of every chunk stores
  const dim3 grid(rc_ceil_div(nchunk, cpw), N, S), block(ROWS);
  rc_form_set(dzp, (rc_lat_f16_mode() >> 1) & 1);
  if (midfit_v8() && ovf_flags != nullptr) {               // (no flag buffer: the fp32-arithmetic kernel alone)
    // ovf_flags[0 .. S*N) = the generation in which an agent was last flagged, ovf_flags[S*N] = the generation counter,
    // bumped on the device so that a launch pair replayed from a hipGraph draws a fresh one
    int* flags = ovf_flags;
    int* gen = ovf_flags + (size_t)S * N;
    RCMARL_LAUNCH(k_bump_generation, dim3(1), dim3(1), 0, stream, gen);
    RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_fit_v8<HID_, true>), grid, block, 0, stream, const_cast<float*>(a1t), theta, y,
                                     partials, N, B, in_dim, ldp, ldb, nchunk, cpw, (unsigned char*)dzp, dzp_rt, dzp_kt, flags, (const int*)gen));
    // the fix-up: same grid, same records, same packed rows -- a workgroup whose agent is not flagged returns at once
    RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_fit_v5<HID_, true, true>), grid, block, 0, stream, const_cast<float*>(a1t), theta, y,
                                     partials, N, B, in_dim, ldp, ldb, nchunk, cpw, (unsigned char*)dzp, dzp_rt, dzp_kt,
                                     (const int*)flags, (const int*)gen));
  } else if (rc_lat_f16_mode() & 2) {
    RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_fit_v5<HID_, true, true>), grid, block, 0, stream, const_cast<float*>(a1t), theta, y,
                                     partials, N, B, in_dim, ldp, ldb, nchunk, cpw, (unsigned char*)dzp, dzp_rt, dzp_kt,
                                     (const int*)nullptr, (const int*)nullptr));
  } else {
    RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_fit_v5<HID_, true>), grid, block, 0, stream, const_cast<float*>(a1t), theta, y,
                                     partials, N, B, in_dim, ldp, ldb, nchunk, cpw, (unsigned char*)dzp, dzp_rt, dzp_kt,
                                     (const int*)nullptr, (const int*)nullptr));
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_small_sgd(const float* partials, float* theta, const int* mask, float* loss_out,
                                   int S, int N, int B, int in_dim, int hid, int ldp, float lr, void* stream) {
  if (!partials || !theta || S <= 0 || N <= 0 || B <= 0) return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS), nrec = rc_ceil_div(nchunk, midfit_cpw(nchunk, (long)S * N));   // = rcmarl_mid_fit's grid
  const dim3 grid(N, S), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_small_sgd<HID_>), grid, block, 0, stream, partials, theta, mask, loss_out, N, B,
                                   in_dim, ldp, nrec, lr));
  return rcmarl_check_launch();
}

static int mid_value_impl(const float* a1t, const float* theta, const float* r_applied, float gamma, float* out, int S, int N, int B,
                          int in_dim, int hid, int ldp, int ldb, void* stream, bool f32_only) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !out) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(B, ROWS), N, S), block(ROWS);
  // RCMARL_MIDVALUE_MX (default 1): layer 2 on the f16 matrix core (k_mid_value_mx; 20 units, a two-piece operand form selected)
  const char* mxe = getenv("RCMARL_MIDVALUE_MX");
  if (!f32_only && hid == 20 && !(mxe && atoi(mxe) == 0) && rc_lat_f16_mode() != 0) {
    const int nchunk = rc_ceil_div(B, ROWS);
    const long pairs = (long)N * S;
    int wgs = pairs >= 2048 ? 2 : (pairs >= 512 ? 4 : nchunk);
    if (wgs * 32 < nchunk) wgs = (nchunk + 31) / 32;                       // (a wavefront keeps one bit per chunk)
    const int cpw = (nchunk + wgs - 1) / wgs;
    const dim3 gmx((unsigned)((nchunk + cpw - 1) / cpw), N, S);
    RCMARL_LAUNCH((k_mid_value_mx<20>), gmx, block, 0, stream, a1t, theta, r_applied, gamma, out, N, B, in_dim, ldp, ldb, nchunk, cpw);
    return rcmarl_check_launch();
  }
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_value<HID_>), grid, block, 0, stream, a1t, theta, r_applied, gamma, out, N,
                                   B, in_dim, ldp, ldb));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_mid_value(const float* a1t, const float* theta, const float* r_applied, float gamma,
                                   float* out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                                   void* stream) {
  return mid_value_impl(a1t, theta, r_applied, gamma, out, S, N, B, in_dim, hid, ldp, ldb, stream, false);
}

RCMARL_EXPORT int rcmarl_mid_value_f32(const float* a1t, const float* theta, const float* r_applied, float gamma,
                                       float* out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                                       void* stream) {
  return mid_value_impl(a1t, theta, r_applied, gamma, out, S, N, B, in_dim, hid, ldp, ldb, stream, true);
}

template <int DD, int HH>
static bool launch_consensus_head_mx(dim3 grid, dim3 block, void* stream, const float* a1t, const float* theta, const float* msg,
                                     const int* nbr, const int* coop, float* partials, float* agg_out, int N, int B, int in_dim,
                                     int ldp, int ldb, int nchunk) {
  if constexpr (DD + 1 <= 32) {
    // chunks per workgroup: the whole agent when there are enough (seed, agent) pairs to fill the GPU, else fewer
    const long pairs = (long)grid.y * grid.z;
    int wgs = pairs >= 2048 ? 2 : (pairs >= 512 ? 4 : nchunk);
    if (wgs * 32 < nchunk) wgs = (nchunk + 31) / 32;                      // (a wavefront keeps one bit per chunk)
    int cpw = (nchunk + wgs - 1) / wgs;
    const char* ce = getenv("RCMARL_K2_CPW");                             // (tests: chunks per workgroup, 1..32)
    if (ce && atoi(ce) >= 1 && atoi(ce) <= 32) cpw = atoi(ce);
    grid.x = (unsigned)((nchunk + cpw - 1) / cpw);
    RCMARL_LAUNCH((k_consensus_head_mx<DD, HH>), grid, block, 0, stream, a1t, theta, msg, nbr, coop, partials, agg_out, N, B, in_dim,
                  ldp, ldb, nchunk, cpw);
    return true;
  } else {
    return false;
  }
}

RCMARL_EXPORT int rcmarl_consensus_head(const float* a1t, const float* theta, const float* msg, const int* nbr,
                                        const int* coop, float* partials, float* agg_out, int S, int N,
                                        int B, int in_dim, int hid, int ldp, int ldb, int d, int H, void* stream) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !msg || !nbr || !coop || !partials || d <= 0 || H < 0 ||
      d < 2 * H + 1)
    return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS);
  const dim3 grid(nchunk, N, S), block(ROWS);
  bool done = false;
  // RCMARL_K2_MX (default 1): layer 2 and the d + 1 heads on the f16 matrix core (k_consensus_head_mx: 20 units, d + 1 <= 32 heads, a
  // generated selection network); 0: everything on the vector ALUs (k_consensus_head)
  // ... and in the EXACT operand form of the lattice path (RCMARL_LAT_F16 = 0: no operand narrower than fp32 anywhere)
  const char* mxe = getenv("RCMARL_K2_MX");
  const bool mx = hid == 20 && d + 1 <= 32 && !(mxe && atoi(mxe) == 0) && rc_lat_f16_mode() != 0;
#define RC_CASE(DD, HH)                                                                                          \
  if (!done && d == DD && H == HH) {                                                                             \
    if (mx) done = launch_consensus_head_mx<DD, HH>(grid, block, stream, a1t, theta, msg, nbr, coop, partials, agg_out, N, B, in_dim,  \
                                                    ldp, ldb, nchunk);                                           \
    if (!done) {                                                                                                 \
      RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_consensus_head<HID_, DD, HH>), grid, block, 0, stream, a1t, theta, msg,  \
                                       nbr, coop, partials, agg_out, N, B, in_dim, ldp, ldb, nchunk));          \
      done = true;                                                                                               \
    }                                                                                                            \
  }
  RCMARL_SELNET_COMBOS(RC_CASE)
#undef RC_CASE
  if (!done) {
    const size_t smem = (size_t)d * ROWS * sizeof(float);
    if (smem > 60 * 1024) return RCMARL_ERR_UNSUPPORTED;
    RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_consensus_head_generic<HID_>), grid, block, smem, stream, a1t, theta, msg, nbr,
                                     coop, partials, agg_out, N, B, in_dim, ldp, ldb, nchunk, d, H));
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_projection_residual(const float* a1t, const float* theta, const float* agg, const int* coop,
                                             float* partials, int S, int N, int B, int in_dim, int hid, int ldp,
                                             int ldb, void* stream) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !agg || !coop || !partials) return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS);
  const dim3 grid(nchunk, N, S), block(ROWS);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_projection<HID_>), grid, block, 0, stream, a1t, theta, agg, coop, partials, N, B,
                                   in_dim, ldp, ldb, nchunk));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_head_apply(const float* partials, float* theta, const int* coop, int S, int N,
                                    int B, int in_dim, int hid, int ldp, void* stream) {
  if (!partials || !theta || !coop || S <= 0 || N <= 0 || B <= 0 || hid >= 64) return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, ROWS);
  const dim3 grid(N, S), block(64);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_head_apply<HID_>), grid, block, 0, stream, partials, theta, coop, N, B, in_dim,
                                   ldp, nchunk));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_mid_actor(float* a1t, const float* theta, const float* act_t, const float* delta,
                                   float* partials, int S, int N, int B, int in_dim, int hid, int n_actions, int ldp,
                                   int ldb, void* stream) {
  if (bad_mid(a1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !act_t || !delta || !partials) return RCMARL_ERR_ARG;
  if (n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  const int nchunk = rc_ceil_div(B, ROWS);
  const dim3 grid(nchunk, N, S), block(ROWS);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_mid_actor<HID_, 5>), grid, block, 0, stream, a1t, theta, act_t, delta, partials,
                                   N, B, in_dim, ldp, ldb, nchunk));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_small_adam(const float* partials, float* theta, float* adam_m, float* adam_v,
                                    const int* mask, float* loss_out, int S, int N, int B, int in_dim,
                                    int hid, int n_actions, int ldp, float alpha, float one_m_b1, float one_m_b2,
                                    float eps, void* stream) {
  if (!partials || !theta || !adam_m || !adam_v || S <= 0 || N <= 0 || B <= 0) return RCMARL_ERR_ARG;
  if (n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  const int nchunk = rc_ceil_div(B, ROWS);
  const dim3 grid(N, S), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_small_adam<HID_, 5>), grid, block, 0, stream, partials, theta, adam_m, adam_v,
                                   mask, loss_out, N, B, in_dim, ldp, nchunk, alpha, one_m_b1, one_m_b2, eps));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_team_reward(const float* r, long seed_stride, const int* coop, int n_coop,
                                     float* rcoop, int S, int N, int B, int ldb, void* stream) {
  if (!r || !coop || !rcoop || S <= 0 || N <= 0 || B <= 0 || n_coop <= 0 || ldb < B) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(B, 256), S), block(256);
  RCMARL_LAUNCH(k_team_reward, grid, block, 0, stream, r, seed_stride, coop, n_coop, rcoop, N, B, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_gather_agent_major(const float* src, long seed_stride, const float* rcoop, const int* mode,
                                            float* out, int S, int N, int B, int ldb, void* stream) {
  if (!src || !out || S <= 0 || N <= 0 || B <= 0 || ldb < B || (mode && !rcoop)) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(B, 256), N, S), block(256);
  RCMARL_LAUNCH(k_gather_agent_major, grid, block, 0, stream, src, seed_stride, rcoop, mode, out, N, B, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_gather_rows(const float* src, long seed_stride, int first, int step, int n_rows, int width, float* dst,
                                     int S, void* stream) {
  if (!src || !dst || S <= 0 || n_rows <= 0 || width <= 0 || first < 0 || step <= 0) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_gather_rows, dim3(rc_ceil_div(width, 256), n_rows, S), dim3(256), 0, stream, src, seed_stride, first, step, n_rows,
                width, dst);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_scatter_values(const float* v, const float* r_applied, float gamma, float* out, int first, int step,
                                        int n_rows, int S, int N, int ldb, void* stream) {
  if (!v || !out || S <= 0 || N <= 0 || n_rows <= 0 || first < 0 || step <= 0 || first + (n_rows - 1) * step >= ldb) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_scatter_values, dim3(rc_ceil_div(n_rows, 256), S * N), dim3(256), 0, stream, v, r_applied, gamma, out, first, step,
                n_rows, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_td_error(const float* r_team, const float* v_next, const float* v_cur, float gamma,
                                  float* delta, long n_total, void* stream) {
  if (!r_team || !v_next || !v_cur || !delta || n_total <= 0) return RCMARL_ERR_ARG;
  const dim3 grid((unsigned)((n_total + 255) / 256)), block(256);
  RCMARL_LAUNCH(k_td_error, grid, block, 0, stream, r_team, v_next, v_cur, gamma, delta, n_total);
  return rcmarl_check_launch();
}
