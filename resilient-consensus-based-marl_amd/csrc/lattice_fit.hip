// One launch per SGD step of the cooperative agents' local fit (agents/resilient_CAC_agents.py:118,136: Keras
// fit(batch_size=B, epochs=5) of a 20-unit critic / team-reward net), forward AND backward down to dz1:
//
//   rcmarl_fit_fused_lattice = rcmarl_layer1_forward_lattice + rcmarl_mid_fit_lattice
//
// with the layer-1 activations never leaving the chip.  The unfused pair writes a1t (fp32, 80 B per (agent, replay
// row)) from the GEMM and reads it back in the mid kernel: 2 GB of HBM traffic per step at BASELINE configs[3].
//
// Layout that makes it possible (rcmarl_lattice.h, "fit order"): the A operand W' (three exact bf16 pieces of
// alpha_k*W1, rows = (agent, unit) columns) is stored with its rows permuted so that, in the 32x32 MFMA accumulator
// layout, lane (n, h) of a wavefront ends the k-loop holding ALL 20 layer-1 pre-activations of agents 4h..4h+3 of the
// workgroup's 8-agent group on its replay row -- in registers, at compile-time indices.  Layers 2-3, the MSE gradient
// and the way back to dz1 then are per-lane fmaf chains (two units per v_pk_fma_f32, weights from broadcast
// ds_read_b128), identical in order to k_mid_fit_v3, so dz1 is bit-identical to the unfused path.
//
// Workgroup = 8 wavefronts = 160 W' rows (8 agents) x 256 replay rows; wavefront = 160 x 32 (five accumulators, 80
// registers -- a 160 x 64 wavefront tile was tried first: 160 accumulator registers + the epilogue's live values do
// not fit 256 and hipcc spilled 1400 dwords).  k-loop: A through two 30-KiB LDS stages filled by LDS-DMA (shared by
// the 8 wavefronts); the B operand (K, one piece) is NOT shared between wavefronts (each owns its 32 rows), so its
// fragments go straight from L2 into registers, one k16 step ahead.
// Epilogue per agent pair: panels [a1 | 1]^T [dz2] -> gW2, gb2 on the f32 matrix core (exact fp32 products, the
// contraction over the wavefront's 32 rows), the 42 plain row sums by fused-DPP adds, the eight wavefronts' records
// summed through LDS in a fixed order (deterministic, no atomics); dz1 leaves as three exact bf16 pieces in the
// packed layout rcmarl_layer1_backward_sgd_lattice reads.
#include "rcmarl_lattice.h"
#include <stdlib.h>

namespace {

constexpr int HID = 20, H2 = HID / 2;
constexpr int FAG = RC_FIT_AGENTS;            // agents per workgroup
constexpr int FBM = RC_FIT_ROWS;              // W' rows per workgroup
constexpr int FMB = FBM / 32;                 // 32-row MFMA blocks
constexpr int NW = 8;                         // wavefronts per workgroup
constexpr int FWN = 32;                       // replay rows per wavefront (one MFMA column block)
constexpr int FBN = NW * FWN;                 // replay rows per workgroup = the chunk of a partial record
constexpr int A_PIECE = FBM * 64;             // bytes of one piece of one k32 stage
constexpr int A_STAGE = 3 * A_PIECE;          // 30720
constexpr int A_BURSTS = A_STAGE / 1024;      // 30 LDS-DMA bursts of 1 KiB per stage
// record layout = FitPart<20> of mid_kernels.hip
constexpr int P_GW2 = 0, P_GB2 = HID * HID, P_GW3 = P_GB2 + HID, P_GB3 = P_GW3 + HID, P_GB1 = P_GB3 + 1,
              P_LOSS = P_GB1 + HID, P_SIZE = P_LOSS + 1;                  // 462 floats
// epilogue LDS (aliases the k-loop stages): weights of the 8 agents | per-wavefront panels | staging area
constexpr int W_B1 = 0, W_W2 = HID, W_B2 = W_W2 + HID * HID, W_W3 = W_B2 + HID, W_B3 = W_W3 + HID, W_W2T = 464;
constexpr int WS = W_W2T + HID * HID;         // floats per agent: b1(20) | W2(400) | b2(20) | W3(20) | b3 | pad | W2^T(400)
constexpr int PLD = FWN + 1;                  // panel row stride (floats): (unit*33 + row) % 32 distinct over units
constexpr int PA_ROWS = HID + 2, PB_ROWS = HID + 1;                      // a1 | ones | zeros ;  dz2 | zeros
constexpr int PANEL = ((PA_ROWS + PB_ROWS) * PLD + 63) / 64 * 64;        // floats per wavefront
constexpr int LDS_WTS = 0, LDS_PANELS = FAG * WS * 4, LDS_STAGING = LDS_PANELS + NW * PANEL * 4;
static_assert(LDS_STAGING >= 2 * A_STAGE, "the k-loop stages lie inside the weights + panels area");
constexpr int LDS_TOTAL = LDS_STAGING + NW * 2 * P_SIZE * 4;             // ~102 KiB: one 8-wavefront workgroup per CU

typedef unsigned rc_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }

// B-operand fragment straight from global memory (L2): 16 B per lane, wave-uniform base + 32-bit lane offset.
// Issued as inline asm so that the loads stay where they are written (one k16 step ahead of their use) instead of
// being sunk to the use by the register-pressure scheduler; RC_WAIT_B orders them.
#ifdef RCMARL_EMU
typedef uint4 rc_bfrag;
__device__ __forceinline__ void ldg_b(rc_bfrag& dst, const unsigned char* sbase, unsigned voff) { dst = ld_u4(sbase + voff); }
__device__ __forceinline__ uint4 as_u4(const rc_bfrag& v) { return v; }
#define RC_WAIT_B(n, x) ((void)0)
#else
typedef rc_u4 rc_bfrag;
__device__ __forceinline__ void ldg_b(rc_bfrag& dst, const unsigned char* sbase, unsigned voff) {
  // (s_nop 4: the base may come straight out of an SALU add; an SGPR written by the SALU needs wait states before a
  // vector-memory instruction reads it, and hipcc pads nothing inside an asm statement)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ uint4 as_u4(const rc_bfrag& v) { return __builtin_bit_cast(uint4, v); }
// wait until at most n vector-memory operations of this wavefront are outstanding; x is the register the caller is
// about to read (the dependency keeps its uses behind the wait)
#define RC_WAIT_B(n, x) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(n) : "memory")
#endif

// workgroup id -> (seed, tile within the seed); all tiles of a seed on one XCD when S % 8 == 0 (as lattice_gemm.hip)
__device__ __forceinline__ void fit_decode(int g, int per_seed, int S, int& seed, int& w) {
  if ((S & 7) == 0) {
    const int xcd = g & 7, q = g >> 3;
    seed = xcd + 8 * (q / per_seed);
    w = q % per_seed;
  } else {
    seed = g / per_seed;
    w = g - seed * per_seed;
  }
}

__global__ __launch_bounds__(512) void k_lat_fit(const unsigned char* __restrict__ wpf, int wpf_rt, int wpf_kt,
                                                 const unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                 const float* __restrict__ theta, const float* __restrict__ y,
                                                 float* __restrict__ partials, unsigned char* __restrict__ dzp,
                                                 int dzp_rt, int dzp_kt, int S, int N, int B, int in_dim, int ldp,
                                                 int ldb, int mtiles, int ntiles) {
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int s, w;
  fit_decode(blockIdx.x, mtiles * ntiles, S, s, w);
  const int bn = w % ntiles, bm = w / ntiles;        // n fastest: the workgroups resident together share W' panels
  const int n_ktiles = (in_dim + 31) >> 5;

  // ---- k-loop ------------------------------------------------------------------------------------------------
  rc_f32x16 acc[FMB];
#pragma unroll
  for (int mb = 0; mb < FMB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
  {
    // A: burst q = piece*10 + j covers rows 16j..16j+15 of one piece; wavefront `wave` issues bursts wave, wave+8, ..
    const unsigned char* wp_s = wpf + (long)s * wpf_rt * wpf_kt * (3 * RC_PK_BLOCK);
    const unsigned char* gsrc[4];
    unsigned gdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int q = wave + NW * i;
      if (q >= A_BURSTS) q = A_BURSTS - 1;          // (not issued: see n_bursts)
      const int p = q / 10, j = q - 10 * p;
      const int R = bm * FBM + 16 * j;
      gsrc[i] = wp_s + ((long)(R >> 7) * wpf_kt * 3 + p) * RC_PK_BLOCK + (R & 127) * 64;
      gdst[i] = p * A_PIECE + j * 1024;
    }
    const int n_bursts = wave < A_BURSTS - 3 * NW ? 4 : 3;          // 30 bursts over 8 wavefronts
    const unsigned lane16 = lane * 16;
    const rc_lds_t lds0 = rc_lds_addr(lds);
    auto stage = [&](int buf, int t) {
#pragma unroll
      for (int i = 0; i < 3; ++i) RC_GLDS16S(gsrc[i] + (long)t * (3 * RC_PK_BLOCK), lane16, lds0 + buf * A_STAGE + gdst[i]);
      if (n_bursts == 4) RC_GLDS16S(gsrc[3] + (long)t * (3 * RC_PK_BLOCK), lane16, lds0 + buf * A_STAGE + gdst[3]);
    };
    // A fragment: row m = 32*mb + l31, chunk (2*ks + half) ^ ((m>>2)&3)
    const int swa = (l31 >> 2) & 3;
    const int coa0 = ((0 + half) ^ swa) << 4, coa1 = ((2 + half) ^ swa) << 4;
    int offA[FMB];
#pragma unroll
    for (int mb = 0; mb < FMB; ++mb) offA[mb] = (32 * mb + l31) * 64;
    // B fragment: replay row r = row0 + l31, 16 B at chunk (2*ks + half) ^ ((r>>2)&3) of k-tile t
    const unsigned char* kp_s = kp + (long)s * kp_rt * kp_kt * RC_PK_BLOCK;
    unsigned offB0, offB1;
    {
      const int r = bn * FBN + wave * FWN + l31;
      const unsigned base = (unsigned)((r >> 7) * kp_kt) * RC_PK_BLOCK + (r & 127) * 64;
      const int sw = (r >> 2) & 3;
      offB0 = base + (((0 + half) ^ sw) << 4);
      offB1 = base + (((2 + half) ^ sw) << 4);
    }
    rc_bfrag b0, b1;                                   // fragments of ks = 0 / ks = 1
    stage(0, 0);
    ldg_b(b0, kp_s, offB0);
    auto kstep = [&](const unsigned char* st, int co, const uint4 bf) {
#pragma unroll
      for (int p = 2; p >= 0; --p) {                   // smallest pieces first (as lattice_gemm.hip: same fp32 sums)
        uint4 af[FMB];
#pragma unroll
        for (int mb = 0; mb < FMB; ++mb) af[mb] = ld_u4(st + p * A_PIECE + offA[mb] + co);
#pragma unroll
        for (int mb = 0; mb < FMB; ++mb) acc[mb] = rc_mfma_bf16(af[mb], bf, acc[mb]);
      }
    };
    for (int t = 0; t < n_ktiles; ++t) {
      const int cur = t & 1;
      RC_WAIT_B(0, b0);                                // this wavefront's A bursts of tile t and B(t, ks 0) have landed
      __syncthreads();                                 // ... everybody's bursts; all reads of the other stage are done
      const bool more = t + 1 < n_ktiles;
      ldg_b(b1, kp_s + (long)t * RC_PK_BLOCK, offB1);
      if (more) stage(cur ^ 1, t + 1);
      const unsigned char* st = lds + cur * A_STAGE;
      kstep(st, coa0, as_u4(b0));
      if (more) {
        ldg_b(b0, kp_s + (long)(t + 1) * RC_PK_BLOCK, offB0);
        // B(t, ks 1) is older than the bursts of tile t+1 and the load just issued
        if (n_bursts == 4) RC_WAIT_B(5, b1); else RC_WAIT_B(4, b1);
      } else {
        RC_WAIT_B(0, b1);
      }
      kstep(st, coa1, as_u4(b1));
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------
  __syncthreads();                                     // all fragment reads done: the stage area becomes scratch
  float* wts = reinterpret_cast<float*>(lds + LDS_WTS);
  float* panel = reinterpret_cast<float*>(lds + LDS_PANELS) + wave * PANEL;
  float* sA = panel;
  float* sB = panel + PA_ROWS * PLD;
  float* staging = reinterpret_cast<float*>(lds + LDS_STAGING);
  {
    // [b1 | W2 | b2 | W3 | b3] of an agent are 461 consecutive floats of its parameter row, starting at b1
    const int o_b1 = in_dim * HID;
    for (int e = threadIdx.x; e < FAG * W_W2T; e += 64 * NW) {
      const int a8 = e / W_W2T, o = e - a8 * W_W2T;
      const int agent = bm * FAG + a8;
      const float v = (agent < N && o < W_B3 + 1) ? theta[((long)s * N + agent) * ldp + o_b1 + o] : 0.f;
      wts[a8 * WS + o] = v;
      if (o >= W_W2 && o < W_B2) {                     // W2[j][k] -> W2T[k][j]
        const int jk = o - W_W2, j = jk / HID, k = jk - j * HID;
        wts[a8 * WS + W_W2T + k * HID + j] = v;
      }
    }
    // constant rows of this wavefront's panels: ones (-> gb2), zeros (padding of the 32x32 tile)
    if (lane < FWN) {
      sA[HID * PLD + lane] = 1.f;
      sA[(HID + 1) * PLD + lane] = 0.f;
      sB[HID * PLD + lane] = 0.f;
    }
  }
  __syncthreads();
  const int row_l = bn * FBN + wave * FWN + l31;       // the lane's replay row
  const bool valid = row_l < B;
  const float fb = (float)B;
  // lane part of the dz store offset (rcmarl_lattice.h packed layout, k = replay row): k-tile, chunk, position in chunk
  const unsigned dz_lane = (unsigned)(row_l >> 5) * (3 * RC_PK_BLOCK) + ((((unsigned)row_l & 31u) >> 3) << 4) + ((unsigned)row_l & 7u) * 2;
  unsigned char* dzp_s = dzp + (long)s * dzp_rt * dzp_kt * (3 * RC_PK_BLOCK);
  const int nchunk = ntiles;

#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int a8 = 4 * half + g;
    const int agent = bm * FAG + a8;                   // differs between the two halves of the wavefront
    const bool agent_ok = agent < N;
    const float* wl = wts + a8 * WS;
    float* st_w = staging + wave * (2 * P_SIZE);
    // ---- layer 1: a1 = lrelu(z1 + b1)
    float a1[HID];
#pragma unroll
    for (int q = 0; q < HID / 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(wl + W_B1 + 4 * q);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = HID * g + 4 * q + e;
        const float z = acc[t >> 4][t & 15] + bb[e];
        a1[4 * q + e] = fmaxf(z, RC_LEAK * z);
      }
    }
    // ---- layer 2 forward: z2[k] = sum_j a1[j] W2[j][k] (j ascending), a2 = lrelu(z2 + b2).  Plain v_fma_f32: the
    // packed form has the same throughput on this VALU and costs extra beside MFMAs (MI355X_MICROARCH.md).  Two rows of
    // W2 per basic block (RC_OPAQUE_TRUE: keeps the weight reads next to the FMAs that consume them).
    float z2[HID];
#pragma unroll
    for (int k = 0; k < HID; ++k) z2[k] = 0.f;
#pragma unroll
    for (int jj = 0; jj < HID; jj += 2) {
      if (RC_OPAQUE_TRUE()) {
#pragma unroll
        for (int j = jj; j < jj + 2; ++j) {
#pragma unroll
          for (int q4 = 0; q4 < HID / 4; ++q4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + W_W2 + j * HID + 4 * q4);
            z2[4 * q4 + 0] = fmaf(a1[j], w4.x, z2[4 * q4 + 0]);
            z2[4 * q4 + 1] = fmaf(a1[j], w4.y, z2[4 * q4 + 1]);
            z2[4 * q4 + 2] = fmaf(a1[j], w4.z, z2[4 * q4 + 2]);
            z2[4 * q4 + 3] = fmaf(a1[j], w4.w, z2[4 * q4 + 3]);
          }
        }
      }
    }
    float a2[HID], w3[HID];
    float v = 0.f;
    if (RC_OPAQUE_TRUE()) {
#pragma unroll
      for (int q4 = 0; q4 < HID / 4; ++q4) {
        const float4 b4 = *reinterpret_cast<const float4*>(wl + W_B2 + 4 * q4);
        const float4 w4 = *reinterpret_cast<const float4*>(wl + W_W3 + 4 * q4);
        a2[4 * q4 + 0] = rc_lrelu(z2[4 * q4 + 0] + b4.x);
        a2[4 * q4 + 1] = rc_lrelu(z2[4 * q4 + 1] + b4.y);
        a2[4 * q4 + 2] = rc_lrelu(z2[4 * q4 + 2] + b4.z);
        a2[4 * q4 + 3] = rc_lrelu(z2[4 * q4 + 3] + b4.w);
        w3[4 * q4] = w4.x; w3[4 * q4 + 1] = w4.y; w3[4 * q4 + 2] = w4.z; w3[4 * q4 + 3] = w4.w;
      }
#pragma unroll
      for (int k = 0; k < HID; ++k) v = fmaf(a2[k], w3[k], v);
      v += wl[W_B3];
    }
    // ---- MSE gradient
    const bool row_ok = agent_ok && valid;
    const float yv = row_ok ? y[((long)s * N + agent) * ldb + row_l] : 0.f;
    const float diff = row_ok ? v - yv : 0.f;
    const float dv = (2.0f * diff) / fb;
    // ---- dz2[k] = dv * W3[k] * lrelu'(z2[k]); the row sums that need a2: gW3[k] = sum a2[k]*dv, gb3 = sum dv
    float dz2[HID];
    {
      float sm[HID + 1];
#pragma unroll
      for (int k = 0; k < HID; ++k) {
        sm[k] = a2[k] * dv;
        dz2[k] = dv * w3[k] * rc_lrelu_grad_from_act(a2[k]);
      }
      sm[HID] = dv;
      static_assert((HID + 1) % 3 == 0, "sums are reduced three at a time");
#pragma unroll
      for (int q = 0; q < (HID + 1) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
      if (l31 == 31) {
#pragma unroll
        for (int k = 0; k < HID + 1; ++k) st_w[half * P_SIZE + P_GW3 + k] = sm[k];     // gW3[0..19], gb3
      }
    }
    // ---- gW2 = a1^T dz2, gb2 = 1^T dz2 of this wavefront's 32 rows: f32 matrix core, one half's agent at a time
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      RC_WAVE_SYNC();                                  // the previous product's fragment reads are done
      if (half == hh) {
#pragma unroll
        for (int k = 0; k < HID; ++k) {
          sA[k * PLD + l31] = a1[k];
          sB[k * PLD + l31] = dz2[k];
        }
      }
      RC_WAVE_SYNC();
      rc_f32x16 g1;
#pragma unroll
      for (int q = 0; q < 16; ++q) g1[q] = 0.f;
      const int ia = (l31 < HID + 1 ? l31 : HID + 1) * PLD + half;      // A rows: a1 units | ones | zeros
      const int ib = (l31 < HID ? l31 : HID) * PLD + half;              // B rows: dz2 units | zeros
#pragma unroll
      for (int m = 0; m < FWN / 2; ++m) g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[ia + 2 * m], sB[ib + 2 * m], g1, 0, 0, 0);
      // D[row = (q&3) + 8*(q>>2) + 4*half][col = l31] -> gW2[row][col] (row < 20), gb2[col] (row == 20)
      if (l31 < HID) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * half;
          if (row <= HID) st_w[hh * P_SIZE + row * HID + l31] = g1[q];      // row 20 lands on P_GB2 + col
        }
      }
    }
    // ---- layer 2 backward: dz1[j] = (sum_k dz2[k] W2[j][k]) * lrelu'(z1[j]), k ascending (rows of W2^T)
    float da[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) da[j] = 0.f;
#pragma unroll
    for (int kk = 0; kk < HID; kk += 2) {
      if (RC_OPAQUE_TRUE()) {
#pragma unroll
        for (int k = kk; k < kk + 2; ++k) {
#pragma unroll
          for (int q4 = 0; q4 < HID / 4; ++q4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + W_W2T + k * HID + 4 * q4);
            da[4 * q4 + 0] = fmaf(dz2[k], w4.x, da[4 * q4 + 0]);
            da[4 * q4 + 1] = fmaf(dz2[k], w4.y, da[4 * q4 + 1]);
            da[4 * q4 + 2] = fmaf(dz2[k], w4.z, da[4 * q4 + 2]);
            da[4 * q4 + 3] = fmaf(dz2[k], w4.w, da[4 * q4 + 3]);
          }
        }
      }
    }
    // dz1 leaves as three exact bf16 pieces (packed rows = (agent, unit) in NATURAL order, k = replay row) and joins
    // gb1[j] = sum dz1[j]; loss = sum diff^2
    {
      float sm[HID + 1];
#pragma unroll
      for (int q = 0; q < H2; ++q) {
        const float d0 = da[2 * q] * rc_lrelu_grad_from_act(a1[2 * q]);
        const float d1 = da[2 * q + 1] * rc_lrelu_grad_from_act(a1[2 * q + 1]);
        sm[2 * q] = d0;
        sm[2 * q + 1] = d1;
        if (agent_ok) {
          unsigned ph, pm, pl;
          rc_split3_pair(d0, d1, ph, pm, pl);                      // bits 0-15: unit 2q, bits 16-31: unit 2q+1
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int R = agent * HID + 2 * q + u;
            const unsigned roff = (unsigned)(R >> 7) * (unsigned)dzp_kt * (3 * RC_PK_BLOCK) + (unsigned)(R & 127) * 64;
            const unsigned sw = (unsigned)(((R >> 2) & 3) << 4);
            unsigned char* p = dzp_s + roff + (dz_lane ^ sw);
            *reinterpret_cast<unsigned short*>(p) = (unsigned short)(u ? ph >> 16 : ph);
            *reinterpret_cast<unsigned short*>(p + RC_PK_BLOCK) = (unsigned short)(u ? pm >> 16 : pm);
            *reinterpret_cast<unsigned short*>(p + 2 * RC_PK_BLOCK) = (unsigned short)(u ? pl >> 16 : pl);
          }
        }
      }
      sm[HID] = diff * diff;
#pragma unroll
      for (int q = 0; q < (HID + 1) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
      if (l31 == 31) {
#pragma unroll
        for (int k = 0; k < HID + 1; ++k) st_w[half * P_SIZE + P_GB1 + k] = sm[k];     // gb1[0..19], loss
      }
    }
    __syncthreads();
    // ---- the eight wavefronts' records, summed in a fixed order -> partials[s][agent][bn]
    for (int e = threadIdx.x; e < 2 * P_SIZE; e += 64 * NW) {
      const int hh = e / P_SIZE, idx = e - hh * P_SIZE;
      const int ag = bm * FAG + 4 * hh + g;
      if (ag < N) {
        float r[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) r[ww] = staging[ww * 2 * P_SIZE + e];
        partials[(((long)s * N + ag) * nchunk + bn) * P_SIZE + idx] =
            ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      }
    }
    __syncthreads();
  }
}

// W1 of every agent -> three bf16 pieces of alpha_k*W1 in FIT ORDER (rcmarl_lattice.h): the A operand of k_lat_fit.
// One workgroup = 128 natural columns x 32 features; every 16-byte chunk lands in the row rc_fit_row() names.
__global__ __launch_bounds__(256) void k_w1_split_fit(const float* __restrict__ theta, const float* __restrict__ alpha,
                                                      unsigned char* __restrict__ wpf, int N, int in_dim, int ldp,
                                                      int wpf_rt, int wpf_kt) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, r = t & 127;
  const int col = rt * 128 + r, ncols = N * HID;
  const bool col_ok = col < ncols;
  const int ag = col_ok ? col / HID : 0, j = col - ag * HID;
  const float* th = theta + ((long)s * N + ag) * ldp + j;
  const int R = rc_fit_row(ag, col_ok ? j : 0);
  unsigned char* blk = wpf + (long)s * wpf_rt * wpf_kt * 3 * RC_PK_BLOCK + ((long)(R >> 7) * wpf_kt + kt) * 3 * RC_PK_BLOCK;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c4 = (t >> 7) + 2 * q;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kt * 32 + 8 * c4 + e;
      w[e] = (col_ok && k < in_dim) ? th[(long)k * HID] * alpha[k] : 0.f;
    }
    uint4 vh, vm, vl;
    rc_split3_pair(w[0], w[1], vh.x, vm.x, vl.x);
    rc_split3_pair(w[2], w[3], vh.y, vm.y, vl.y);
    rc_split3_pair(w[4], w[5], vh.z, vm.z, vl.z);
    rc_split3_pair(w[6], w[7], vh.w, vm.w, vl.w);
    if (col_ok) {
      const int o = (R & 127) * 64 + ((c4 ^ ((R >> 2) & 3)) << 4);
      *reinterpret_cast<uint4*>(blk + o) = vh;
      *reinterpret_cast<uint4*>(blk + RC_PK_BLOCK + o) = vm;
      *reinterpret_cast<uint4*>(blk + 2 * RC_PK_BLOCK + o) = vl;
    }
  }
}

bool fit_want_lds() {
#ifndef RCMARL_EMU
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lat_fit), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             LDS_TOTAL) == hipSuccess;
  return ok;
#else
  return true;
#endif
}

}  // namespace

RCMARL_EXPORT int rcmarl_fit_rows(int n_agents) { return rc_ceil_div(n_agents, RC_FIT_AGENTS) * RC_FIT_ROWS; }

RCMARL_EXPORT int rcmarl_w1_split_fit(const float* theta, const float* alpha, void* wpf, int S, int N, int in_dim, int hid,
                                      int ldp, int wpf_rt, int wpf_kt, void* stream) {
  if (!theta || !alpha || !wpf || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63) || ldp < in_dim * hid + hid) return RCMARL_ERR_ARG;
  if (hid != HID) return RCMARL_ERR_UNSUPPORTED;
  if ((long)wpf_rt * 128 < (long)rcmarl_fit_rows(N) || wpf_kt * 32 < in_dim) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(in_dim, 32), rc_ceil_div(N * HID, 128), S), block(256);
  RCMARL_LAUNCH(k_w1_split_fit, grid, block, 0, stream, theta, alpha, (unsigned char*)wpf, N, in_dim, ldp, wpf_rt, wpf_kt);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_fit_fused_lattice(const void* kp, int kp_rt, int kp_kt, const void* wpf, int wpf_rt, int wpf_kt,
                                           const float* theta, const float* y, float* partials, void* dzp, int dzp_rt,
                                           int dzp_kt, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                                           void* stream) {
  if (!kp || !wpf || !theta || !y || !partials || !dzp || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || (ldp & 63) ||
      (ldb & 63) || ldb < B || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid != HID) return RCMARL_ERR_UNSUPPORTED;
  const int mtiles = rc_ceil_div(N, FAG), ntiles = rc_ceil_div(B, FBN), ktiles = rc_ceil_div(in_dim, 32);
  if ((long)wpf_rt * 128 < (long)mtiles * FBM || kp_rt < 2 * ntiles || wpf_kt < ktiles || kp_kt < ktiles) return RCMARL_ERR_ARG;
  if (dzp_rt < rc_ceil_div(N * HID, 128) || dzp_kt < ntiles * (FBN / 32)) return RCMARL_ERR_ARG;
  if (!fit_want_lds()) return RCMARL_ERR_LAUNCH;
  const dim3 grid((unsigned)(S * mtiles * ntiles)), block(64 * NW);
  RCMARL_LAUNCH(k_lat_fit, grid, block, LDS_TOTAL, stream, (const unsigned char*)wpf, wpf_rt, wpf_kt, (const unsigned char*)kp,
                kp_rt, kp_kt, theta, y, partials, (unsigned char*)dzp, dzp_rt, dzp_kt, S, N, B, in_dim, ldp, ldb, mtiles,
                ntiles);
  return rcmarl_check_launch();
}
