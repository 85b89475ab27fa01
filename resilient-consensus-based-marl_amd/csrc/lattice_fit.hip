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
#ifndef RC_FIT_NW
#define RC_FIT_NW 4
#endif
constexpr int NW = RC_FIT_NW;                 // wavefronts per workgroup (4: two workgroups per CU, one in its k-loop while
                                              // the other is in its epilogue; 8: one workgroup per CU)
constexpr int NB_MAX = (30 + NW - 1) / NW;    // LDS-DMA bursts per wavefront and stage (the last wavefronts issue one less)
constexpr int FWN = 32;                       // replay rows per wavefront (one MFMA column block)
constexpr int FBN = NW * FWN;                 // replay rows per workgroup = the chunk of a partial record
constexpr int A_PIECE = FBM * 64;             // bytes of one piece of one k32 stage
constexpr int A_STAGE = 3 * A_PIECE;          // 30720
constexpr int A_BURSTS = A_STAGE / 1024;      // 30 LDS-DMA bursts of 1 KiB per stage
// record layout = FitPart<20> of mid_kernels.hip
constexpr int P_GB2 = HID * HID, P_GW3 = P_GB2 + HID, P_GB3 = P_GW3 + HID, P_GB1 = P_GB3 + 1,
              P_LOSS = P_GB1 + HID, P_SIZE = P_LOSS + 1;                  // 462 floats
// epilogue LDS (aliases the k-loop stages): weights of the 8 agents | per-wavefront panels | staging area
constexpr int W_B1 = 0, W_W2 = HID, W_B2 = W_W2 + HID * HID, W_W3 = W_B2 + HID, W_B3 = W_W3 + HID, W_W2T = 464;
constexpr int WS = W_W2T + HID * HID;         // floats per agent: b1(20) | W2(400) | b2(20) | W3(20) | b3 | pad | W2^T(400)
constexpr int PLD = FWN + 1;                  // panel row stride (floats): (unit*33 + row) % 32 distinct over units
// panels of the reduction product G = A^T B over the wavefront's 32 rows (one half's agent at a time), 32 x 32:
//   A rows: a1[0..19] | ones | a2[0..10]*dv          B rows: dz2[0..19] | ones | dv | diff^2 | a2[11..19]*dv
//   G[i<20][j<20] = gW2, G[20][j<20] = gb2, G[21+i][20] = gW3[i], G[20][21] = gb3, G[20][22] = loss, G[20][23+i] = gW3[11+i]
constexpr int PA_ROWS = 32, PB_ROWS = 32, GW3_IN_A = 11;
constexpr int PANEL = ((PA_ROWS + PB_ROWS) * PLD + 63) / 64 * 64;        // floats per wavefront
constexpr int LDS_WTS = 0, LDS_PANELS = FAG * WS * 4, LDS_STAGING = LDS_PANELS + NW * PANEL * 4;
static_assert(LDS_STAGING >= 2 * A_STAGE, "the k-loop stages lie inside the weights + panels area");
static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts");
constexpr int P_STRIDE = P_SIZE + 64 + 2;     // a staged record + one dump word per lane (elements of the 32x32 tile outside the record)
constexpr int LDS_TOTAL = LDS_STAGING + NW * 2 * P_STRIDE * 4;           // ~126 KiB: one 8-wavefront workgroup per CU

__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }

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

__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_lat_fit(const unsigned char* __restrict__ wpf, int wpf_rt, int wpf_kt,
                                                 const unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                 const float* __restrict__ theta, const float* __restrict__ y,
                                                 float* __restrict__ partials, unsigned char* __restrict__ dzp,
                                                 int dzp_rt, int dzp_kt, int S, int N, int B, int in_dim, int ldp,
                                                 int ldb, int mtiles, int ntiles, int dbg) {
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int s, w;
  fit_decode(blockIdx.x, mtiles * ntiles, S, s, w);
  const int bn = w % ntiles, bm = w / ntiles;        // n fastest: the workgroups resident together share W' panels
  const int n_ktiles = (in_dim + 31) >> 5;
#ifndef RCMARL_EMU
  // De-phasing (pure scheduling, no effect on results): the two workgroups that share a CU would start together and
  // run their k-loops (matrix core) and their epilogues (VALU, f32 MFMA, stores) in lockstep.  In the first round of
  // residency the workgroup that got the UPPER LDS allocation sleeps about half a tile (dbg bits 8-15 x ~3.4 us), so
  // one workgroup's epilogue runs beside the other's k-loop; later workgroups inherit the offset.
  if ((dbg >> 8) & 0xff) {
    // dbg bit 16 selects WHO sleeps: 0 = the workgroup whose LDS allocation is the upper one (HW_REG_LDS_ALLOC.LDS_BASE != 0),
    // 1 = workgroups 256..511 (the second round of the dispatcher over the 256 CUs)
    const bool second = (dbg >> 16) & 1 ? (blockIdx.x >= 256u && blockIdx.x < 512u)
                                        : (blockIdx.x < 512u && (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)) & 0xff) != 0);
    if (second)
      for (int i = 0; i < ((dbg >> 8) & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // ---- k-loop ------------------------------------------------------------------------------------------------
  rc_f32x16 acc[FMB];
#pragma unroll
  for (int mb = 0; mb < FMB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
  {
    // A: burst q = piece*10 + j covers rows 16j..16j+15 of one piece; wavefront `wave` issues bursts wave, wave+8, ..
    const unsigned char* wp_s = wpf + (long)s * wpf_rt * wpf_kt * (3 * RC_PK_BLOCK);
    const unsigned char* gsrc[NB_MAX];
    unsigned gdst[NB_MAX];
#pragma unroll
    for (int i = 0; i < NB_MAX; ++i) {
      int q = wave + NW * i;
      if (q >= A_BURSTS) q = A_BURSTS - 1;          // (not issued: see n_bursts)
      const int p = q / 10, j = q - 10 * p;
      const int R = bm * FBM + 16 * j;
      gsrc[i] = wp_s + ((long)(R >> 7) * wpf_kt * 3 + p) * RC_PK_BLOCK + (R & 127) * 64;
      gdst[i] = p * A_PIECE + j * 1024;
    }
    const bool last_burst = wave < A_BURSTS - (NB_MAX - 1) * NW;    // 30 bursts over NW wavefronts
    const unsigned lane16 = lane * 16;
    const rc_lds_t lds0 = rc_lds_addr(lds);
    auto stage = [&](int buf, int t) {
#pragma unroll
      for (int i = 0; i < NB_MAX - 1; ++i) RC_GLDS16S(gsrc[i] + (long)t * (3 * RC_PK_BLOCK), lane16, lds0 + buf * A_STAGE + gdst[i]);
      if (last_burst) RC_GLDS16S(gsrc[NB_MAX - 1] + (long)t * (3 * RC_PK_BLOCK), lane16, lds0 + buf * A_STAGE + gdst[NB_MAX - 1]);
    };
    // A fragment: row m = 32*mb + l31, chunk (2*ks + half) ^ ((m>>2)&3)
    const int swa = (l31 >> 2) & 3;
    const int coa0 = ((0 + half) ^ swa) << 4, coa1 = ((2 + half) ^ swa) << 4;
    int offA[FMB];
#pragma unroll
    for (int mb = 0; mb < FMB; ++mb) offA[mb] = (32 * mb + l31) * 64;
    // B fragment: replay row r = row0 + l31, 16 B at chunk (2*ks + half) ^ ((r>>2)&3) of k-tile t
    const unsigned char* kp_s = kp + (long)s * kp_rt * kp_kt * RC_PK_BLOCK;
    unsigned offB0, offB1;                             // (32-bit: one seed's K image is a few MB)
    {
      const int r = bn * FBN + wave * FWN + l31;
      const unsigned base = (unsigned)((r >> 7) * kp_kt) * RC_PK_BLOCK + (r & 127) * 64;
      const int sw = (r >> 2) & 3;
      offB0 = base + (((0 + half) ^ sw) << 4);
      offB1 = base + (((2 + half) ^ sw) << 4);
    }
    // B fragments are PLAIN loads (hipcc counts them; the LDS-DMA bursts it cannot see only make its vmcnt waits
    // conservative), each pinned one k16 step ahead of its use by a scheduling fence.  (Hand-issued asm loads with a
    // separate asm wait were tried first: hipcc placed register copies of the not-yet-landed destination BEFORE the
    // wait on two of three paths -- intermittently stale fragments at 24 k-tiles.)
    uint4 b0, b1;                                      // fragments of ks = 0 / ks = 1
    stage(0, 0);
    b0 = ld_u4(kp_s + offB0);
    RC_SCHED_FENCE();
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
    for (int t = 0; t < ((dbg & 2) ? 1 : n_ktiles); ++t) {          // (dbg 2: one k-tile only = the epilogue's time)
      const int cur = t & 1;
      RC_WAIT_VMEM();                                  // this wavefront's A bursts of tile t (and B(t, ks 0)) have landed
      __syncthreads();                                 // ... everybody's bursts; all reads of the other stage are done
      const bool more = t + 1 < n_ktiles;
      b1 = ld_u4(kp_s + (long)t * RC_PK_BLOCK + offB1);
      RC_SCHED_FENCE();
      if (more) stage(cur ^ 1, t + 1);
      const unsigned char* st = lds + cur * A_STAGE;
      kstep(st, coa0, b0);
      RC_SCHED_FENCE();
      if (more) b0 = ld_u4(kp_s + (long)(t + 1) * RC_PK_BLOCK + offB0);
      RC_SCHED_FENCE();
      kstep(st, coa1, b1);
    }
  }

  if (dbg & 1) {                                       // measurement aid (RCMARL_FIT_DBG=1): k-loop only, results WRONG
    float t = 0.f;
#pragma unroll
    for (int mb = 0; mb < FMB; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[mb][i];
    if (t == 12345.678f) partials[0] = t;
    return;
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
    // constant rows of this wavefront's panels: the ones row of A (-> column sums of B) and of B (-> row sums of A)
    if (lane < FWN) {
      sA[HID * PLD + lane] = 1.f;
      sB[HID * PLD + lane] = 1.f;
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
  // where element q of this lane's 32x32 result tile D[i = (q&3) + 8*(q>>2) + 4*half][j = l31] goes inside a staged record
  // (layout of the product: see PA_ROWS above); elements outside the record go to the lane's dump word.  Same for every
  // agent: computed once, sixteen registers.
  int rec_idx[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * half, j = l31;
    int idx = -1;
    if (i < HID) idx = j < HID ? i * HID + j : -1;
    else if (i == HID) idx = j < HID ? P_GB2 + j : (j == HID + 1 ? P_GB3 : (j == HID + 2 ? P_LOSS : (j >= HID + 3 ? P_GW3 + GW3_IN_A + (j - HID - 3) : -1)));
    else idx = j == HID ? P_GW3 + (i - HID - 1) : -1;
    rec_idx[q] = idx >= 0 ? idx : P_SIZE + lane;
  }

#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int a8 = 4 * half + g;
    const int agent = bm * FAG + a8;                   // differs between the two halves of the wavefront
    const bool agent_ok = agent < N;
    const float* wl = wts + a8 * WS;
    float* st_w = staging + wave * (2 * P_STRIDE);
    // ---- layer 1: a1 = lrelu(z1 + b1), row per lane (the lane's own agent)
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
    // ---- layer 2 forward on the f32 matrix core, TRANSPOSED so that the result lands row-per-lane again:
    //   Z_b[unit i][row j] = sum_m W2_b[m][i] * a1_b[row j][m]     (block b = the half's agent; two-block 32x32x1 MFMA)
    // A operand: lane (i, b) reads W2_b[m][i] from LDS (units >= 20 read finite neighbours: unused rows of Z);
    // B operand: lane (row, b) supplies ITS OWN register a1[m].  20 instructions, m ascending = the fmaf chain of the
    // VALU form bit for bit.  Lane (row, h) then holds units e + 8q + 4h of BOTH agents (registers 16b + 4q + e);
    // twelve v_permlane32_swap hand the other agent's values across: afterwards unit e + 8q of the lane's own agent
    // is in register 4q + e and unit e + 8q + 4 in register 16 + 4q + e, for both halves alike.
    float z2[HID];
    {
      rc_f32x32 zz;
#pragma unroll
      for (int r = 0; r < 32; ++r) zz[r] = 0.f;
      const float* wcol = wl + W_W2 + l31;
      if (!(dbg & 32))
#pragma unroll
      for (int m = 0; m < HID; ++m) zz = __builtin_amdgcn_mfma_f32_32x32x1f32(wcol[m * HID], a1[m], zz, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = zz[4 * q + e], hi = zz[16 + 4 * q + e];
          rc_swap32(lo, hi);
          z2[8 * q + e] = lo;
          if (8 * q + 4 + e < HID) z2[8 * q + 4 + e] = hi;
        }
    }
    // ---- a2 = lrelu(z2 + b2); head v = a2 . W3 + b3 (k ascending); MSE gradient
    float a2[HID], w3[HID];
    float v = 0.f;
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
    const bool row_ok = agent_ok && valid;
    const float yv = row_ok ? y[((long)s * N + agent) * ldb + row_l] : 0.f;
    const float diff = row_ok ? v - yv : 0.f;
    const float dv = (2.0f * diff) / fb;
    // ---- dz2[k] = dv * W3[k] * lrelu'(z2[k]); g3[k] = a2[k]*dv (summed over rows = gW3)
    float dz2[HID], g3[HID];
#pragma unroll
    for (int k = 0; k < HID; ++k) {
      g3[k] = a2[k] * dv;
      dz2[k] = dv * w3[k] * rc_lrelu_grad_from_act(a2[k]);
    }
    // ---- everything summed over the wavefront's 32 rows except gb1: ONE 32x32 product per agent on the f32 matrix core
    // (panel layout above), one half's agent at a time through the wavefront's panel buffer
    if (!(dbg & 16))
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      RC_WAVE_SYNC();                                  // the previous product's fragment reads are done
      if (half == hh) {
#pragma unroll
        for (int k = 0; k < HID; ++k) {
          sA[k * PLD + l31] = a1[k];
          sB[k * PLD + l31] = dz2[k];
        }
#pragma unroll
        for (int i = 0; i < GW3_IN_A; ++i) sA[(HID + 1 + i) * PLD + l31] = g3[i];
        sB[(HID + 1) * PLD + l31] = dv;
        sB[(HID + 2) * PLD + l31] = diff * diff;
#pragma unroll
        for (int i = 0; i < HID - GW3_IN_A; ++i) sB[(HID + 3 + i) * PLD + l31] = g3[GW3_IN_A + i];
      }
      RC_WAVE_SYNC();
      rc_f32x16 g1;
#pragma unroll
      for (int q = 0; q < 16; ++q) g1[q] = 0.f;
      const int ia = l31 * PLD + half, ib = l31 * PLD + half;
#pragma unroll
      for (int m = 0; m < FWN / 2; ++m) g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[ia + 2 * m], sB[ib + 2 * m], g1, 0, 0, 0);
      float* rec = st_w + hh * P_STRIDE;
#pragma unroll
      for (int q = 0; q < 16; ++q) rec[rec_idx[q]] = g1[q];
    }
    // ---- layer 2 backward, same transposed form:  DA_b[unit j][row] = sum_k W2_b[j][k] * dz2_b[row][k]  (k ascending)
    float da[HID];
    {
      rc_f32x32 dd;
#pragma unroll
      for (int r = 0; r < 32; ++r) dd[r] = 0.f;
      const float* wtcol = wl + W_W2T + l31;           // W2^T[k][j]: lane j reads a conflict-free column
      if (!(dbg & 32))
#pragma unroll
      for (int k = 0; k < HID; ++k) dd = __builtin_amdgcn_mfma_f32_32x32x1f32(wtcol[k * HID], dz2[k], dd, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = dd[4 * q + e], hi = dd[16 + 4 * q + e];
          rc_swap32(lo, hi);
          da[8 * q + e] = lo;
          if (8 * q + 4 + e < HID) da[8 * q + 4 + e] = hi;
        }
    }
    // ---- dz1[j] = da[j] * lrelu'(z1[j]) leaves as three exact bf16 pieces (packed rows = (agent, unit) in NATURAL
    // order, k = replay row) and joins gb1[j] = sum dz1[j]
    {
      float sm[HID + 1];
      const unsigned dz_agent = agent_ok ? (unsigned)agent * HID : 0u;
#pragma unroll
      for (int q = 0; q < H2; ++q) {
        const float d0 = da[2 * q] * rc_lrelu_grad_from_act(a1[2 * q]);
        const float d1 = da[2 * q + 1] * rc_lrelu_grad_from_act(a1[2 * q + 1]);
        sm[2 * q] = d0;
        sm[2 * q + 1] = d1;
        unsigned ph, pm, pl;
        rc_split3_pair(d0, d1, ph, pm, pl);                        // bits 0-15: unit 2q, bits 16-31: unit 2q+1
        if (agent_ok && !(dbg & 4)) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const unsigned R = dz_agent + 2 * q + u;
            // 32-bit offset from the seed's wave-uniform base (one seed's dz image is < 4 GiB): saddr + voffset stores
            const unsigned off = ((R >> 7) * (unsigned)dzp_kt * (3 * RC_PK_BLOCK) + (R & 127u) * 64u + (dz_lane ^ (((R >> 2) & 3u) << 4)));
            *reinterpret_cast<unsigned short*>(dzp_s + off) = (unsigned short)(u ? ph >> 16 : ph);
            *reinterpret_cast<unsigned short*>(dzp_s + (off + RC_PK_BLOCK)) = (unsigned short)(u ? pm >> 16 : pm);
            *reinterpret_cast<unsigned short*>(dzp_s + (off + 2 * RC_PK_BLOCK)) = (unsigned short)(u ? pl >> 16 : pl);
          }
        }
      }
      sm[HID] = 0.f;
      static_assert((HID + 1) % 3 == 0, "sums are reduced three at a time");
      if (!(dbg & 8))
#pragma unroll
      for (int q = 0; q < (HID + 1) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
      if (l31 == 31) {
#pragma unroll
        for (int k = 0; k < HID; ++k) st_w[half * P_STRIDE + P_GB1 + k] = sm[k];
      }
    }
    if (dbg & 64) continue;
    __syncthreads();
    // ---- the eight wavefronts' records, summed in a fixed order -> partials[s][agent][bn]
    for (int e = threadIdx.x; e < 2 * P_SIZE; e += 64 * NW) {
      const int hh = e / P_SIZE, idx = e - hh * P_SIZE;
      const int ag = bm * FAG + 4 * hh + g;
      if (ag < N) {
        float r[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) r[ww] = staging[(ww * 2 + hh) * P_STRIDE + idx];
        float tot = (r[0] + r[1]) + (r[2] + r[3]);
        if constexpr (NW == 8) tot += (r[4] + r[5]) + (r[6] + r[7]);
        partials[(((long)s * N + ag) * nchunk + bn) * P_SIZE + idx] = tot;
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
  static const bool ok = rc_want_lds(k_lat_fit, (size_t)LDS_TOTAL);
  return ok;
}

}  // namespace

RCMARL_EXPORT int rcmarl_fit_rows(int n_agents) { return rc_ceil_div(n_agents, RC_FIT_AGENTS) * RC_FIT_ROWS; }
// partial records per (seed, agent) rcmarl_fit_fused_lattice writes for B replay rows (one per workgroup row tile)
RCMARL_EXPORT int rcmarl_fit_fused_chunks(int B) { return rc_ceil_div(B, FBN); }

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
  if ((long)wpf_rt * 128 < (long)mtiles * FBM || kp_rt * 128 < ntiles * FBN || wpf_kt < ktiles || kp_kt < ktiles) return RCMARL_ERR_ARG;
  if (dzp_rt < rc_ceil_div(N * HID, 128) || dzp_kt < ntiles * (FBN / 32)) return RCMARL_ERR_ARG;
  if (!fit_want_lds()) return RCMARL_ERR_LAUNCH;
  // bits 0-7: measurement aids (0 in production); bits 8-15: de-phasing sleep of the second workgroup of a CU
  static const int dbg = (getenv("RCMARL_FIT_DBG") ? atoi(getenv("RCMARL_FIT_DBG")) & 0xff : 0) |
                         ((getenv("RCMARL_FIT_STAGGER") ? atoi(getenv("RCMARL_FIT_STAGGER")) & 0xff : 0) << 8) |
                         ((getenv("RCMARL_FIT_STAGGER_MODE") ? atoi(getenv("RCMARL_FIT_STAGGER_MODE")) & 1 : 0) << 16);
  const dim3 grid((unsigned)(S * mtiles * ntiles)), block(64 * NW);
  RCMARL_LAUNCH(k_lat_fit, grid, block, LDS_TOTAL, stream, (const unsigned char*)wpf, wpf_rt, wpf_kt, (const unsigned char*)kp,
                kp_rt, kp_kt, theta, y, partials, (unsigned char*)dzp, dzp_rt, dzp_kt, S, N, B, in_dim, ldp, ldb, mtiles,
                ntiles, dbg);
  return rcmarl_check_launch();
}
