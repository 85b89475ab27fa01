// PROTOTYPE (round 4; moved out of the product library in round 5, ABI 3): exact against the three-launch path, measured 0.8x / 1.0x of
// its speed: NOT built, NOT exported.  To build it again: copy to csrc/, restore the "Wf" helpers
// below into csrc/rcmarl_lattice.h, the wf_out epilogue of k_lat_backward_sgd (git show abb9a39:resilient-consensus-based-marl_amd/csrc/
// lattice_gemm.hip), rcmarl_small_sgd_records (same commit, mid_kernels.hip) and the declarations (same commit, include/rcmarl.h);
// its tests and kbench modes are in the same commit (tests/kernel_checks.py check_fit_encode / check_fused_fit / check_forward_mid_fit).
//
// ---- was in csrc/rcmarl_lattice.h:
#if 0
// ---------------------------------------------------------------------------------------------
// "Wf": the forward operand of the fused kernels (fused_fit.hip) -- the two f16 pieces of 2^10 alpha_k W1 of THREE consecutive
// agents (one group) as ready-made MFMA A fragments:  [group][k16 step][piece][slot tile 0/1][lane 64][8 x f16], lane =
// (slot row i = lane & 31, k-group = lane >> 5).  Slot sigma = 10 * (agent % 3) + local unit u lives in tile sigma >> 4, row
// acc_row(sigma & 15, h) = (q & 3) + 8 * (q >> 2) + 4 * h, where unit = v8_unit(h, u): the accumulator layout of the product then IS
// k_mid_fit_v8's "ten units per lane" layout.  Four of the 64 slots are padding (zero).
#define RC_WF_FRAG 1024
#define RC_WF_UT 2
#define RC_WF_STEP (2 * RC_WF_UT * RC_WF_FRAG)
#define RC_WF_KC 8                     // the reduction is padded to a multiple of 2 * RC_WF_KC k16 steps (zeros)
__host__ __device__ static inline int rc_wf_ksp(int in_dim) {
  const int ks = 2 * ((in_dim + 31) / 32);
  return (ks + 2 * RC_WF_KC - 1) / (2 * RC_WF_KC) * (2 * RC_WF_KC);
}
// byte offset, inside one seed's Wf, of the 16-byte chunk row of (agent, unit) at k16 step 0, piece 0, k-group 0
__host__ __device__ static inline long rc_wf_row_offset(int agent, int unit, int ksp) {
  const int g = agent / 3, a = agent - 3 * g;
  const int h = unit < 16 ? (unit >> 3) : ((unit - 16) >> 1), u = unit < 16 ? (unit & 7) : 8 + ((unit - 16) & 1);
  const int sg = 10 * a + u, t = sg >> 4, q = sg & 15, i = (q & 3) + 8 * (q >> 2) + 4 * h;
  return (long)g * ksp * RC_WF_STEP + t * RC_WF_FRAG + i * 16;
}

#endif
// The WHOLE local fit of the cooperative agents' critic / team-reward messages in ONE launch ("fused fit", round 4):
// critic.fit / TR.fit of agents/resilient_CAC_agents.py:118,136 -- `nsteps` full-batch SGD steps on the message copy --
// with the layer-1 activations and dz1 never leaving the compute unit.
//
// Why.  The unfused path runs a step as three launches (lattice forward GEMM -> k_mid_fit_v8 -> lattice backward GEMM) that move
// ~6 GB through HBM per step at BASELINE configs[3], about 4 GB of it the a1t / dz1 round trips between them.  The
// local fit of an agent depends on nothing but the seed's replay rows (shared, read-only) and the agent's own parameters, so a
// workgroup that owns a few agents can run ALL steps by itself: no grid-wide dependency, no partial records, no launches.
//
// Work split.  One workgroup = one seed x G = 3 consecutive agents = 60 hidden units = two 32-slot tiles of the matrix core
// (4 padding slots), 4 wavefronts, ONE wavefront per SIMD (512 registers each: the layer-1 gradient of the three agents stays
// in accumulator registers for the whole step).  Per step the workgroup walks the replay rows in tiles of 256:
//   F  forward     z1^T[slot][row] = sum_k W'[slot][k] K[row][k]   v_mfma_f32_32x32x16_f16, A = two f16 pieces of 2^10 alpha W1
//                  (streamed through LDS by LDS-DMA, 32-KiB stages), B = the integer lattice image of the rows (rcmarl_lattice.h)
//                  loaded STRAIGHT from L2 into registers: a wavefront owns 64 rows, nobody else needs their fragments.
//                  The accumulator layout (lane = replay row, registers = units) IS k_mid_fit_v8's "ten units per lane" layout:
//                  slot sigma = 10*agent + local unit lives in tile sigma>>4, register sigma&15 -- no transposition.
//   M  layers 2-3  forward + MSE + backward per (agent, 32-row block) exactly as k_mid_fit_v8 (f16 pieces, four products per fp32
//                  product), two row blocks in lockstep for instruction-level parallelism; dz1 goes, as two f16 pieces of
//                  2^8 dz1, into LDS planes laid out as MFMA B fragments.
//   B  backward    gW1[feature][slot] += sum_row K^T[feature][row] dz1[row][slot]: a wavefront owns FTW feature tiles x both slot
//                  tiles (accumulators persist over all tiles of the step), A = K^T fragments straight from L2, B = the planes.
// Step end: the workgroup reduces its small gradient records, applies the SGD step to its agents' parameters (W1 from the
// accumulators, the rest from the records), and writes the next step's W' pieces.  Arithmetic per element is that of the
// unfused f16 path (same pieces, same scales, same accumulation order over k); only the order of the row sums differs.
//
// Operand images ("fragment-major", written by rcmarl_fit_encode): a fragment = what the 64 lanes of one MFMA operand hold,
// 1 KiB contiguous, lane l = (row l&31, k-group l>>5) holding 8 consecutive reduction elements:
//   Kf [seed][row tile of 32][k16 step][lane][8 x f16]      rows x features  (forward B operand)
//   KTf[seed][feature tile of 32][row step of 16][lane][8]  features x rows  (backward A operand)
//   Wf [seed][group][k16 step][piece][slot tile][lane][8]   scratch: W' of the group, written and read by its workgroup only
// Range: as k_mid_fit_v8, an agent whose operands leave the f16 range is FLAGGED (flags[seed][agent] = 1); its result is
// not to be used (the caller redoes it on the unfused path, which has the fp32 fix-up).
#include "rcmarl_lattice.h"
#include <stdlib.h>
#include <type_traits>

namespace {
namespace ff {

constexpr int HID = 20, LU = 10;
constexpr int G = 3;                       // agents per workgroup
constexpr int UT = 2;                      // 32-slot tiles (60 units + 4 padding slots)
constexpr int NW = 4;                      // wavefronts per workgroup, one per SIMD
constexpr int RB = 2;                      // 32-row blocks per wavefront and tile
constexpr int TILE = NW * RB * 32;         // replay rows per tile
constexpr int KC = 8;                      // k16 steps per W' stage
constexpr int PD = 4;                      // prefetch distance (row steps) of the backward's straight-from-L2 fragments
#ifndef FF_PDF
#define FF_PDF 8
#endif
constexpr int PDF = FF_PDF;                // ... of the forward's (k16 steps)
constexpr int FRAG = 1024;                 // bytes of one operand fragment
constexpr int WSTEP = 2 * UT * FRAG;       // W' bytes per k16 step: [piece][slot tile]
constexpr int STAGE = KC * WSTEP;
constexpr int ZBYTES = 2 * STAGE;          // two W' stages during F; the dz1 planes of a tile during M and B
static_assert((TILE / 16) * WSTEP == ZBYTES, "the dz1 planes of one tile fill the region of the two W' stages");
static_assert(KC % PDF == 0 && (TILE / 16) % PD == 0, "prefetch slots rotate with a fixed period");
constexpr int PC = 24, PLANE = 32 * PC, PANEL_B = 2 * 2 * PLANE * 2;     // per (wavefront, row block): k_mid_fit_v8's A and B planes
constexpr int WF_AGENT = 2 * 2 * 2 * 2 * 32;                             // uint4 of one agent's W2 fragments (both orientations)
constexpr int SV = 44;                     // floats per agent: b2 | W3 | b3 | pad
constexpr int SUMREC = 48;                 // floats per (wavefront, agent): gb1[20] | gW3[20] | gb3 | loss | pad
constexpr int GREC = HID * HID + HID;      // floats per (wavefront, agent): gW2 | gb2, the matrix-core row reduction
constexpr int MAX_K = 768;                 // features (4 wavefronts x 6 feature tiles)
constexpr int SMALL = 3 * HID + HID * HID + 1;      // b1 | W2 | b2 | W3 | b3: the 461 floats behind W1 in a parameter row
constexpr int SMP = 464;                   // ... padded
struct FitRec { static constexpr int gb2 = HID * HID, gW3 = gb2 + HID, gb3 = gW3 + HID, gb1 = gb3 + 1, loss = gb1 + HID, SIZE = loss + 1; };
constexpr float S2 = 1024.f, US2 = 0.0009765625f, RANGE = 65000.f;        // k_mid_fit_v8's scale of W2 / dz2 and its range bound

constexpr int LDS_Z = 0;
constexpr int LDS_PN = LDS_Z + ZBYTES;
constexpr int LDS_WF = LDS_PN + NW * RB * PANEL_B;
constexpr int LDS_SV = LDS_WF + G * WF_AGENT * 16;
constexpr int LDS_B1 = LDS_SV + G * SV * 4;
constexpr int LDS_SUM = LDS_B1 + 64 * 4;
constexpr int LDS_G = LDS_SUM + NW * G * SUMREC * 4;
constexpr int LDS_BYTES = LDS_G + NW * G * GREC * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(G * SMP * 4 <= ZBYTES, "the small arrays are staged in the Z region between steps");

// accumulator row of register q in lane half h (v_mfma_f32_32x32x*: D[(q&3) + 8*(q>>2) + 4*h][lane&31])
__host__ __device__ constexpr int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }
// slot (tile t, row i) -> agent within the group and unit, or a < 0 for the four padding slots
__device__ __forceinline__ void slot_decode(int t, int i, int& a, int& unit) {
  const int h = (i >> 2) & 1, q = (i & 3) + 4 * (i >> 3), sg = 16 * t + q;
  if (sg >= G * LU) { a = -1; unit = 0; return; }
  a = sg / LU;
  unit = v8_unit(h, sg - a * LU);
}

__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }

struct Args {
  const unsigned char* kf; long kf_seed;        // bytes per seed
  const unsigned char* ktf; long ktf_seed;
  unsigned char* wf; long wf_seed;
  const float* alpha;
  float* theta;
  const float* y;
  const int* mask;
  float* loss_out;
  int* flags;
  int S, N, B, in_dim, ldp, ldb, KS, RS, FTILES, NG, nsteps;
  float lr;
};

// ---------------------------------------------------------------------------------------------
// rows x features -> both fragment-major images.  One workgroup = 32 rows x 128 features through LDS.
__global__ __launch_bounds__(256) void k_fit_encode(const float* __restrict__ x, long x_seed_stride, const float* __restrict__ alpha,
                                                    int B, int in_dim, unsigned char* __restrict__ kf, long kf_seed, int KS,
                                                    unsigned char* __restrict__ ktf, long ktf_seed, int RS, int FTILES) {
  __shared__ unsigned short tile[32][128 + 2];
  const int s = blockIdx.z, b0 = blockIdx.y * 32, c0 = blockIdx.x * 128, t = threadIdx.x;
  {
    const int cl = t & 127, c = c0 + cl;
    const float al = c < in_dim ? alpha[c] : 1.f;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int bl = (t >> 7) + 2 * i, b = b0 + bl;
      float kq = 0.f;
      if (b < B && c < in_dim) kq = rintf(x[(long)s * x_seed_stride + (long)b * in_dim + c] / al);
      tile[bl][cl] = (unsigned short)rc_f16_rne(kq);
    }
  }
  __syncthreads();
  unsigned char* kf_s = kf + (long)s * kf_seed;
  unsigned char* ktf_s = ktf + (long)s * ktf_seed;
  const int rt = b0 >> 5;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = t + 256 * q, l = idx & 63;
    {                                                // Kf: fragment (rt, ks), lane l = row l&31, features 16 ks + 8 (l>>5) ..
      const int ksl = idx >> 6, ks = (c0 >> 4) + ksl;
      if (ks < KS) {
        const unsigned short* src = &tile[l & 31][16 * ksl + 8 * (l >> 5)];
        uint4 v;
        v.x = pack2(src[0], src[1]); v.y = pack2(src[2], src[3]); v.z = pack2(src[4], src[5]); v.w = pack2(src[6], src[7]);
        *reinterpret_cast<uint4*>(kf_s + (((long)rt * KS + ks) * 64 + l) * 16) = v;
      }
    }
    {                                                // KTf: fragment (ft, rs), lane l = feature l&31, rows 16 rs + 8 (l>>5) ..
      const int fr = idx >> 6, ftl = fr >> 1, rsl = fr & 1, ft = (c0 >> 5) + ftl, rs = (b0 >> 4) + rsl;
      if (ft < FTILES) {
        const int col = 32 * ftl + (l & 31), r0 = 16 * rsl + 8 * (l >> 5);
        uint4 v;
        v.x = pack2(tile[r0 + 0][col], tile[r0 + 1][col]); v.y = pack2(tile[r0 + 2][col], tile[r0 + 3][col]);
        v.z = pack2(tile[r0 + 4][col], tile[r0 + 5][col]); v.w = pack2(tile[r0 + 6][col], tile[r0 + 7][col]);
        *reinterpret_cast<uint4*>(ktf_s + (((long)ft * RS + rs) * 64 + l) * 16) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Layers 2-3 of ONE agent on NU row blocks of 32 in lockstep (k_mid_fit_v8's block body; see its header in mid_kernels.hip for the
// layout argument).  z1: the lane's ten layer-1 pre-activations (x 2^10, without bias) per block, straight out of the forward
// accumulators.  Outputs: dz1 pieces into the dz planes (zrow[n] = the lane's base address: row step, k-group, element and lane
// half already applied), the row reduction G = [a1 | 1]^T dz2 into g1, the per-lane partial sums of gb1 / gW3 / gb3 / loss.
// TOZ: dz1 goes to the dz planes (the fused fit); otherwise 2^8 dz1 is handed back in dz1o (rcmarl_forward_mid packs it for the
// backward GEMM of the three-launch path).
template <int NU, int AG, bool TOZ = true>
__device__ __forceinline__ void mid_units(const float (&z1)[NU][LU], const bool (&valid)[NU], const float (&ycur)[NU], int B,
                                          const uint4* __restrict__ wfA, const float* __restrict__ sVa,
                                          const float* __restrict__ sB1a, unsigned short* __restrict__ planes,
                                          unsigned char* const (&zrow)[NU], int lane, rc_f32x16& g1, float (&gb1l)[LU],
                                          float (&gw3l)[LU], float& gb3a, float& lossa, float& amax, float (&dz1o)[NU][LU]) {
  const int l31 = lane & 31, half = lane >> 5;
  auto loadA = [&](int prod, int ks) {
    V8Pieces a;
    a.h = wfA[((prod * 2 + ks) * 2 + 0) * 64];
    a.l = wfA[((prod * 2 + ks) * 2 + 1) * 64];
    return a;
  };
  const int tr_off = (8 * half + ((lane & 15) >> 2)) * PC + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int wr8 = l31 * PC + 8 * half, wr2 = l31 * PC + 16 + 2 * half;
  uint4 z4;
  z4.x = z4.y = z4.z = z4.w = 0u;
  const float b3 = sVa[2 * HID];
  float a1l[NU][LU];
  V8Pieces pa0[NU], pa1[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const float zb = fmaf(z1[n][u], RC_F16_W_UNSCALE, sB1a[v8_unit(half, u)]);
      a1l[n][u] = valid[n] ? fmaxf(zb, RC_LEAK * zb) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LU; u += 2) amax = fmaxf(amax, fmaxf(fabsf(a1l[n][u]), fabsf(a1l[n][u + 1])));
    const float x0[8] = {a1l[n][0], a1l[n][1], a1l[n][2], a1l[n][3], a1l[n][4], a1l[n][5], a1l[n][6], a1l[n][7]};
    pa0[n] = v8_split8<false>(x0, 1.f);
    pa1[n].h = z4; pa1[n].l = z4;
    rc_split2h_pair(a1l[n][8], a1l[n][9], pa1[n].h.x, pa1[n].l.x);
  }
  RC_SCHED_FENCE();
  RC_WAVE_SYNC();                                        // the previous agent's transpose reads of the planes are done
  rc_f32x16 zz[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    unsigned short* pA = planes + n * (PANEL_B / 2);
    *reinterpret_cast<uint4*>(pA + 0 * PLANE + wr8) = pa0[n].h;
    *reinterpret_cast<uint4*>(pA + 1 * PLANE + wr8) = pa0[n].l;
    *reinterpret_cast<unsigned*>(pA + 0 * PLANE + wr2) = pa1[n].h.x;
    *reinterpret_cast<unsigned*>(pA + 1 * PLANE + wr2) = pa1[n].l.x;
#pragma unroll
    for (int q = 0; q < 16; ++q) zz[n][q] = 0.f;
  }
  {
    const V8Pieces w1 = loadA(0, 1), w0 = loadA(0, 0);
#pragma unroll
    for (int n = 0; n < NU; ++n) zz[n] = v8_mfma4(w1, pa1[n], zz[n]);
#pragma unroll
    for (int n = 0; n < NU; ++n) zz[n] = v8_mfma4(w0, pa0[n], zz[n]);
  }
  RC_SCHED_FENCE();
  float dz2l[NU][LU];
  V8Pieces pd0[NU], pd1[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    float a2l[LU], vp = 0.f;
#pragma unroll
    for (int u = 0; u < LU; ++u) a2l[u] = rc_lrelu(fmaf(zz[n][u], US2, sVa[v8_unit(half, u)]));
#pragma unroll
    for (int u = 0; u < LU; ++u) vp = fmaf(a2l[u], sVa[HID + v8_unit(half, u)], vp);
    float va = vp, vb = vp;
    rc_swap32(va, vb);
    const float v = (vp + (half ? va : vb)) + b3;
    const float diff = valid[n] ? v - ycur[n] : 0.f;
    const float dv = (2.0f * diff) / (float)B;
    const float dvs = dv * S2;
    if (half == 0) { gb3a += dv; lossa = fmaf(diff, diff, lossa); }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      gw3l[u] = fmaf(a2l[u], dv, gw3l[u]);
      dz2l[n][u] = dvs * sVa[HID + v8_unit(half, u)] * rc_lrelu_grad_from_act(a2l[u]);
    }
#pragma unroll
    for (int u = 0; u < LU; u += 2) amax = fmaxf(amax, fmaxf(fabsf(dz2l[n][u]), fabsf(dz2l[n][u + 1])));
    const float x0[8] = {dz2l[n][0], dz2l[n][1], dz2l[n][2], dz2l[n][3], dz2l[n][4], dz2l[n][5], dz2l[n][6], dz2l[n][7]};
    pd0[n] = v8_split8<false>(x0, 1.f);
    pd1[n].h = z4; pd1[n].l = z4;
    rc_split2h_pair(dz2l[n][8], dz2l[n][9], pd1[n].h.x, pd1[n].l.x);
    unsigned short* pB = planes + n * (PANEL_B / 2) + 2 * PLANE;
    *reinterpret_cast<uint4*>(pB + 0 * PLANE + wr8) = pd0[n].h;
    *reinterpret_cast<uint4*>(pB + 1 * PLANE + wr8) = pd0[n].l;
    *reinterpret_cast<unsigned*>(pB + 0 * PLANE + wr2) = pd1[n].h.x;
    *reinterpret_cast<unsigned*>(pB + 1 * PLANE + wr2) = pd1[n].l.x;
  }
  rc_f32x16 dd[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n)
#pragma unroll
    for (int q = 0; q < 16; ++q) dd[n][q] = 0.f;
  {
    const V8Pieces w1 = loadA(1, 1), w0 = loadA(1, 0);
#pragma unroll
    for (int n = 0; n < NU; ++n) dd[n] = v8_mfma4(w1, pd1[n], dd[n]);
#pragma unroll
    for (int n = 0; n < NU; ++n) dd[n] = v8_mfma4(w0, pd0[n], dd[n]);
  }
  RC_SCHED_FENCE();
  // ---- the row reduction: operands read back TRANSPOSED from the planes
  RC_WAVE_SYNC();
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    const unsigned short* pA = planes + n * (PANEL_B / 2);
    const unsigned short* pB = pA + 2 * PLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      V8Pieces ra, rb;
      const int o = tr_off + 16 * PC * ks;
      uint2 t0, t1;
      t0 = rc_lds_read_tr16(pA + 0 * PLANE + o); t1 = rc_lds_read_tr16(pA + 0 * PLANE + o + 4 * PC);
      ra.h.x = t0.x; ra.h.y = t0.y; ra.h.z = t1.x; ra.h.w = t1.y;
      t0 = rc_lds_read_tr16(pA + 1 * PLANE + o); t1 = rc_lds_read_tr16(pA + 1 * PLANE + o + 4 * PC);
      ra.l.x = t0.x; ra.l.y = t0.y; ra.l.z = t1.x; ra.l.w = t1.y;
      t0 = rc_lds_read_tr16(pB + 0 * PLANE + o); t1 = rc_lds_read_tr16(pB + 0 * PLANE + o + 4 * PC);
      rb.h.x = t0.x; rb.h.y = t0.y; rb.h.z = t1.x; rb.h.w = t1.y;
      t0 = rc_lds_read_tr16(pB + 1 * PLANE + o); t1 = rc_lds_read_tr16(pB + 1 * PLANE + o + 4 * PC);
      rb.l.x = t0.x; rb.l.y = t0.y; rb.l.z = t1.x; rb.l.w = t1.y;
      g1 = v8_mfma4(ra, rb, g1);
    }
  }
  // ---- dz1 = da1 * lrelu'(a1), carried as 2^8 dz1; its two f16 pieces go to the dz planes in B-fragment order:
  // fragment (row step, piece, slot tile), lane (slot row, k-group), element = row & 7
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    float dz1l[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      dz1l[u] = (dd[n][u] * (US2 * US2 * RC_F16_DZ_SCALE)) * rc_lrelu_grad_from_act(a1l[n][u]);
      gb1l[u] += dz1l[u];
    }
#pragma unroll
    for (int u = 0; u < LU; u += 2) amax = fmaxf(amax, fmaxf(fabsf(dz1l[u]), fabsf(dz1l[u + 1])));
    if constexpr (!TOZ) {
#pragma unroll
      for (int u = 0; u < LU; ++u) dz1o[n][u] = dz1l[u];
    } else
#pragma unroll
    for (int qq = 0; qq < LU / 2; ++qq) {
      unsigned ph, pl;
      rc_split2h_pair(dz1l[2 * qq], dz1l[2 * qq + 1], ph, pl);
      const int sg0 = LU * AG + 2 * qq, sg1 = sg0 + 1;
      const int o0 = ((sg0 >> 4) * 64 + acc_row(sg0 & 15, 0)) * 16, o1 = ((sg1 >> 4) * 64 + acc_row(sg1 & 15, 0)) * 16;
      *reinterpret_cast<unsigned short*>(zrow[n] + o0) = (unsigned short)ph;
      *reinterpret_cast<unsigned short*>(zrow[n] + o0 + UT * FRAG) = (unsigned short)pl;
      *reinterpret_cast<unsigned short*>(zrow[n] + o1) = (unsigned short)(ph >> 16);
      *reinterpret_cast<unsigned short*>(zrow[n] + o1 + UT * FRAG) = (unsigned short)(pl >> 16);
    }
  }
}

#ifdef RCMARL_EMU
#define RC_FF_OCC
#else
#define RC_FF_OCC __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1)))
#endif

// FTW: feature tiles per wavefront in the backward (FTILES <= 4 FTW); NU: row blocks the mid step runs in lockstep (1 or 2)
template <int FTW, int NU>
__global__ RC_FF_OCC void k_fit_fused(const Args A) {
  static_assert(NU == 1 || NU == 2, "row blocks in lockstep");
  RCMARL_DYN_SMEM(unsigned char, lds);
  unsigned char* Z = lds + LDS_Z;
  unsigned char* sPn = lds + LDS_PN;
  uint4* sWf = reinterpret_cast<uint4*>(lds + LDS_WF);
  float* sV = reinterpret_cast<float*>(lds + LDS_SV);
  float* sB1 = reinterpret_cast<float*>(lds + LDS_B1);
  float* sSum = reinterpret_cast<float*>(lds + LDS_SUM);
  float* sG = reinterpret_cast<float*>(lds + LDS_G);
  float* sSm = reinterpret_cast<float*>(lds + LDS_Z);     // (the Z region is idle between the last tile of a step and the first of the next)

  int s, g;
  if ((A.S & 7) == 0) {                                  // all groups of a seed on one XCD (workgroup b -> XCD b % 8)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    s = xcd + 8 * (q / A.NG);
    g = q % A.NG;
  } else {
    s = blockIdx.x / A.NG;
    g = blockIdx.x - s * A.NG;
  }
  const int r = threadIdx.x, lane = r & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ag0 = g * G, N = A.N, B = A.B, KS = A.KS, in_dim = A.in_dim, ldp = A.ldp;
  const NetGeom geo = make_geom(in_dim, HID, 1);
  float* theta_s = A.theta + (long)s * N * ldp;
  const int NS = 2 * ((KS + 2 * KC - 1) / (2 * KC)), KSP = NS * KC;      // stages (an even number: they alternate between two
  //                                                                         register sets); k16 steps incl. the zero padding
  unsigned char* wf_wg = A.wf + (long)s * A.wf_seed + (long)g * KSP * WSTEP;
  const unsigned char* kf_s = A.kf + (long)s * A.kf_seed;
  const unsigned char* ktf_s = A.ktf + (long)s * A.ktf_seed;
  const int ntiles = (B + TILE - 1) / TILE;
  const float lr = A.lr, lr_dz = A.lr * RC_F16_DZ_UNSCALE;
  rc_f16_saturate();

  bool live[G], upd[G];
#pragma unroll
  for (int a = 0; a < G; ++a) {
    live[a] = ag0 + a < N;
    upd[a] = live[a] && (A.mask == nullptr || A.mask[live[a] ? ag0 + a : 0] != 0);
  }

  // this lane's column of a gradient tile: slot (t, l31) -> parameter row and unit
  int col_a[UT], col_unit[UT];
  bool col_ok[UT], col_upd[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    slot_decode(t, l31, col_a[t], col_unit[t]);
    col_ok[t] = col_a[t] >= 0 && ag0 + col_a[t] < N;
    col_upd[t] = col_ok[t] && (A.mask == nullptr || A.mask[col_ok[t] ? ag0 + col_a[t] : 0] != 0);
  }
  bool wflag = false;                                    // a W' piece of this lane's column would saturate

  rc_f32x16 gacc[FTW][UT];
  // W1 <- W1 - lr alpha_k gW1 (UPDATE) and the f16 pieces of 2^10 alpha_k W1 into Wf, for this wavefront's feature tiles
  auto w1_pass = [&](auto update_tag) {
    constexpr bool UPDATE = decltype(update_tag)::value;
    // every address below is formed from these two, which the optimiser cannot see through: otherwise loop-invariant code
    // motion lifts the whole address arithmetic of this pass (hundreds of registers) out of the step loop and keeps it alive
    const int wv = rc_opaque_s(wave), hf = rc_opaque_v(half);
    // Two phases: ALL old weights of the wavefront's tiles are requested first (independent loads, one round trip), then the
    // updates are computed and stored.
    float wold[FTW][UT][16];
#pragma unroll
    for (int f = 0; f < FTW; ++f) {
      const int ft = wv * FTW + f, k0 = 32 * ft + 4 * hf;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const float* th = theta_s + (long)(ag0 + (col_ok[t] ? col_a[t] : 0)) * ldp + col_unit[t] + (long)k0 * HID;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          wold[f][t][q] = (ft < A.FTILES && col_ok[t] && k0 + acc_row(q, 0) < in_dim) ? th[acc_row(q, 0) * HID] : 0.f;
      }
    }
#pragma unroll
    for (int f = 0; f < FTW; ++f) {
      const int ft = wv * FTW + f;
      if (ft < A.FTILES) {
        const int k0 = 32 * ft + 4 * hf;                 // the lane's first feature of this tile; register q adds acc_row(q, 0)
        float alq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) alq[q] = k0 + acc_row(q, 0) < in_dim ? A.alpha[k0 + acc_row(q, 0)] : 0.f;
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          float* th = theta_s + (long)(ag0 + (col_ok[t] ? col_a[t] : 0)) * ldp + col_unit[t] + (long)k0 * HID;
          unsigned char* wf_t = wf_wg + (long)(2 * ft) * WSTEP + t * FRAG + l31 * 16 + 8 * hf;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            float wn4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 4 * gq + e;
              float w = wold[f][t][q];
              if (UPDATE) {
                if (col_upd[t] && k0 + acc_row(q, 0) < in_dim) {
                  w = w - lr_dz * (alq[q] * gacc[f][t][q]);
                  th[acc_row(q, 0) * HID] = w;
                }
              }
              wn4[e] = (w * alq[q]) * RC_F16_W_SCALE;
              if (fabsf(wn4[e]) > RANGE) wflag = true;
            }
            unsigned h0, l0, h1, l1;
            rc_split2h_pair(wn4[0], wn4[1], h0, l0);
            rc_split2h_pair(wn4[2], wn4[3], h1, l1);
            // features 32 ft + 8 gq + 4 half + (0..3): k16 step 2 ft + (gq >> 1), k-group gq & 1, this lane's half of the chunk
            unsigned char* q8 = wf_t + (gq >> 1) * WSTEP + (gq & 1) * 512;
            *reinterpret_cast<uint2*>(q8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(q8 + UT * FRAG) = make_uint2(l0, l1);
          }
        }
      }
    }
  };
  w1_pass(std::false_type{});
  for (int e = r; e < G * SMP; e += 256) {
    const int a = e / SMP, i = e - a * SMP;
    sSm[e] = (i < SMALL && ag0 + a < N) ? theta_s[(long)(ag0 + a) * ldp + geo.o_b1 + i] : 0.f;
  }
  for (int e = r; e < (KSP - KS) * (WSTEP / 16); e += 256) {        // W' is zero on the padding steps of the last stage
    uint4 z;
    z.x = z.y = z.z = z.w = 0u;
    *reinterpret_cast<uint4*>(wf_wg + (long)KS * WSTEP + (long)e * 16) = z;
  }
  // feature tiles beyond the last wavefront's share but inside KS (KS = 2 FTILES: none) -- nothing to zero
  __threadfence();

  // W' reaches LDS through registers: a wavefront carries its quarter of TWO stages (wreg[0]: the next even stage, wreg[1]: the
  // next odd one); a set is requested from L2 two stages before it is written to LDS.  (LDS-DMA was measured slower here: an
  // LDS-DMA instruction blocks the wavefront's issue for 60-180 cycles, and with one wavefront per SIMD nobody fills them.)
  uint4 wreg[2][KC];
  uint4 xb[PDF][RB];                                     // the row fragments of the next PDF steps of the forward reduction
  auto load_wset = [&](auto set_tag, int st, int lane_x) {
    constexpr int SET = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < KC; ++i) wreg[SET][i] = ld_u4(wf_wg + (((long)(st * KC + i) * 4 + wave) * 64 + lane_x) * 16);
  };
  // what the forward phase of `tile` starts from: stages 0 and 1 of W' and the row fragments of its first PD steps
  auto f_prefetch = [&](int tile) {
    const int lane_x = rc_opaque_v(lane);
    load_wset(std::integral_constant<int, 0>{}, 0, lane_x);
    load_wset(std::integral_constant<int, 1>{}, 1, lane_x);
    const unsigned char* kf_w = kf_s + ((long)(tile * (TILE / 32) + wave * RB) * KS * 64 + lane_x) * 16;
#pragma unroll
    for (int d = 0; d < PDF; ++d)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xb[d][rb] = ld_u4(kf_w + ((long)rb * KS + min(d, KS - 1)) * FRAG);
  };
  float amax[G];
#pragma unroll
  for (int a = 0; a < G; ++a) amax[a] = 0.f;

#ifdef FF_TIMING
  long long tmF = 0, tmM = 0, tmB = 0, tmE = 0, tm0 = wall_clock64(), tm1;
#define FF_TICK(acc) do { tm1 = wall_clock64(); acc += tm1 - tm0; tm0 = tm1; } while (0)
#else
#define FF_TICK(acc) ((void)0)
#endif
  for (int step = 0; step < A.nsteps; ++step) {
    // ---- the agents' small arrays, from their image in LDS (sSm: b1 | W2 | b2 | W3 | b3 per agent; loaded once before the first
    // step, kept current by the step end): W2 as f16 pieces in A-fragment order (both orientations), b2 | W3 | b3, b1
    __syncthreads();
    for (int e = r; e < G * 2 * 32 * 32; e += 256) {
      const int a = e >> 11, e2 = e & 2047, prod = e2 >> 10, ri = (e2 >> 5) & 31, k = e2 & 31;
      const int ui = v8_row_unit(ri), uk = v8_slot_unit(k);
      float w = 0.f;
      if (ui >= 0 && uk >= 0) w = sSm[a * SMP + HID + (prod == 0 ? uk * HID + ui : ui * HID + uk)];
      unsigned ph, pl;
      rc_split2h_pair(w * S2, 0.f, ph, pl);
      if (fabsf(w) * S2 > RANGE) A.flags[s * N + ag0 + a] = 1;
      const int ks = k >> 4, kg = (k >> 3) & 1;
      unsigned short* wf16 = reinterpret_cast<unsigned short*>(sWf + a * WF_AGENT);
      const int base = ((((prod * 2 + ks) * 2 + 0) * 2 + kg) * 32 + ri) * 8 + (k & 7);
      wf16[base] = (unsigned short)ph;
      wf16[base + 2 * 32 * 8] = (unsigned short)pl;
    }
    for (int e = r; e < G * SV; e += 256) {
      const int a = e / SV, i = e - a * SV;
      sV[e] = i < 2 * HID + 1 ? sSm[a * SMP + HID + HID * HID + i] : 0.f;
    }
    if (r < 64) sB1[r] = r < G * HID ? sSm[(r / HID) * SMP + r % HID] : 0.f;
    for (int e = r; e < NW * G * SUMREC; e += 256) sSum[e] = 0.f;
    // column 20 of the A planes is the constant 1 (-> gb2), the other spare columns are zero (as k_mid_fit_v8)
    for (int e = lane; e < RB * 2 * 32 * (PC - 20); e += 64) {
      const int n = e / (2 * 32 * (PC - 20)), e2 = e - n * (2 * 32 * (PC - 20));
      const int pc = e2 / (32 * (PC - 20)), rw = (e2 / (PC - 20)) & 31, cl = 20 + e2 % (PC - 20);
      unsigned short* pA = reinterpret_cast<unsigned short*>(sPn + (wave * RB + n) * PANEL_B);
      pA[pc * PLANE + rw * PC + cl] = (pc == 0 && cl == 20) ? (unsigned short)0x3C00 : (unsigned short)0;
    }
#pragma unroll
    for (int f = 0; f < FTW; ++f)
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) gacc[f][t][q] = 0.f;
    for (int e = r; e < NW * G * GREC; e += 256) sG[e] = 0.f;
    __syncthreads();                                     // (also: sSm is read out, the Z region belongs to the tiles again)
    f_prefetch(0);

    for (int tile = 0; tile < ntiles; ++tile) {
      // ================= F: z1^T = W' K^T for this wavefront's 64 rows ===================================================
      rc_f32x16 acc[UT][RB];
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[t][rb][q] = 0.f;
      const int lane_f = rc_opaque_v(lane);
      const unsigned char* kf_w = kf_s + ((long)(tile * (TILE / 32) + wave * RB) * KS * 64 + lane_f) * 16;
      // No branch inside the reduction loop (hipcc's s_waitcnt pass gives up its load bookkeeping at every control-flow join and
      // waits for vmcnt(0)): the reduction runs over KSP = NS * KC steps -- Wf is zero beyond KS --, and a fragment request past
      // the end re-reads the last one.
      // One stage: KC k16 steps from LDS buffer BUF while this wavefront's quarter of the NEXT stage goes from register set
      // 1 - BUF to the other buffer, one 1-KiB burst per step, each register then re-requested with the stage after that.
      unsigned char* zst = Z + (wave * 64 + lane_f) * 16;                // + buffer * STAGE + burst * 4 KiB
      auto stage = [&](auto buf_tag, int st) {
        constexpr int BUF = decltype(buf_tag)::value;
        const unsigned char* stg = Z + BUF * STAGE + lane_f * 16;
        const int st2 = (st + 3) % NS;                                 // (past the end: next tile's stages -- same W')
        uint4 afA[2][UT], afB[2][UT];
        auto ldsA = [&](int ksl, uint4 (&af)[2][UT]) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < UT; ++t) af[p][t] = ld_u4(stg + ((ksl * 2 + p) * UT + t) * FRAG);
        };
        // One k16 step: the NEXT step's W' fragments are requested from LDS before this step's eight matrix-core instructions
        // and the step-after-PD's row fragments from L2 behind them (straight into the slot their readers were just issued
        // from); the scheduling fences keep hipcc from hoisting every load of the unrolled stage to its top.
        auto kstep = [&](int ksl, uint4 (&cur)[2][UT], uint4 (&nxt)[2][UT]) {
          const int ks = st * KC + ksl;
          if (ksl + 1 < KC) ldsA(ksl + 1, nxt);
#pragma unroll
          for (int p = 1; p >= 0; --p)                   // smallest pieces first
#pragma unroll
            for (int t = 0; t < UT; ++t)
#pragma unroll
              for (int rb = 0; rb < RB; ++rb) acc[t][rb] = rc_mfma_f16(cur[p][t], xb[ksl % PDF][rb], acc[t][rb]);
          RC_SCHED_FENCE();
          const int kn = min(ks + PDF, KS - 1);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) xb[ksl % PDF][rb] = ld_u4(kf_w + ((long)rb * KS + kn) * FRAG);
          *reinterpret_cast<uint4*>(zst + (1 - BUF) * STAGE + ksl * (4 * FRAG)) = wreg[1 - BUF][ksl];
          wreg[1 - BUF][ksl] = ld_u4(wf_wg + (((long)(st2 * KC + ksl) * 4 + wave) * 64 + lane_f) * 16);
          RC_SCHED_FENCE();
        };
        ldsA(0, afA);
#pragma unroll
        for (int ksl = 0; ksl < KC; ksl += 2) {
          kstep(ksl, afA, afB);
          kstep(ksl + 1, afB, afA);
        }
        __syncthreads();                                 // buffer 1 - BUF is written, buffer BUF is read out
      };
      // stage 0 of W' into buffer 0; its registers then take stage 2
#pragma unroll
      for (int i = 0; i < KC; ++i) *reinterpret_cast<uint4*>(zst + i * (4 * FRAG)) = wreg[0][i];
      {
        const int st2 = 2 % NS;
#pragma unroll
        for (int i = 0; i < KC; ++i) wreg[0][i] = ld_u4(wf_wg + (((long)(st2 * KC + i) * 4 + wave) * 64 + lane_f) * 16);
      }
      __syncthreads();
      for (int st = 0; st < NS; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        stage(std::integral_constant<int, 1>{}, st + 1);
      }
      FF_TICK(tmF);                                      // (the last stage ended with a barrier: Z becomes this tile's dz planes)

      // ================= M: layers 2-3 of the three agents on this wavefront's two row blocks ===========================
      {
        // (lane-derived addresses and predicates of this phase are formed per tile from an opaque copy of the lane id: left to
        // loop-invariant code motion they would all be lifted out of the tile loop and held in registers through F and B)
        const int lane_m = rc_opaque_v(lane);
        const int lane = lane_m, l31 = lane_m & 31, half = lane_m >> 5;
        const int brow = tile * TILE + wave * (RB * 32);
        bool valid[RB];
        unsigned char* zrow[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int rl = wave * (RB * 32) + rb * 32 + l31;       // row within the tile
          valid[rb] = brow + rb * 32 + l31 < B;
          zrow[rb] = Z + (rl >> 4) * WSTEP + ((rl >> 3) & 1) * 512 + (rl & 7) * 2 + 64 * half;
        }
        float yv[G][RB];
#pragma unroll
        for (int a = 0; a < G; ++a)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const int b = brow + rb * 32 + l31;
            yv[a][rb] = (upd[a] && valid[rb]) ? A.y[((long)s * N + ag0 + a) * A.ldb + b] : 0.f;
          }
        unsigned short* planes = reinterpret_cast<unsigned short*>(sPn + wave * RB * PANEL_B);
        auto agent = [&](auto ag_tag) {
          constexpr int AG = decltype(ag_tag)::value;
          if (!upd[AG]) return;                          // (workgroup-uniform; its dz slots keep stale bytes: columns nobody reads)
          float gb1l[LU], gw3l[LU], gb3a = 0.f, lossa = 0.f;
#pragma unroll
          for (int u = 0; u < LU; ++u) gb1l[u] = gw3l[u] = 0.f;
          rc_f32x16 g1;
#pragma unroll
          for (int q = 0; q < 16; ++q) g1[q] = 0.f;
          const uint4* wfA = sWf + AG * WF_AGENT + half * 32 + l31;
          if constexpr (NU == 2) {
            float z1[2][LU];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
              for (int u = 0; u < LU; ++u) z1[rb][u] = acc[(LU * AG + u) >> 4][rb][(LU * AG + u) & 15];
            const float yc[2] = {yv[AG][0], yv[AG][1]};
            float nodz[2][LU];
            mid_units<2, AG>(z1, valid, yc, B, wfA, sV + AG * SV, sB1 + AG * HID, planes, zrow, lane, g1, gb1l, gw3l, gb3a,
                             lossa, amax[AG], nodz);
          } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
              float z1[1][LU];
#pragma unroll
              for (int u = 0; u < LU; ++u) z1[0][u] = acc[(LU * AG + u) >> 4][rb][(LU * AG + u) & 15];
              const bool v1[1] = {valid[rb]};
              const float yc[1] = {yv[AG][rb]};
              unsigned char* const zr[1] = {zrow[rb]};
              float nodz[1][LU];
              mid_units<1, AG>(z1, v1, yc, B, wfA, sV + AG * SV, sB1 + AG * HID, planes, zr, lane, g1, gb1l, gw3l, gb3a,
                               lossa, amax[AG], nodz);
            }
          }
          // the row reduction of this tile joins the wavefront's running record (element q of the lane's accumulator tile is
          // G[acc_row(q, half)][l31]: rows 0..19 = gW2, row 20 = gb2; one owner per address)
          {
            float* gr = sG + (wave * G + AG) * GREC;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int ii = acc_row(q, half);
              if (ii <= HID && l31 < HID) gr[ii * HID + l31] += g1[q] * US2;
            }
          }
          // what is summed over rows outside the matrix core: over the 32 lanes of each half, results in lanes 31 and 63
          float sm[2 * LU + 4];
#pragma unroll
          for (int u = 0; u < LU; ++u) { sm[u] = gb1l[u] * RC_F16_DZ_UNSCALE; sm[LU + u] = gw3l[u]; }
          sm[2 * LU] = gb3a; sm[2 * LU + 1] = lossa; sm[2 * LU + 2] = sm[2 * LU + 3] = 0.f;
#pragma unroll
          for (int q = 0; q < (2 * LU + 4) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
          if (l31 == 31) {
            float* rec = sSum + (wave * G + AG) * SUMREC;
#pragma unroll
            for (int u = 0; u < LU; ++u) {
              rec[v8_unit(half, u)] += sm[u];
              rec[HID + v8_unit(half, u)] += sm[LU + u];
            }
            if (half == 0) { rec[2 * HID] += sm[2 * LU]; rec[2 * HID + 1] += sm[2 * LU + 1]; }
          }
        };
        agent(std::integral_constant<int, 0>{});
        agent(std::integral_constant<int, 1>{});
        agent(std::integral_constant<int, 2>{});
      }
      __syncthreads();                                   // every wavefront's dz1 pieces are in the planes
      FF_TICK(tmM);

      // ================= B: gW1 += K^T dz1 over this tile's 256 rows ====================================================
      {
        // (no branches here either: a feature tile beyond the last one re-reads the last one, and its accumulators are never
        // looked at; a request past the tile's last row step re-reads that step)
        f_prefetch(min(tile + 1, ntiles - 1));           // what the next tile's forward phase starts from (registers; harmless after the last tile)
        int ftc[FTW];
#pragma unroll
        for (int f = 0; f < FTW; ++f) ftc[f] = min(wave * FTW + f, A.FTILES - 1);
        uint4 kb[PD][FTW];
        const int lane_b = rc_opaque_v(lane);
        const unsigned char* ktf_w = ktf_s + ((long)tile * (TILE / 16) * 64 + lane_b) * 16;     // + (ft * RS + rs) * FRAG
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
          for (int f = 0; f < FTW; ++f) kb[d][f] = ld_u4(ktf_w + ((long)ftc[f] * A.RS + d) * FRAG);
        const unsigned char* zl = Z + lane_b * 16;
        uint4 bfA[2][UT], bfB[2][UT];
        auto ldsB = [&](int rs, uint4 (&bf)[2][UT]) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < UT; ++t) bf[p][t] = ld_u4(zl + ((rs * 2 + p) * UT + t) * FRAG);
        };
        auto bstep = [&](int rs0, int j, uint4 (&cur)[2][UT], uint4 (&nxt)[2][UT]) {      // row step rs0 + j, j = its prefetch slot
          const int rs = rs0 + j;
          ldsB(min(rs + 1, TILE / 16 - 1), nxt);
#pragma unroll
          for (int p = 1; p >= 0; --p)
#pragma unroll
            for (int f = 0; f < FTW; ++f)
#pragma unroll
              for (int t = 0; t < UT; ++t) gacc[f][t] = rc_mfma_f16(kb[j][f], cur[p][t], gacc[f][t]);
          RC_SCHED_FENCE();
          const int rn = min(rs + PD, TILE / 16 - 1);    // (behind its readers, straight into the slot: see the forward loop)
#pragma unroll
          for (int f = 0; f < FTW; ++f) kb[j][f] = ld_u4(ktf_w + ((long)ftc[f] * A.RS + rn) * FRAG);
          RC_SCHED_FENCE();
        };
        static_assert(PD == 4, "the row-step loop is unrolled by the prefetch distance");
        ldsB(0, bfA);
        for (int rs0 = 0; rs0 < TILE / 16; rs0 += PD) {
          bstep(rs0, 0, bfA, bfB);
          bstep(rs0, 1, bfB, bfA);
          bstep(rs0, 2, bfA, bfB);
          bstep(rs0, 3, bfB, bfA);
        }
      }
      __syncthreads();                                   // the planes are read out: Z takes the next tile's W' stages
      FF_TICK(tmB);
    }

    // ================= step end: records -> small gradients, SGD on every array of the three agents ========================
    {
      for (int e = r; e < G * FitRec::SIZE; e += 256) {
        const int a = e / FitRec::SIZE, idx = e - a * FitRec::SIZE;
        float sum = 0.f;
        if (idx < FitRec::gW3) {
#pragma unroll
          for (int w = 0; w < NW; ++w) sum += sG[(w * G + a) * GREC + idx];
        } else {
          int si;
          if (idx < FitRec::gb3) si = HID + (idx - FitRec::gW3);
          else if (idx == FitRec::gb3) si = 2 * HID;
          else if (idx < FitRec::loss) si = idx - FitRec::gb1;
          else si = 2 * HID + 1;
#pragma unroll
          for (int w = 0; w < NW; ++w) sum += sSum[(w * G + a) * SUMREC + si];
        }
        const bool up = a == 0 ? upd[0] : (a == 1 ? upd[1] : upd[2]);      // (no dynamic index into a register array)
        if (idx == FitRec::loss) {
          if (up && step == 0 && A.loss_out) A.loss_out[(long)s * N + ag0 + a] = sum / (float)B;
        } else {
          int o;                                         // offset behind W1: b1 | W2 | b2 | W3 | b3
          if (idx < FitRec::gb2) o = HID + idx;
          else if (idx < FitRec::gW3) o = HID + HID * HID + (idx - FitRec::gb2);
          else if (idx < FitRec::gb3) o = 2 * HID + HID * HID + (idx - FitRec::gW3);
          else if (idx < FitRec::gb1) o = 3 * HID + HID * HID;
          else o = idx - FitRec::gb1;
          // the new value goes to global memory and to the image in LDS the next step builds its operands from (the Z region is
          // idle from here to the next step's first tile)
          float* th = theta_s + (long)(ag0 + a) * ldp + geo.o_b1 + o;
          float v = ag0 + a < N ? *th : 0.f;
          if (up) { v = v - lr * sum; *th = v; }
          sSm[a * SMP + o] = v;
        }
      }
      w1_pass(std::true_type{});
      __threadfence();                                   // the next step's loads (other lanes, LDS-DMA) see these stores
      FF_TICK(tmE);
    }
  }
  __syncthreads();
#ifdef FF_TIMING
  if (r == 0 && A.loss_out && ag0 + 2 < N) {            // measurement builds only: 10-ns ticks per phase, all steps (E includes the small-array loads)
    float* o = A.loss_out + (long)s * N + ag0;
    o[0] = (float)tmF; o[1] = (float)tmM; o[2] = (float)tmB;
    if (g == 0) o[3] = (float)tmE;
  }
#endif
#pragma unroll
  for (int a = 0; a < G; ++a)
    if (live[a] && amax[a] > RANGE) A.flags[s * N + ag0 + a] = 1;
#pragma unroll
  for (int t = 0; t < UT; ++t)
    if (wflag && col_ok[t]) A.flags[s * N + ag0 + col_a[t]] = 1;
}


// =============================================================================================================================
// Forward + mid in one launch ("rcmarl_forward_mid"): the F and M phases above as a kernel of their own, one workgroup per
// (seed, three agents, 256 replay rows), SEVERAL workgroups per CU -- while one workgroup's wavefronts run the reduction loop
// (matrix core, LDS, L2) another's run layers 2-3 (vector ALUs): the layer-1 activations never reach HBM (a1t: 1 GB written and
// read back per step at BASELINE configs[3]) and the two phases overlap across workgroups.  Outputs are those of
// rcmarl_mid_fit_lattice: dz1 as two f16 pieces of 2^8 dz1 in the packed layout the backward GEMM reads, and one
// partial-gradient record per (seed, agent, 256-row tile).  Operands: Kf (rcmarl_fit_encode), Wf (rcmarl_fit_wf_split, or the
// backward GEMM's epilogue: rcmarl_layer1_backward_sgd_lattice_wf), the agents' W2 fragments (rcmarl_fit_w2_frags).
namespace fm {
#ifndef RC_FM_WAVES
#define RC_FM_WAVES 2
#endif
constexpr int RC_FM_WAVES_ = RC_FM_WAVES;
#ifndef FM_KC2
#define FM_KC2 4
#endif
#ifndef FM_RECS_PER_WAVE
#define FM_RECS_PER_WAVE 0
#endif
constexpr int KC2 = FM_KC2;                             // k16 steps per W' stage (two LDS buffers of KC2 * 4 KiB)
constexpr bool RECS_PER_WAVE = FM_RECS_PER_WAVE != 0;   // every wavefront leaves its own record (no staging in LDS: 4 x the records)
constexpr int STAGE2 = KC2 * WSTEP;
constexpr int PD2 = KC2 < 4 ? KC2 : 4;
constexpr int REC = 464;                                // floats of one staged record (>= FitRec::SIZE)
constexpr int LDS_A = 0;                                // F: the two W' stages; M: the wavefronts' planes, then the staged records
constexpr int A_BYTES = NW * PANEL_B + (RECS_PER_WAVE ? 0 : NW * G * REC * 4);
static_assert(A_BYTES >= 2 * STAGE2, "the stage buffers fit the aliased region");
constexpr int LDS_WF2 = LDS_A + A_BYTES;
constexpr int LDS_SV2 = LDS_WF2 + G * WF_AGENT * 16;
constexpr int LDS_B12 = LDS_SV2 + G * SV * 4;
constexpr int LDS_BYTES2 = LDS_B12 + 64 * 4;
static_assert(RC_FM_WAVES_ * LDS_BYTES2 <= 160 * 1024, "the workgroups of a CU fit its LDS");

struct Args2 {
  const unsigned char* kf; long kf_seed;
  const unsigned char* wf; long wf_seed;
  const uint4* w2f;                                     // [S][N][WF_AGENT]
  const float* theta;
  const float* y;
  float* partials;                                      // [S][N][ntiles][FitRec::SIZE]
  unsigned char* dzp; int dzp_rt, dzp_kt;
  int* flags;
  int S, N, B, in_dim, ldp, ldb, KS, NG, ntiles;
};

// the agents' 2^10 W2 as f16 pieces in A-fragment order, both orientations (k_mid_fit_v8's sWf image): 8 KiB per agent
__global__ __launch_bounds__(256) void k_w2_frags(const float* __restrict__ theta, uint4* __restrict__ w2f, int* __restrict__ flags,
                                                  int N, int in_dim, int ldp) {
  const int s = blockIdx.y, i = blockIdx.x, r = threadIdx.x;
  const NetGeom geo = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  unsigned short* wf16 = reinterpret_cast<unsigned short*>(w2f + ((long)s * N + i) * WF_AGENT);
  rc_f16_saturate();
  for (int e = r; e < 2 * 32 * 32; e += 256) {
    const int prod = e >> 10, ri = (e >> 5) & 31, k = e & 31;
    const int ui = v8_row_unit(ri), uk = v8_slot_unit(k);
    float w = 0.f;
    if (ui >= 0 && uk >= 0) w = prod == 0 ? th[geo.o_W2 + uk * HID + ui] : th[geo.o_W2 + ui * HID + uk];
    unsigned ph, pl;
    rc_split2h_pair(w * S2, 0.f, ph, pl);
    if (fabsf(w) * S2 > RANGE) flags[s * N + i] = 1;
    const int ks = k >> 4, kg = (k >> 3) & 1;
    const int base = ((((prod * 2 + ks) * 2 + 0) * 2 + kg) * 32 + ri) * 8 + (k & 7);
    wf16[base] = (unsigned short)ph;
    wf16[base + 2 * 32 * 8] = (unsigned short)pl;
  }
}

// theta -> Wf: the two f16 pieces of 2^10 alpha_k W1 of every agent group in fragment order (what the fused fit's prologue writes)
__global__ __launch_bounds__(256) void k_wf_split(const float* __restrict__ theta, const float* __restrict__ alpha,
                                                  unsigned char* __restrict__ wf, long wf_seed, int* __restrict__ flags, int N,
                                                  int in_dim, int ldp, int KS, int KSP, int FTILES) {
  const int s = blockIdx.y, g = blockIdx.x, r = threadIdx.x, lane = r & 63, wave = r >> 6, l31 = lane & 31, half = lane >> 5;
  const int ag0 = g * G;
  unsigned char* wf_wg = wf + (long)s * wf_seed + (long)g * KSP * WSTEP;
  rc_f16_saturate();
  bool wflag = false;
  int col_a[UT], col_unit[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) slot_decode(t, l31, col_a[t], col_unit[t]);
  for (int ft = wave; ft < FTILES; ft += NW) {
    const int k0 = 32 * ft + 4 * half;
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const bool ok = col_a[t] >= 0 && ag0 + col_a[t] < N;
      const float* th = theta + ((long)s * N + ag0 + (ok ? col_a[t] : 0)) * ldp + col_unit[t] + (long)k0 * HID;
      unsigned char* wf_t = wf_wg + (long)(2 * ft) * WSTEP + t * FRAG + l31 * 16 + 8 * half;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float wn4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + 8 * gq + e;
          const float w = (ok && k < in_dim) ? th[(8 * gq + e) * HID] : 0.f;
          wn4[e] = (w * (k < in_dim ? alpha[k] : 0.f)) * RC_F16_W_SCALE;
          if (fabsf(wn4[e]) > RANGE) wflag = true;
        }
        unsigned h0, l0, h1, l1;
        rc_split2h_pair(wn4[0], wn4[1], h0, l0);
        rc_split2h_pair(wn4[2], wn4[3], h1, l1);
        unsigned char* q8 = wf_t + (gq >> 1) * WSTEP + (gq & 1) * 512;
        *reinterpret_cast<uint2*>(q8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(q8 + UT * FRAG) = make_uint2(l0, l1);
      }
      if (wflag && ok) flags[s * N + ag0 + col_a[t]] = 1;
    }
  }
  for (int e = r; e < (KSP - KS) * (WSTEP / 16); e += 256) {
    uint4 z;
    z.x = z.y = z.z = z.w = 0u;
    *reinterpret_cast<uint4*>(wf_wg + (long)KS * WSTEP + (long)e * 16) = z;
  }
}

#ifdef RCMARL_EMU
#define RC_FM_OCC
#else
#define RC_FM_OCC __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(RC_FM_WAVES, RC_FM_WAVES)))
#endif

__global__ RC_FM_OCC void k_forward_mid(const Args2 A) {
  RCMARL_DYN_SMEM(unsigned char, lds);
  unsigned char* Z = lds + LDS_A;
  uint4* sWf = reinterpret_cast<uint4*>(lds + LDS_WF2);
  float* sV = reinterpret_cast<float*>(lds + LDS_SV2);
  float* sB1 = reinterpret_cast<float*>(lds + LDS_B12);
  const int per_seed = A.NG * A.ntiles;
  int s, w;
  if ((A.S & 7) == 0) {                                  // all workgroups of a seed on one XCD (workgroup b -> XCD b % 8)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    s = xcd + 8 * (q / per_seed);
    w = q % per_seed;
  } else {
    s = blockIdx.x / per_seed;
    w = blockIdx.x - s * per_seed;
  }
  const int g = w % A.NG, tile = w / A.NG;               // neighbours share the rows' fragments
  const int r = threadIdx.x, lane = r & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ag0 = g * G, N = A.N, B = A.B, KS = A.KS, in_dim = A.in_dim, ldp = A.ldp;
  const int NS = 2 * ((KS + 2 * KC - 1) / (2 * KC)) * (KC / KC2);          // stages of KC2 steps over the padded reduction (even)
  const NetGeom geo = make_geom(in_dim, HID, 1);
  const float* theta_s = A.theta + (long)s * N * ldp;
  // (uniform 64-bit bases + one 32-bit lane offset: the loads take their address as SGPR pair + VGPR offset, no 64-bit vector adds)
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned char* wf_u = A.wf + (long)s * A.wf_seed + (long)g * (NS * KC2) * WSTEP + (long)wave * FRAG;
  const unsigned char* kf_u = A.kf + (long)s * A.kf_seed + (long)(tile * (TILE / 32) + wave * RB) * KS * FRAG;
  rc_f16_saturate();
#ifdef RC_FM_STAGGER                                     // de-phase the workgroups that share a CU (their phases use different pipes)
  if ((blockIdx.x >> RC_FM_STAGGER_BIT) & 1) rc_sleep(RC_FM_STAGGER);
#endif
  // ---- requests first: W' stages 0 and 1 and the first row fragments (registers), the small arrays (LDS)
  uint4 wreg[2][KC2], xb[PD2][RB];
#pragma unroll
  for (int i = 0; i < KC2; ++i) {
    wreg[0][i] = ld_u4(wf_u + (long)(0 * KC2 + i) * (4 * FRAG) + lane16);
    wreg[1][i] = ld_u4(wf_u + (long)(1 * KC2 + i) * (4 * FRAG) + lane16);
  }
#pragma unroll
  for (int d = 0; d < PD2; ++d)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xb[d][rb] = ld_u4(kf_u + ((long)rb * KS + min(d, KS - 1)) * FRAG + lane16);
  for (int e = r; e < G * WF_AGENT; e += 256) {
    const int a = e / WF_AGENT;
    uint4 v;
    v.x = v.y = v.z = v.w = 0u;
    if (ag0 + a < N) v = A.w2f[((long)s * N + ag0) * WF_AGENT + e];
    sWf[e] = v;
  }
  for (int e = r; e < G * SV; e += 256) {
    const int a = e / SV, i = e - a * SV;
    sV[e] = (i < 2 * HID + 1 && ag0 + a < N) ? theta_s[(long)(ag0 + a) * ldp + geo.o_b2 + i] : 0.f;
  }
  if (r < 64) {
    const int a = r / HID, i = r - a * HID;
    sB1[r] = (r < G * HID && ag0 + a < N) ? theta_s[(long)(ag0 + a) * ldp + geo.o_b1 + i] : 0.f;
  }
  bool live[G];
#pragma unroll
  for (int a = 0; a < G; ++a) live[a] = ag0 + a < N;

  // ================= F (the fused fit's forward phase with 16-KiB stages) =====================================================
  rc_f32x16 acc[UT][RB];
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][rb][q] = 0.f;
  {
    unsigned char* zst = Z + (wave * 64 + lane) * 16;
    auto stage = [&](auto buf_tag, int st) {
      constexpr int BUF = decltype(buf_tag)::value;
      const unsigned char* stg = Z + BUF * STAGE2 + lane * 16;
      const int st2 = (st + 3) % NS;
      uint4 afA[2][UT], afB[2][UT];
      auto ldsA = [&](int ksl, uint4 (&af)[2][UT]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int t = 0; t < UT; ++t) af[p][t] = ld_u4(stg + ((ksl * 2 + p) * UT + t) * FRAG);
      };
      auto kstep = [&](int ksl, uint4 (&cur)[2][UT], uint4 (&nxt)[2][UT]) {
        const int ks = st * KC2 + ksl;
        if (ksl + 1 < KC2) ldsA(ksl + 1, nxt);
#pragma unroll
        for (int p = 1; p >= 0; --p)
#pragma unroll
          for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[t][rb] = rc_mfma_f16(cur[p][t], xb[ksl % PD2][rb], acc[t][rb]);
        RC_SCHED_FENCE();
        const int kn = min(ks + PD2, KS - 1);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xb[ksl % PD2][rb] = ld_u4(kf_u + ((long)rb * KS + kn) * FRAG + lane16);
        *reinterpret_cast<uint4*>(zst + (1 - BUF) * STAGE2 + ksl * (4 * FRAG)) = wreg[1 - BUF][ksl];
        wreg[1 - BUF][ksl] = ld_u4(wf_u + (long)(st2 * KC2 + ksl) * (4 * FRAG) + lane16);
        RC_SCHED_FENCE();
      };
      static_assert(KC2 % PD2 == 0 && KC2 % 2 == 0, "prefetch slots rotate with the stage");
      ldsA(0, afA);
#pragma unroll
      for (int ksl = 0; ksl < KC2; ksl += 2) {
        kstep(ksl, afA, afB);
        kstep(ksl + 1, afB, afA);
      }
      __syncthreads();
    };
#pragma unroll
    for (int i = 0; i < KC2; ++i) *reinterpret_cast<uint4*>(zst + i * (4 * FRAG)) = wreg[0][i];
#pragma unroll
    for (int i = 0; i < KC2; ++i) wreg[0][i] = ld_u4(wf_u + (long)((2 % NS) * KC2 + i) * (4 * FRAG) + lane16);
    __syncthreads();
#if !defined(FM_KNOCK) || FM_KNOCK != 1                  // (measurement builds only: 1 = no reduction loop, 2 = no mid step)
    for (int st = 0; st < NS; st += 2) {
      stage(std::integral_constant<int, 0>{}, st);
      stage(std::integral_constant<int, 1>{}, st + 1);
    }
#endif
  }

  // ================= M: layers 2-3 of the three agents on this wavefront's two row blocks; dz1 packed for the backward GEMM ====
  unsigned short* planes = reinterpret_cast<unsigned short*>(Z + wave * PANEL_B);
  float* recs = reinterpret_cast<float*>(Z + NW * PANEL_B);              // [wavefront][agent][REC]
  for (int e = lane; e < 2 * 32 * (PC - 20); e += 64) {                  // column 20 of the A planes = 1 (-> gb2), other spares 0
    const int pc = e / (32 * (PC - 20)), rw = (e / (PC - 20)) & 31, cl = 20 + e % (PC - 20);
    planes[pc * PLANE + rw * PC + cl] = (pc == 0 && cl == 20) ? (unsigned short)0x3C00 : (unsigned short)0;
  }
  const int brow = tile * TILE + wave * (RB * 32);
  float amax[G];
  auto agent = [&](auto ag_tag) {
    constexpr int AG = decltype(ag_tag)::value;
    amax[AG] = 0.f;
    float* rec = RECS_PER_WAVE ? A.partials + (((long)s * N + (live[AG] ? ag0 + AG : 0)) * (A.ntiles * NW) + tile * NW + wave) * FitRec::SIZE
                               : recs + (wave * G + AG) * REC;
    float gb1l[LU], gw3l[LU], gb3a = 0.f, lossa = 0.f;
#pragma unroll
    for (int u = 0; u < LU; ++u) gb1l[u] = gw3l[u] = 0.f;
    rc_f32x16 g1;
#pragma unroll
    for (int q = 0; q < 16; ++q) g1[q] = 0.f;
    const uint4* wfA = sWf + AG * WF_AGENT + half * 32 + l31;
    const long R0 = (long)(ag0 + AG) * HID;               // the agent's first row of the packed dz image
    unsigned char* dz_s = A.dzp + (long)s * A.dzp_rt * A.dzp_kt * (2 * RC_PK_BLOCK);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int b = brow + rb * 32 + l31;
      float z1[1][LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) z1[0][u] = acc[(LU * AG + u) >> 4][rb][(LU * AG + u) & 15];
      const bool v1[1] = {b < B};
      const float yc[1] = {(live[AG] && b < B) ? A.y[((long)s * N + ag0 + AG) * A.ldb + b] : 0.f};
      unsigned char* const zr[1] = {nullptr};
      float dz1l[1][LU];
      mid_units<1, AG, false>(z1, v1, yc, B, wfA, sV + AG * SV, sB1 + AG * HID, planes, zr, lane, g1, gb1l, gw3l, gb3a, lossa,
                              amax[AG], dz1l);
      // packed f16 pieces of 2^8 dz1 (rcmarl_lattice.h): row = agent*HID + unit, k = replay row; the block's 32 rows x 40
      // (unit, piece) values are transposed through the (now idle) B planes so that the stores are 16-byte chunks (k_mid_fit_v8)
      unsigned short* stg = planes + 2 * PLANE;
      RC_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < LU / 2; ++q) {
        unsigned ph, pl;
        rc_split2h_pair(dz1l[0][2 * q], dz1l[0][2 * q + 1], ph, pl);
        const int u0 = v8_unit(half, 2 * q), u1 = v8_unit(half, 2 * q + 1);
        stg[(u0 * 2 + 0) * 32 + l31] = (unsigned short)ph;
        stg[(u0 * 2 + 1) * 32 + l31] = (unsigned short)pl;
        stg[(u1 * 2 + 0) * 32 + l31] = (unsigned short)(ph >> 16);
        stg[(u1 * 2 + 1) * 32 + l31] = (unsigned short)(pl >> 16);
      }
      RC_WAVE_SYNC();
      const int kt = (brow + 32 * rb) >> 5;
#pragma unroll
      for (int it = 0; it < (HID * 2 * 4 + 63) / 64; ++it) {
        const int c = it * 64 + lane;
        if (c < HID * 2 * 4 && kt < A.dzp_kt && live[AG]) {
          const int up = c >> 2, c4 = c & 3;
          const int unit = up >> 1, piece = up & 1;
          const int R = (int)R0 + unit;
          const uint4 v4 = *reinterpret_cast<const uint4*>(stg + up * 32 + 8 * c4);
          const unsigned off = (unsigned)(((R >> 7) * A.dzp_kt + kt) * 2 + piece) * RC_PK_BLOCK + (unsigned)(R & 127) * 64 +
                               (unsigned)((c4 ^ ((R >> 2) & 3)) << 4);
          *reinterpret_cast<uint4*>(dz_s + off) = v4;
        }
      }
    }
    // the wavefront's record of this agent: row sums by fused DPP adds, the row reduction from the accumulator tile
    float sm[2 * LU + 4];
#pragma unroll
    for (int u = 0; u < LU; ++u) { sm[u] = gb1l[u] * RC_F16_DZ_UNSCALE; sm[LU + u] = gw3l[u]; }
    sm[2 * LU] = gb3a; sm[2 * LU + 1] = lossa; sm[2 * LU + 2] = sm[2 * LU + 3] = 0.f;
#pragma unroll
    for (int q = 0; q < (2 * LU + 4) / 3; ++q) rc_half_sum3_lane31(sm[3 * q], sm[3 * q + 1], sm[3 * q + 2]);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ii = acc_row(q, half);
      if (ii <= HID && l31 < HID && (!RECS_PER_WAVE || live[AG])) rec[ii * HID + l31] = g1[q] * US2;     // rows 0..19 = gW2, row 20 = gb2
    }
    if (l31 == 31 && (!RECS_PER_WAVE || live[AG])) {
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        rec[FitRec::gb1 + v8_unit(half, u)] = sm[u];
        rec[FitRec::gW3 + v8_unit(half, u)] = sm[LU + u];
      }
      if (half == 0) { rec[FitRec::gb3] = sm[2 * LU]; rec[FitRec::loss] = sm[2 * LU + 1]; }
    }
  };
#if defined(FM_KNOCK) && FM_KNOCK == 2
  if (B == 12345) {
#endif
  agent(std::integral_constant<int, 0>{});
  agent(std::integral_constant<int, 1>{});
  agent(std::integral_constant<int, 2>{});
#if defined(FM_KNOCK) && FM_KNOCK == 2
  } else { recs[r] = acc[0][0][0] + acc[1][1][3] + acc[0][1][7] + acc[1][0][15]; amax[0] = amax[1] = amax[2] = 0.f; }
#endif
  if (!RECS_PER_WAVE) {
    __syncthreads();
    for (int e = r; e < G * FitRec::SIZE; e += 256) {
      const int a = e / FitRec::SIZE, idx = e - a * FitRec::SIZE;
      if (ag0 + a < N) {
        const float v = (recs[(0 * G + a) * REC + idx] + recs[(1 * G + a) * REC + idx]) +
                        (recs[(2 * G + a) * REC + idx] + recs[(3 * G + a) * REC + idx]);
        A.partials[(((long)s * N + ag0 + a) * A.ntiles + tile) * FitRec::SIZE + idx] = v;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < G; ++a)
    if (live[a] && amax[a] > RANGE) A.flags[s * N + ag0 + a] = 1;
}
}  // namespace fm

template <int FTW, int NU>
int launch(const Args& a, void* stream) {
  static const bool ok = rc_want_lds(k_fit_fused<FTW, NU>, (size_t)LDS_BYTES);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_fit_fused<FTW, NU>), dim3((unsigned)(a.S * a.NG)), dim3(256), (size_t)LDS_BYTES, stream, a);
  return rcmarl_check_launch();
}

int g_nu = -1;
int nu_mode() {
  if (g_nu < 0) {
    const char* e = getenv("RCMARL_FUSED_NU");
    g_nu = (e && atoi(e) == 1) ? 1 : 2;
  }
  return g_nu;
}

}  // namespace ff
}  // namespace

// geometry of the fused fit's operand images for (N agents, in_dim features, rows_alloc replay rows, a multiple of 256):
// bytes per seed of Kf, KTf and Wf.  Returns RCMARL_ERR_UNSUPPORTED when no kernel is compiled for the shape.
RCMARL_EXPORT int rcmarl_fit_fused_geometry(int N, int in_dim, int hid, int rows_alloc, long* kf_bytes, long* ktf_bytes, long* wf_bytes) {
  if (N <= 0 || in_dim <= 0 || rows_alloc <= 0 || (rows_alloc % ff::TILE)) return RCMARL_ERR_ARG;
  if (hid != ff::HID || in_dim > ff::MAX_K) return RCMARL_ERR_UNSUPPORTED;
  const long ftiles = rc_ceil_div(in_dim, 32), ks = 2 * ftiles, ng = rc_ceil_div(N, ff::G);
  if (kf_bytes) *kf_bytes = (long)(rows_alloc / 32) * ks * ff::FRAG;
  if (ktf_bytes) *ktf_bytes = ftiles * (rows_alloc / 16) * ff::FRAG;
  if (wf_bytes) *wf_bytes = ng * ((ks + 2 * ff::KC - 1) / (2 * ff::KC) * (2 * ff::KC)) * ff::WSTEP;
  return RCMARL_OK;
}

RCMARL_EXPORT int rcmarl_fit_encode(const float* x, long x_seed_stride, const float* alpha, int S, int B, int in_dim, int rows_alloc,
                                    void* kf, void* ktf, void* stream) {
  if (!x || !alpha || !kf || !ktf || S <= 0 || B <= 0 || in_dim <= 0 || rows_alloc < B || (rows_alloc % ff::TILE)) return RCMARL_ERR_ARG;
  if (in_dim > ff::MAX_K) return RCMARL_ERR_UNSUPPORTED;
  const int ftiles = rc_ceil_div(in_dim, 32), ks = 2 * ftiles, rs = rows_alloc / 16;
  const long kf_seed = (long)(rows_alloc / 32) * ks * ff::FRAG, ktf_seed = (long)ftiles * rs * ff::FRAG;
  const int b_pad = rc_ceil_div(B, ff::TILE) * ff::TILE;
  RCMARL_LAUNCH(ff::k_fit_encode, dim3(rc_ceil_div(ftiles * 32, 128), b_pad / 32, S), dim3(256), 0, stream, x, x_seed_stride, alpha, B,
                in_dim, (unsigned char*)kf, kf_seed, ks, (unsigned char*)ktf, ktf_seed, rs, ftiles);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_fit_fused(const void* kf, const void* ktf, void* wf, const float* alpha, float* theta, const float* y,
                                   const int* mask, float* loss_out, int* flags, int S, int N, int B, int in_dim, int hid, int ldp,
                                   int ldb, int rows_alloc, int nsteps, float lr, void* stream) {
  if (!kf || !ktf || !wf || !alpha || !theta || !y || !flags || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || nsteps <= 0 ||
      (ldp & 63) || ldp < in_dim * hid + hid || ldb < B || rows_alloc < B || (rows_alloc % ff::TILE))
    return RCMARL_ERR_ARG;
  if (hid != ff::HID || in_dim > ff::MAX_K) return RCMARL_ERR_UNSUPPORTED;
  ff::Args a;
  const int ftiles = rc_ceil_div(in_dim, 32);
  a.KS = 2 * ftiles; a.RS = rows_alloc / 16; a.FTILES = ftiles; a.NG = rc_ceil_div(N, ff::G);
  a.kf = (const unsigned char*)kf; a.kf_seed = (long)(rows_alloc / 32) * a.KS * ff::FRAG;
  a.ktf = (const unsigned char*)ktf; a.ktf_seed = (long)ftiles * a.RS * ff::FRAG;
  a.wf = (unsigned char*)wf; a.wf_seed = (long)a.NG * ((a.KS + 2 * ff::KC - 1) / (2 * ff::KC) * (2 * ff::KC)) * ff::WSTEP;
  a.alpha = alpha; a.theta = theta; a.y = y; a.mask = mask; a.loss_out = loss_out; a.flags = flags;
  a.S = S; a.N = N; a.B = B; a.in_dim = in_dim; a.ldp = ldp; a.ldb = ldb; a.nsteps = nsteps; a.lr = lr;
  const int ftw = rc_ceil_div(ftiles, ff::NW);
  const bool two = ff::nu_mode() == 2;
#define RC_FF(FTW) (two ? ff::launch<FTW, 2>(a, stream) : ff::launch<FTW, 1>(a, stream))
#ifdef FF_ONLY
  (void)ftw; (void)two;
  return ff::launch<FF_ONLY, FF_NU>(a, stream);
#else
  switch (ftw) {
    case 1: return RC_FF(1);
    case 2: return RC_FF(2);
    case 3: return RC_FF(3);
    case 4: return RC_FF(4);
    case 5: case 6: return RC_FF(6);
    default: return RCMARL_ERR_UNSUPPORTED;
  }
#endif
#undef RC_FF
}

// ---- forward + mid in one launch (see namespace fm above) -----------------------------------------------------------------------
RCMARL_EXPORT int rcmarl_fit_w2_frags(const float* theta, void* w2f, int* flags, int S, int N, int in_dim, int hid, int ldp,
                                      void* stream) {
  if (!theta || !w2f || !flags || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63) || ldp < in_dim * hid + hid) return RCMARL_ERR_ARG;
  if (hid != ff::HID) return RCMARL_ERR_UNSUPPORTED;
  RCMARL_LAUNCH(ff::fm::k_w2_frags, dim3(N, S), dim3(256), 0, stream, theta, (uint4*)w2f, flags, N, in_dim, ldp);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_fit_wf_split(const float* theta, const float* alpha, void* wf, int* flags, int S, int N, int in_dim, int hid,
                                      int ldp, void* stream) {
  if (!theta || !alpha || !wf || !flags || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63) || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid != ff::HID || in_dim > ff::MAX_K) return RCMARL_ERR_UNSUPPORTED;
  const int ftiles = rc_ceil_div(in_dim, 32), ks = 2 * ftiles, ksp = (ks + 2 * ff::KC - 1) / (2 * ff::KC) * (2 * ff::KC);
  const int ng = rc_ceil_div(N, ff::G);
  RCMARL_LAUNCH(ff::fm::k_wf_split, dim3(ng, S), dim3(256), 0, stream, theta, alpha, (unsigned char*)wf, (long)ng * ksp * ff::WSTEP,
                flags, N, in_dim, ldp, ks, ksp, ftiles);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_forward_mid(const void* kf, const void* wf, const void* w2f, const float* theta, const float* y,
                                     float* partials, void* dzp, int dzp_rt, int dzp_kt, int* flags, int S, int N, int B,
                                     int in_dim, int hid, int ldp, int ldb, int rows_alloc, void* stream) {
  if (!kf || !wf || !w2f || !theta || !y || !partials || !dzp || !flags || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 ||
      (ldp & 63) || ldp < in_dim * hid + hid || ldb < B || rows_alloc < B || (rows_alloc % ff::TILE))
    return RCMARL_ERR_ARG;
  if (hid != ff::HID || in_dim > ff::MAX_K) return RCMARL_ERR_UNSUPPORTED;
  ff::fm::Args2 a;
  const int ftiles = rc_ceil_div(in_dim, 32);
  a.KS = 2 * ftiles; a.NG = rc_ceil_div(N, ff::G); a.ntiles = rc_ceil_div(B, ff::TILE);
  if (dzp_rt * 128 < N * hid || dzp_kt * 32 < a.ntiles * ff::TILE) return RCMARL_ERR_ARG;
  a.kf = (const unsigned char*)kf; a.kf_seed = (long)(rows_alloc / 32) * a.KS * ff::FRAG;
  a.wf = (const unsigned char*)wf; a.wf_seed = (long)a.NG * ((a.KS + 2 * ff::KC - 1) / (2 * ff::KC) * (2 * ff::KC)) * ff::WSTEP;
  a.w2f = (const uint4*)w2f; a.theta = theta; a.y = y; a.partials = partials;
  a.dzp = (unsigned char*)dzp; a.dzp_rt = dzp_rt; a.dzp_kt = dzp_kt; a.flags = flags;
  a.S = S; a.N = N; a.B = B; a.in_dim = in_dim; a.ldp = ldp; a.ldb = ldb;
  rc_form_set(dzp, 1);
  static const bool ok = rc_want_lds(ff::fm::k_forward_mid, (size_t)ff::fm::LDS_BYTES2);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH(ff::fm::k_forward_mid, dim3((unsigned)(S * a.NG * a.ntiles)), dim3(256), (size_t)ff::fm::LDS_BYTES2, stream, a);
  return rcmarl_check_launch();
}
