// Wide networks (any hidden width; BASELINE configs[4]: the 512-unit critic): the per-agent MLP layers become true
// dense GEMMs on the matrix core (v_mfma_f32_32x32x2_f32), the 1-unit head and its consensus become column / row
// passes over the feature-major activations.  The 20-unit networks of the reference (main.py:59-82) stay on the
// fused kernels of mid_kernels.hip; this file serves the SAME reference functions for hid != 20:
//
//   rcmarl_dense_forward        Dense + LeakyReLU of one layer, all agents of all seeds  (model(x), :95-97,:114)
//   rcmarl_dense_backward_data  dz_in = (W dz_out) * lrelu'(a_in)                        (fit(), :118,:136)
//   rcmarl_dense_backward_sgd   W -= lr * in^T dz                                        (fit(), plain SGD)
//   rcmarl_wide_head_value      V = a2 . W3 + b3  [, r + gamma V]                        (:114-115, :95-97)
//   rcmarl_wide_head_fit        MSE head: dz3, dz2 (in place of a2), gW3, gb3, gb2, loss (fit(), :118)
//   rcmarl_wide_bias_grad       gb1 = row sums of dz1
//   rcmarl_wide_small_sgd       b1, b2, W3, b3 -= lr * grad; loss of step 0
//   rcmarl_wide_consensus_head  estimate consensus + projection residual (K2+K3)         (:168-206, :60-84)
//   rcmarl_wide_head_apply      W3 += gW3/B, b3 += gb3/B
//
// (file:line = agents/resilient_CAC_agents.py of the reference.)  Activations are feature-major
// act[S][N*hid][ldb] like a1t; parameters are the Keras-ordered rows of theta[S][N][ldp] (rcmarl_common.h).
//
// GEMM: C[M x Nc] = A[M x K] B[K x Nc] per (seed, agent); 128x128x16 tiles, 4 wavefronts of 64x64 (2x2 MFMA
// 32x32 accumulators), operands staged through LDS as [k][m] / [k][n] (one float per lane per MFMA operand is the
// v_mfma_f32_32x32x2_f32 contract), register-prefetched double buffer, one barrier per k-tile.  Both operands
// come in either orientation (contiguous along k, or along m/n), which covers forward (W^T x), backward-data
// (W dz) and backward-weights (x^T dz^T) without materialising a transpose.
#include "rcmarl_common.h"
#include "rcmarl_lattice.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "selnet_generated.inc"

namespace {

constexpr int WBM = 128, WBN = 128, WBK = 16, WLD = 132;
enum { WEPI_BIAS_LRELU = 0, WEPI_BIAS = 1, WEPI_LRELU_GRAD = 2, WEPI_SGD = 3 };

struct WArgs {
  const float* A; long A_zs, A_za; int lda;
  const float* B; long B_zs, B_za; int ldb;
  float* C; long C_zs, C_za; int ldc;
  const float* aux; long aux_zs, aux_za; int ldaux;     // bias[m] (BIAS*) or activation(m, n) (LRELU_GRAD)
  const int* mask; float lr;                            // SGD
  int M, N, K, NA;
  float sa, sb;                                         // k_wgemm16: power-of-two scales of A and B before the split into f16 pieces
};

// global -> registers: this thread's 8 floats of a [128 x 16] operand tile (zero-filled outside the matrix).
// KC: the operand is contiguous along k (element (r, k) at P[r*ld + k]); otherwise along r (P[k*ld + r]).
template <bool KC, bool VEC>
__device__ __forceinline__ void w_load(const float* __restrict__ P, int ld, int r0, int k0, int R, int K, float (&reg)[8]) {
  const int t = threadIdx.x;
  if (VEC) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!KC) {
        const int k = idx >> 5, r4 = (idx & 31) * 4;
        if (k0 + k < K && r0 + r4 < R) v = *reinterpret_cast<const float4*>(P + (long)(k0 + k) * ld + r0 + r4);
      } else {
        const int r = idx >> 2, k4 = (idx & 3) * 4;
        if (r0 + r < R && k0 + k4 < K) v = *reinterpret_cast<const float4*>(P + (long)(r0 + r) * ld + k0 + k4);
      }
      reg[4 * i] = v.x; reg[4 * i + 1] = v.y; reg[4 * i + 2] = v.z; reg[4 * i + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = t + 256 * i;
      if (!KC) {
        const int k = idx >> 7, r = idx & 127;
        reg[i] = (k0 + k < K && r0 + r < R) ? P[(long)(k0 + k) * ld + r0 + r] : 0.f;
      } else {
        const int r = idx >> 4, k = idx & 15;
        reg[i] = (r0 + r < R && k0 + k < K) ? P[(long)(r0 + r) * ld + k0 + k] : 0.f;
      }
    }
  }
}

// registers -> LDS tile s[k][r] (row stride WLD)
template <bool KC, bool VEC>
__device__ __forceinline__ void w_store(float* __restrict__ s, const float (&reg)[8]) {
  const int t = threadIdx.x;
  if (VEC) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 256 * i;
      if (!KC) {
        const int k = idx >> 5, r4 = (idx & 31) * 4;
        *reinterpret_cast<float4*>(s + k * WLD + r4) = make_float4(reg[4 * i], reg[4 * i + 1], reg[4 * i + 2], reg[4 * i + 3]);
      } else {
        const int r = idx >> 2, k4 = (idx & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[(k4 + j) * WLD + r] = reg[4 * i + j];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = t + 256 * i;
      if (!KC) s[(idx >> 7) * WLD + (idx & 127)] = reg[i];
      else s[(idx & 15) * WLD + (idx >> 4)] = reg[i];
    }
  }
}

// The fp32 k-loop: C tile (m0, n0) of one (seed, agent) accumulated with v_mfma_f32_32x32x2_f32; sA / sB: two stages of [WBK][WLD] floats each.
template <bool A_KC, bool B_KC, bool VEC>
__device__ __forceinline__ void w_loop_f32(const WArgs& a, const float* __restrict__ A, const float* __restrict__ Bp, int m0, int n0,
                                           float* __restrict__ sA, float* __restrict__ sB, rc_f32x16 (&acc)[2][2]) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ra[8], rb[8];
  const int nk = (a.K + WBK - 1) / WBK;
  w_load<A_KC, VEC>(A, a.lda, m0, 0, a.M, a.K, ra);
  w_load<B_KC, VEC>(Bp, a.ldb, n0, 0, a.N, a.K, rb);
  w_store<A_KC, VEC>(sA, ra);
  w_store<B_KC, VEC>(sB, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      w_load<A_KC, VEC>(A, a.lda, m0, (kt + 1) * WBK, a.M, a.K, ra);
      w_load<B_KC, VEC>(Bp, a.ldb, n0, (kt + 1) * WBK, a.N, a.K, rb);
    }
    const float* __restrict__ pa = sA + cur * (WBK * WLD) + (l >> 5) * WLD + wm * 64 + (l & 31);
    const float* __restrict__ pb = sB + cur * (WBK * WLD) + (l >> 5) * WLD + wn * 64 + (l & 31);
#pragma unroll
    for (int kk = 0; kk < WBK; kk += 2) {
      const float a0 = pa[kk * WLD], a1 = pa[kk * WLD + 32];
      const float b0 = pb[kk * WLD], b1 = pb[kk * WLD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      w_store<A_KC, VEC>(sA + (cur ^ 1) * (WBK * WLD), ra);
      w_store<B_KC, VEC>(sB + (cur ^ 1) * (WBK * WLD), rb);
    }
    __syncthreads();
  }
}

// epilogue: register r of tile (i, j) is C[m0 + wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*(l>>5)][n0 + wn*64 + j*32 + (l&31)] -- the layout of
// v_mfma_f32_32x32x2_f32 and of v_mfma_f32_32x32x16_f16 alike.  Full tiles run straight-line (no per-element predicate, so the bias /
// activation / old-weight loads are all issued before the first dependent use); edge tiles keep the predicates.  `unscale`: the
// accumulators hold unscale^-1 times the product (1 for the fp32 loop).
template <int EPI>
__device__ __forceinline__ void w_epilogue(const WArgs& a, int s, int ag, int m0, int n0, const rc_f32x16 (&acc)[2][2], float unscale) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  float* __restrict__ C = a.C + s * a.C_zs + ag * a.C_za;
  const float* __restrict__ aux = a.aux ? a.aux + s * a.aux_zs + ag * a.aux_za : nullptr;
  const int mb = m0 + wm * 64 + 4 * (l >> 5), nb = n0 + wn * 64 + (l & 31);
  auto tile_out = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    float bias[2][16];
    if (EPI == WEPI_BIAS_LRELU || EPI == WEPI_BIAS) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
          bias[i][r] = (FULL || m < a.M) ? aux[m] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = nb + j * 32;
        if (!FULL && n >= a.N) continue;
        float* __restrict__ cp = C + (long)(mb + i * 32) * a.ldc + n;
        const float* __restrict__ xp = (EPI == WEPI_LRELU_GRAD) ? aux + (long)(mb + i * 32) * a.ldaux + n : nullptr;
        float old[16];
        if (EPI == WEPI_LRELU_GRAD || EPI == WEPI_SGD) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const bool ok = FULL || mb + i * 32 + dr < a.M;
            old[r] = !ok ? 0.f : (EPI == WEPI_SGD ? cp[(long)dr * a.ldc] : xp[(long)dr * a.ldaux]);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (!FULL && mb + i * 32 + dr >= a.M) continue;
          const float v = acc[i][j][r] * unscale;              // (a power of two: exact; 1.0 after the fp32 loop)
          float o;
          if (EPI == WEPI_BIAS_LRELU) o = rc_lrelu(v + bias[i][r]);
          else if (EPI == WEPI_BIAS) o = v + bias[i][r];
          else if (EPI == WEPI_LRELU_GRAD) o = v * rc_lrelu_grad_from_act(old[r]);
          else o = old[r] - a.lr * v;
          cp[(long)dr * a.ldc] = o;
        }
      }
  };
  if (m0 + WBM <= a.M && n0 + WBN <= a.N) tile_out(std::true_type{}); else tile_out(std::false_type{});
}

template <bool A_KC, bool B_KC, int EPI, bool VEC>
__global__ __launch_bounds__(256, 3) void k_wgemm(const WArgs a) {      // 3 workgroups per CU: <= 170 VGPRs
  __shared__ __attribute__((aligned(16))) float sA[2 * WBK * WLD];
  __shared__ __attribute__((aligned(16))) float sB[2 * WBK * WLD];
  const int z = blockIdx.z, s = z / a.NA, ag = z - s * a.NA;
  if (EPI == WEPI_SGD && a.mask && !a.mask[ag]) return;        // workgroup-uniform
  const int m0 = blockIdx.y * WBM, n0 = blockIdx.x * WBN;
  rc_f32x16 acc[2][2];
  w_loop_f32<A_KC, B_KC, VEC>(a, a.A + s * a.A_zs + ag * a.A_za, a.B + s * a.B_zs + ag * a.B_za, m0, n0, sA, sB, acc);
  w_epilogue<EPI>(a, s, ag, m0, n0, acc, 1.f);
}

// ---------------------------------------------------------------------------------------------
// The same GEMM on the 16-bit matrix core: both fp32 operands as TWO f16 pieces of the value times a power of two (h = rn(v),
// l = rn(v - h): v to one unit in its last place, rcmarl_lattice.h), three products per fp32 product -- l*h + h*l + h*h, the l*l term
// (2^-22 of the product) dropped -- on v_mfma_f32_32x32x16_f16, fp32 accumulate.  16x the rate of the fp32-input MFMA at 3x the
// instructions.  128 x 128 x 32 tiles, 4 wavefronts of 64 x 64; the loader splits while it stages (global fp32 -> registers -> pieces
// -> LDS), two 32-KiB stages, one barrier per k-tile, two workgroups per CU.
//   operand contiguous along k   -> LDS [row][32 k] f16, 64-byte rows, 16-byte chunks XORed with (row >> 2) & 3; an MFMA fragment
//                                   (row, 8 k) is one ds_read_b128
//   operand contiguous along m/n -> LDS [16-row window][32 k][16 rows] f16, 32-byte rows; a fragment is two TRANSPOSE reads
//                                   (ds_read_b64_tr_b16: lane = row, 4 k each).  k-row placement inside a window: k ^ (w & 3) ^
//                                   4 (w & 1) -- the four windows a 16-lane group of the 8-byte stores covers land on different banks,
//                                   and the two windows a transpose read serves together on different halves of the bank row
//                                   (the conflict rules measured on k_mid_fit_v8, profiles/r04u_lds_conflict_knockouts.txt).
// A workgroup whose operands leave the f16 range (|scaled value| > 65000 anywhere in its row / column panels) recomputes its tile
// with the fp32 loop above before the epilogue: same kernel, no flags, no second launch.
constexpr int W16_BK = 32, W16_PIECE = 128 * W16_BK * 2, W16_STAGE = 4 * W16_PIECE;      // bytes: one piece of one operand, one stage
#define RC_W16_RANGE 65000.f
// Precision window of a two-piece operand (h = rn_f16(x), l = rn_f16(x - h), x = scale * value): x to 2^-22 relative while the
// low piece is a normal f16, i.e. |x| >= 2^-3; below that l is subnormal and x carries an ABSOLUTE error of up to 2^-25 (scaled
// units).  With the fixed scales: weights (2^10) full precision from |w| >= 1.2e-4, absolute 2.9e-11 below; dz (2^8) from
// |dz| >= 4.9e-4, absolute 1.2e-10 below; activations (2^6 since round 5; 1 before: full precision only from 0.125) from
// |a| >= 2.0e-3, absolute 4.7e-10 below -- against fp32's own 6e-8 relative on values of order one.  Above 65000 / scale
// (weights 63, dz 254, activations 1015) the workgroup recomputes its tile in fp32.
#define RC_W16_ACT_SCALE 64.f

template <bool KC>
__device__ __forceinline__ void w16_load(const float* __restrict__ P, int ld, int r0, int k0, int R, int K, float4 (&reg)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      if (r0 + r < R && k0 + k4 < K) v = *reinterpret_cast<const float4*>(P + (long)(r0 + r) * ld + k0 + k4);
    } else {
      const int k = idx >> 5, r4 = (idx & 31) * 4;
      if (k0 + k < K && r0 + r4 < R) v = *reinterpret_cast<const float4*>(P + (long)(k0 + k) * ld + r0 + r4);
    }
    reg[i] = v;
  }
}

__device__ __forceinline__ int w16_krow(int k, int w) { return k ^ (w & 3) ^ (4 * (w & 1)); }

// registers -> the two piece planes of one operand (ph, pl: W16_PIECE bytes each); returns max |scaled value|
template <bool KC>
__device__ __forceinline__ float w16_store(unsigned char* __restrict__ ph, unsigned char* __restrict__ pl, const float4 (&reg)[4],
                                           float scale, float amax) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const float x0 = reg[i].x * scale, x1 = reg[i].y * scale, x2 = reg[i].z * scale, x3 = reg[i].w * scale;
    amax = rc_amax3(rc_amax3(amax, x0, x1), x2, x3);
    uint2 h, lo;
    rc_split2h_pair(x0, x1, h.x, lo.x);
    rc_split2h_pair(x2, x3, h.y, lo.y);
    int off;
    if (KC) {
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      off = r * 64 + (((k4 >> 3) ^ ((r >> 2) & 3)) << 4) + (k4 & 4) * 2;
    } else {
      const int k = idx >> 5, r4 = (idx & 31) * 4, w = r4 >> 4;
      off = w * 1024 + w16_krow(k, w) * 32 + (r4 & 15) * 2;
    }
    *reinterpret_cast<uint2*>(ph + off) = h;
    *reinterpret_cast<uint2*>(pl + off) = lo;
  }
  return amax;
}

// the MFMA fragment (row tr0 + (lane & 31), k = 16 ks + 8 (lane >> 5) .. + 7) of one piece plane
template <bool KC>
__device__ __forceinline__ uint4 w16_frag(const unsigned char* __restrict__ pp, int tr0, int ks) {
  const int l = threadIdx.x & 63, kg = l >> 5;
  if (KC) {
    const int r = tr0 + (l & 31);
    return *reinterpret_cast<const uint4*>(pp + r * 64 + (((2 * ks + kg) ^ ((r >> 2) & 3)) << 4));
  } else {
    const int w = (tr0 >> 4) + ((l >> 4) & 1), j = l & 15, kb = 16 * ks + 8 * kg + (j >> 2);
    const unsigned char* base = pp + w * 1024 + 8 * (j & 3);
    const uint2 t0 = rc_lds_read_tr16(reinterpret_cast<const unsigned short*>(base + w16_krow(kb, w) * 32));
    const uint2 t1 = rc_lds_read_tr16(reinterpret_cast<const unsigned short*>(base + w16_krow(kb + 4, w) * 32));
    uint4 f;
    f.x = t0.x; f.y = t0.y; f.z = t1.x; f.w = t1.y;
    return f;
  }
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256, 2) void k_wgemm16(const WArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * W16_STAGE];
  __shared__ int s_ovf;
  static_assert(2 * W16_STAGE >= (int)(4 * WBK * WLD * sizeof(float)), "the fp32 loop's stages fit the same memory");
  // workgroup id -> ((seed, agent) z, m-tile, n-tile): all tiles of a (seed, agent) on ONE XCD (workgroups go round the eight XCDs by
  // id), m-tile fastest -- the workgroups that share a B panel run side by side and the agent's A matrix stays in that XCD's L2
  const int tm = (a.M + WBM - 1) / WBM, tn = (a.N + WBN - 1) / WBN, per = tm * tn, nz = (int)(gridDim.x / per);
  int z, wq;
  if ((nz & 7) == 0) {
    const int g = blockIdx.x, q = g >> 3;
    z = (g & 7) + 8 * (q / per);
    wq = q % per;
  } else {
    z = blockIdx.x / per;
    wq = blockIdx.x - z * per;
  }
  const int s = z / a.NA, ag = z - s * a.NA;
  if (EPI == WEPI_SGD && a.mask && !a.mask[ag]) return;        // workgroup-uniform
  const int m0 = (wq % tm) * WBM, n0 = (wq / tm) * WBN;
  const float* __restrict__ A = a.A + s * a.A_zs + ag * a.A_za;
  const float* __restrict__ Bp = a.B + s * a.B_zs + ag * a.B_za;
  const int t = threadIdx.x, w = t >> 6, wm = w >> 1, wn = w & 1;
  rc_f16_saturate();
  if (t == 0) s_ovf = 0;
  rc_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4], rb[4];
  float amax = 0.f;
  const int nk = (a.K + W16_BK - 1) / W16_BK;
  w16_load<A_KC>(A, a.lda, m0, 0, a.M, a.K, ra);
  w16_load<B_KC>(Bp, a.ldb, n0, 0, a.N, a.K, rb);
  amax = w16_store<A_KC>(sm, sm + W16_PIECE, ra, a.sa, amax);
  amax = w16_store<B_KC>(sm + 2 * W16_PIECE, sm + 3 * W16_PIECE, rb, a.sb, amax);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* __restrict__ st = sm + (kt & 1) * W16_STAGE;
    unsigned char* __restrict__ nx = sm + ((kt & 1) ^ 1) * W16_STAGE;
    if (kt + 1 < nk) {
      w16_load<A_KC>(A, a.lda, m0, (kt + 1) * W16_BK, a.M, a.K, ra);
      w16_load<B_KC>(Bp, a.ldb, n0, (kt + 1) * W16_BK, a.N, a.K, rb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = w16_frag<A_KC>(st, wm * 64 + 32 * i, ks);
        al[i] = w16_frag<A_KC>(st + W16_PIECE, wm * 64 + 32 * i, ks);
        bh[i] = w16_frag<B_KC>(st + 2 * W16_PIECE, wn * 64 + 32 * i, ks);
        bl[i] = w16_frag<B_KC>(st + 3 * W16_PIECE, wn * 64 + 32 * i, ks);
      }
      // smallest products first; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = rc_mfma_f16(al[i], bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = rc_mfma_f16(ah[i], bl[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = rc_mfma_f16(ah[i], bh[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      amax = w16_store<A_KC>(nx, nx + W16_PIECE, ra, a.sa, amax);
      amax = w16_store<B_KC>(nx + 2 * W16_PIECE, nx + 3 * W16_PIECE, rb, a.sb, amax);
    }
    __syncthreads();
  }
  if (amax > RC_W16_RANGE) s_ovf = 1;                         // (NaN operands do not take this branch: they poison either loop alike)
  __syncthreads();
  float unscale = 1.f / (a.sa * a.sb);
  if (s_ovf) {                                                // workgroup-uniform: out of the f16 range -> the fp32 loop, same tile
    __syncthreads();
    float* fa = reinterpret_cast<float*>(sm);
    w_loop_f32<A_KC, B_KC, true>(a, A, Bp, m0, n0, fa, fa + 2 * WBK * WLD, acc);
    unscale = 1.f;
  }
  w_epilogue<EPI>(a, s, ag, m0, n0, acc, unscale);
}

static inline bool w_al4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// RCMARL_WIDE_F16 (default 1): the dense layers of wide networks on the 16-bit matrix core (k_wgemm16); 0 = the fp32-input MFMA
// kernel.  Read once; rcmarl_wide_set_f16_mode() changes it (tests, bench.py's exact-form line).
std::mutex g_w16_mu;
int g_w16_mode = -1;
int w16_mode() {
  std::lock_guard<std::mutex> lk(g_w16_mu);
  if (g_w16_mode < 0) {
    const char* e = getenv("RCMARL_WIDE_F16");
    g_w16_mode = (e && e[0] == '0') ? 0 : 1;
  }
  return g_w16_mode;
}

template <bool A_KC, bool B_KC, int EPI>
static int w_launch(const WArgs& a, int S, void* stream) {
  // float4 staging needs every tile row 16-B aligned and whole float4s inside the matrix
  const bool vec = w_al4(a.A) && w_al4(a.B) && !(a.lda & 3) && !(a.ldb & 3) && !(a.A_zs & 3) && !(a.A_za & 3) &&
                   !(a.B_zs & 3) && !(a.B_za & 3) && !(a.K & 3) && (A_KC || !(a.M & 3)) && (B_KC || !(a.N & 3));
  const dim3 grid(rc_ceil_div(a.N, WBN), rc_ceil_div(a.M, WBM), S * a.NA), block(256);
  if (vec && a.sa > 0.f && a.sb > 0.f && w16_mode())
    RCMARL_LAUNCH((k_wgemm16<A_KC, B_KC, EPI>), dim3(grid.x * grid.y * grid.z), block, 0, stream, a);
  else if (vec) RCMARL_LAUNCH((k_wgemm<A_KC, B_KC, EPI, true>), grid, block, 0, stream, a);
  else RCMARL_LAUNCH((k_wgemm<A_KC, B_KC, EPI, false>), grid, block, 0, stream, a);
  return rcmarl_check_launch();
}

// ---------------------------------------------------------------------------------------------
// column passes: one thread per replay row b, loop over the hid features (coalesced over b)
constexpr int WROWS = 256;

// V[b] = a2[:, b] . W3 + b3;  out = V  or  r_applied + gamma * V
__global__ __launch_bounds__(256) void k_whead_value(const float* __restrict__ a2, const float* __restrict__ theta,
                                                     const float* __restrict__ r_applied, float gamma,
                                                     float* __restrict__ out, int N, int B, int in_dim, int hid, int ldp,
                                                     int ldb) {
  const int s = blockIdx.z, i = blockIdx.y, b = blockIdx.x * WROWS + threadIdx.x;
  if (b >= B) return;
  const NetGeom g = make_geom(in_dim, hid, 1);
  const float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const float* __restrict__ col = a2 + ((long)s * N + i) * hid * ldb + b;
  float v = 0.f;
  for (int k = 0; k < hid; ++k) v = fmaf(col[(long)k * ldb], th[g.o_W3 + k], v);
  v += th[g.o_b3];
  const long o = ((long)s * N + i) * ldb + b;
  out[o] = r_applied ? r_applied[o] + gamma * v : v;
}

// MSE head, column half: diff = V - y, dz3 = 2 diff / B, loss partial of this chunk
__global__ __launch_bounds__(256) void k_whead_fit_cols(const float* __restrict__ a2, const float* __restrict__ theta,
                                                        const float* __restrict__ y, float* __restrict__ dz3,
                                                        float* __restrict__ losspart, int N, int B, int in_dim,
                                                        int hid, int ldp, int ldb, int nchunk) {
  __shared__ float red[4];
  const int s = blockIdx.z, i = blockIdx.y, b = blockIdx.x * WROWS + threadIdx.x;
  const bool valid = b < B;
  const NetGeom g = make_geom(in_dim, hid, 1);
  const float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const float* __restrict__ col = a2 + ((long)s * N + i) * hid * ldb + (valid ? b : 0);
  float v = 0.f;
  for (int k = 0; k < hid; ++k) v = fmaf(col[(long)k * ldb], th[g.o_W3 + k], v);
  v += th[g.o_b3];
  const long o = ((long)s * N + i) * ldb + b;
  const float diff = valid ? v - y[o] : 0.f;
  if (valid) dz3[o] = (2.0f * diff) / (float)B;
  const float sq = rc_wave_sum(diff * diff);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) losspart[((long)s * N + i) * nchunk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// row passes: one wavefront per feature row j (contiguous over b), results to grads[s][n][...]
// grads record per (seed, agent): [gW3 (hid) | gb3 | gb2 (hid) | gb1 (hid)]
__host__ __device__ static inline int w_grad_size(int hid) { return 3 * hid + 1; }

enum { WROW_FIT = 0, WROW_SUM = 1, WROW_DOT = 2 };

// FIT: gW3[j] = sum_b a2[j][b] dz3[b];  dz2[j][b] = W3[j] dz3[b] lrelu'(a2[j][b]) (in place);  gb2[j] = sum_b dz2;
//      row j == hid: gb3 = sum_b dz3[b]
// SUM: out[off + j] = sum_b act[j][b]
// DOT: out[j] = sum_b act[j][b] vec[b];  row j == hid: out[hid] = sum_b vec[b]
template <int MODE>
__global__ __launch_bounds__(256) void k_wrows(float* __restrict__ act, const float* __restrict__ vec,
                                               const float* __restrict__ theta, const int* __restrict__ coop,
                                               float* __restrict__ grads, int out_off, int N, int B, int in_dim, int hid,
                                               int ldp, int ldb) {
  const int s = blockIdx.z, i = blockIdx.y, j = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  const int nrows = (MODE == WROW_SUM) ? hid : hid + 1;
  if (j >= nrows) return;
  if (coop && !coop[i]) return;
  float* __restrict__ gr = grads + ((long)s * N + i) * w_grad_size(hid);
  const float* __restrict__ v = vec ? vec + ((long)s * N + i) * ldb : nullptr;
  if (j == hid) {                                   // the bias of the head: plain sum of vec
    float acc = 0.f;
    for (int b = l; b < B; b += 64) acc += v[b];
    acc = rc_wave_sum(acc);
    if (l == 0) gr[hid] = acc;
    return;
  }
  float* __restrict__ row = act + (((long)s * N + i) * hid + j) * ldb;
  if (MODE == WROW_FIT) {
    const float w3 = theta[((long)s * N + i) * ldp + make_geom(in_dim, hid, 1).o_W3 + j];
    float gw = 0.f, gb = 0.f;
    for (int b = l; b < B; b += 64) {
      const float a2 = row[b], dv = v[b];
      gw = fmaf(a2, dv, gw);
      const float dz = dv * w3 * rc_lrelu_grad_from_act(a2);
      row[b] = dz;
      gb += dz;
    }
    gw = rc_wave_sum(gw);
    gb = rc_wave_sum(gb);
    if (l == 0) { gr[j] = gw; gr[hid + 1 + j] = gb; }
  } else if (MODE == WROW_SUM) {
    float acc = 0.f;
    for (int b = l; b < B; b += 64) acc += row[b];
    acc = rc_wave_sum(acc);
    if (l == 0) gr[out_off + j] = acc;
  } else {
    float acc = 0.f;
    for (int b = l; b < B; b += 64) acc = fmaf(row[b], v[b], acc);
    acc = rc_wave_sum(acc);
    if (l == 0) gr[j] = acc;
  }
}

// b1, b2, W3, b3 -= lr * grad (masked agents only); loss_out[s][n] = sum(diff^2)/B
__global__ __launch_bounds__(256) void k_wsmall_sgd(const float* __restrict__ grads, const float* __restrict__ losspart,
                                                    float* __restrict__ theta, const int* __restrict__ mask,
                                                    float* __restrict__ loss_out, int N, int B, int in_dim, int hid,
                                                    int ldp, int nchunk, float lr) {
  const int s = blockIdx.y, i = blockIdx.x;
  if (loss_out && threadIdx.x == 0) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += losspart[((long)s * N + i) * nchunk + c];
    loss_out[(long)s * N + i] = sum / (float)B;
  }
  if (mask && !mask[i]) return;
  const NetGeom g = make_geom(in_dim, hid, 1);
  float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const float* __restrict__ gr = grads + ((long)s * N + i) * w_grad_size(hid);
  for (int e = threadIdx.x; e < 3 * hid + 1; e += 256) {
    int o;
    if (e < hid) o = g.o_W3 + e;
    else if (e == hid) o = g.o_b3;
    else if (e < 2 * hid + 1) o = g.o_b2 + (e - hid - 1);
    else o = g.o_b1 + (e - 2 * hid - 1);
    th[o] = th[o] - lr * gr[e];
  }
}

// ---------------------------------------------------------------------------------------------
// K2+K3 for a wide head.  hmat[s][i][m][0..hid) = W3 of msg[nbr[i][m]] (m < d) / of the live net (m == d);
// hb[s][i][m] = the matching b3.
__global__ __launch_bounds__(256) void k_wgather_heads(const float* __restrict__ theta, const float* __restrict__ msg,
                                                       const int* __restrict__ nbr, const int* __restrict__ coop,
                                                       float* __restrict__ hmat, float* __restrict__ hb, int N,
                                                       int in_dim, int hid, int ldp, int d) {
  const int s = blockIdx.z, i = blockIdx.y, m = blockIdx.x;
  if (!coop[i]) return;
  const NetGeom g = make_geom(in_dim, hid, 1);
  const float* __restrict__ src = (m < d) ? msg + ((long)s * N + nbr[i * d + m]) * ldp : theta + ((long)s * N + i) * ldp;
  float* __restrict__ dst = hmat + (((long)s * N + i) * (d + 1) + m) * hid;
  for (int k = threadIdx.x; k < hid; k += 256) dst[k] = src[g.o_W3 + k];
  if (threadIdx.x == 0) hb[((long)s * N + i) * (d + 1) + m] = src[g.o_b3];
}

// est[s][i][m][b] (m < d: neighbours' estimates, m == d: the live head) -> resilient aggregate over m < d
// (own = m == 0, in_nodes[i][0] == i), residual e[b] = (agg - V_live) / (|phi|^2 + 1).  Order statistics by rank
// counting with index tie-break (any d, H), as k_consensus_head_generic.
constexpr int WSEL_ROWS = 128;            // d * 128 floats of LDS per workgroup (d = 66: 33 KiB)
// |phi|^2 + 1 of replay row b: from the per-tile parts a producer left (nparts > 0: csrc/dense_pk.hip writes sum_j a2[j][b]^2 per tile of
// units beside the activations) or by a pass over the hid activations of the row
__device__ __forceinline__ float w_norm1(const float* __restrict__ phi, const float* __restrict__ nparts, int n_parts, long zi, int b,
                                         int hid, int ldb) {
  float nrm = 0.f;
  if (nparts != nullptr) {
    for (int t = 0; t < n_parts; ++t) nrm += nparts[(zi * n_parts + t) * ldb + b];
  } else {
    const float* __restrict__ col = phi + zi * hid * ldb + b;
    for (int k = 0; k < hid; ++k) { const float p = col[(long)k * ldb]; nrm = fmaf(p, p, nrm); }
  }
  return nrm + 1.0f;
}

__global__ __launch_bounds__(WSEL_ROWS) void k_wselect(const float* __restrict__ est, const float* __restrict__ phi,
                                                 const float* __restrict__ nparts, int n_parts,
                                                 const int* __restrict__ coop, float* __restrict__ ebuf,
                                                 float* __restrict__ agg_out, int N, int B, int hid, int ldb, int d,
                                                 int H) {
  RCMARL_DYN_SMEM(float, sv);                 // [d][WSEL_ROWS]
  constexpr int WROWS = WSEL_ROWS;            // (shadows the column-pass chunk: this kernel walks 128 rows)
  const int s = blockIdx.z, i = blockIdx.y;
  if (!coop[i]) return;
  const int r = threadIdx.x, b = blockIdx.x * WROWS + r;
  if (b >= B) return;                         // no barrier below: every thread works on its own LDS column
  const float* __restrict__ e0 = est + ((long)s * N + i) * (d + 1) * ldb + b;
  for (int k = 0; k < d; ++k) sv[k * WROWS + r] = e0[(long)k * ldb];
  const float nrm = w_norm1(phi, nparts, n_parts, (long)s * N + i, b, hid, ldb);
  const float own = sv[r];
  float lo = own, hi = own;
  for (int k = 0; k < d; ++k) {
    const float x = sv[k * WROWS + r];
    int rank = 0;
    for (int m = 0; m < d; ++m) {
      const float yv = sv[m * WROWS + r];
      rank += (yv < x || (yv == x && m < k)) ? 1 : 0;
    }
    if (rank == H) lo = x;
    if (rank == d - H - 1) hi = x;
  }
  const float lower = fminf(lo, own), upper = fmaxf(hi, own);
  float sum = 0.f;
  for (int k = 0; k < d; ++k) sum += __builtin_amdgcn_fmed3f(sv[k * WROWS + r], lower, upper);
  const float agg = sum / (float)d;
  const float v_live = e0[(long)d * ldb];
  const long o = ((long)s * N + i) * ldb + b;
  ebuf[o] = (agg - v_live) / nrm;
  if (agg_out) agg_out[o] = agg;
}

// the same with the generated selection network of (D, H) on registers (selnet_generated.inc) instead of rank counting
template <int D, int H>
__global__ __launch_bounds__(WSEL_ROWS) void k_wselect_net(const float* __restrict__ est, const float* __restrict__ phi,
                                                           const float* __restrict__ nparts, int n_parts,
                                                           const int* __restrict__ coop, float* __restrict__ ebuf,
                                                           float* __restrict__ agg_out, int N, int B, int hid, int ldb) {
  const int s = blockIdx.z, i = blockIdx.y;
  if (!coop[i]) return;
  const int b = blockIdx.x * WSEL_ROWS + threadIdx.x;
  if (b >= B) return;
  const float* __restrict__ e0 = est + ((long)s * N + i) * (D + 1) * ldb + b;
  float v[D];
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = e0[(long)k * ldb];
  const float nrm = w_norm1(phi, nparts, n_parts, (long)s * N + i, b, hid, ldb);
  float lo, hi;
  SelNet<D, H>::run(v, lo, hi);
  const float lower = fminf(lo, v[0]), upper = fmaxf(hi, v[0]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) sum += __builtin_amdgcn_fmed3f(v[k], lower, upper);
  const float agg = sum / (float)D;
  const long o = ((long)s * N + i) * ldb + b;
  ebuf[o] = (agg - e0[(long)D * ldb]) / nrm;
  if (agg_out) agg_out[o] = agg;
}

// residual toward a caller-supplied aggregate (critic_update_team(s, agg), :60-71): e = (agg - V_live)/(|phi|^2+1)
__global__ __launch_bounds__(256) void k_wresidual(const float* __restrict__ phi, const float* __restrict__ theta,
                                                   const float* __restrict__ agg_in, const int* __restrict__ coop,
                                                   float* __restrict__ ebuf, int N, int B, int in_dim, int hid, int ldp,
                                                   int ldb) {
  const int s = blockIdx.z, i = blockIdx.y, b = blockIdx.x * WROWS + threadIdx.x;
  if (!coop[i] || b >= B) return;
  const NetGeom g = make_geom(in_dim, hid, 1);
  const float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const float* __restrict__ col = phi + ((long)s * N + i) * hid * ldb + b;
  float nrm = 0.f, v = 0.f;
  for (int k = 0; k < hid; ++k) {
    const float p = col[(long)k * ldb];
    nrm = fmaf(p, p, nrm);
    v = fmaf(p, th[g.o_W3 + k], v);
  }
  nrm += 1.0f;
  v += th[g.o_b3];
  const long o = ((long)s * N + i) * ldb + b;
  ebuf[o] = (agg_in[o] - v) / nrm;
}

__global__ __launch_bounds__(256) void k_whead_apply(const float* __restrict__ grads, float* __restrict__ theta,
                                                     const int* __restrict__ coop, int N, int B, int in_dim, int hid,
                                                     int ldp) {
  const int s = blockIdx.y, i = blockIdx.x;
  if (!coop[i]) return;
  const NetGeom g = make_geom(in_dim, hid, 1);
  float* __restrict__ th = theta + ((long)s * N + i) * ldp;
  const float* __restrict__ gr = grads + ((long)s * N + i) * w_grad_size(hid);
  for (int e = threadIdx.x; e <= hid; e += 256) {
    const int o = (e < hid) ? g.o_W3 + e : g.o_b3;
    th[o] = th[o] + gr[e] / (float)B;
  }
}

static inline bool w_dims_ok(int S, int N, int B, int K, int J, int ldp, int ldb) {
  return S > 0 && N > 0 && B > 0 && K > 0 && J > 0 && ldp > 0 && ldb >= B;
}

}  // namespace

RCMARL_EXPORT int rcmarl_wide_grad_size(int hid) { return w_grad_size(hid); }
RCMARL_EXPORT int rcmarl_wide_f16_mode() { return w16_mode(); }
RCMARL_EXPORT int rcmarl_wide_set_f16_mode(int mode) {           // 0 / 1; < 0: read RCMARL_WIDE_F16 again at the next call
  std::lock_guard<std::mutex> lk(g_w16_mu);
  g_w16_mode = mode < 0 ? -1 : (mode ? 1 : 0);
  return RCMARL_OK;
}
RCMARL_EXPORT int rcmarl_wide_rows_per_chunk() { return WROWS; }

// out[s][n][j][b] = lrelu(sum_k W[k][j] in(k, b) + bias[j]),  W = theta[s][n] + w_off (K x J, row-major),
// bias = theta[s][n] + b_off.  in: row_major != 0: in[s][b][k] (replay rows, ld_in floats per row, agent stride
// in_agent_stride -- 0 for the shared global state); row_major == 0: in[s][(n)][k][b] feature-major.
RCMARL_EXPORT int rcmarl_dense_forward(const float* in, long in_seed_stride, long in_agent_stride, int in_row_major,
                                       int ld_in, const float* theta, int w_off, int b_off, float* out, int S, int N,
                                       int B, int K, int J, int ldp, int ldb, void* stream) {
  if (!in || !theta || !out || !w_dims_ok(S, N, B, K, J, ldp, ldb) || w_off < 0 || b_off < 0 || ld_in <= 0)
    return RCMARL_ERR_ARG;
  WArgs a{};
  a.A = theta + w_off; a.A_zs = (long)N * ldp; a.A_za = ldp; a.lda = J;
  a.B = in; a.B_zs = in_seed_stride; a.B_za = in_agent_stride; a.ldb = ld_in;
  a.C = out; a.C_zs = (long)N * J * ldb; a.C_za = (long)J * ldb; a.ldc = ldb;
  a.aux = theta + b_off; a.aux_zs = (long)N * ldp; a.aux_za = ldp; a.ldaux = 0;
  a.M = J; a.N = B; a.K = K; a.NA = N;
  a.sa = RC_F16_W_SCALE; a.sb = in_row_major ? 0.f : RC_W16_ACT_SCALE;   // weights x activations (feature-major); raw inputs stay on the fp32 kernel
  return in_row_major ? w_launch<false, true, WEPI_BIAS_LRELU>(a, S, stream)
                      : w_launch<false, false, WEPI_BIAS_LRELU>(a, S, stream);
}

// dz_in[s][n][k][b] = (sum_j W[k][j] dz_out[j][b]) * lrelu'(act_in[k][b])
RCMARL_EXPORT int rcmarl_dense_backward_data(const float* dz_out, const float* theta, int w_off, const float* act_in,
                                             float* dz_in, int S, int N, int B, int K, int J, int ldp, int ldb,
                                             void* stream) {
  if (!dz_out || !theta || !act_in || !dz_in || !w_dims_ok(S, N, B, K, J, ldp, ldb) || w_off < 0) return RCMARL_ERR_ARG;
  WArgs a{};
  a.A = theta + w_off; a.A_zs = (long)N * ldp; a.A_za = ldp; a.lda = J;                 // A(m=k, kk=j) = W[k*J + j]
  a.B = dz_out; a.B_zs = (long)N * J * ldb; a.B_za = (long)J * ldb; a.ldb = ldb;
  a.C = dz_in; a.C_zs = (long)N * K * ldb; a.C_za = (long)K * ldb; a.ldc = ldb;
  a.aux = act_in; a.aux_zs = a.C_zs; a.aux_za = a.C_za; a.ldaux = ldb;
  a.M = K; a.N = B; a.K = J; a.NA = N;
  a.sa = RC_F16_W_SCALE; a.sb = RC_F16_DZ_SCALE;                   // weights x dz
  return w_launch<true, false, WEPI_LRELU_GRAD>(a, S, stream);
}

// W[k][j] -= lr * sum_b in(k, b) dz[j][b]   (agents with mask[n] == 0 are skipped; mask may be NULL)
RCMARL_EXPORT int rcmarl_dense_backward_sgd(const float* in, long in_seed_stride, long in_agent_stride, int in_row_major,
                                            int ld_in, const float* dz, float* theta, int w_off, const int* mask, int S,
                                            int N, int B, int K, int J, int ldp, int ldb, float lr, void* stream) {
  if (!in || !dz || !theta || !w_dims_ok(S, N, B, K, J, ldp, ldb) || w_off < 0 || ld_in <= 0) return RCMARL_ERR_ARG;
  WArgs a{};
  a.A = in; a.A_zs = in_seed_stride; a.A_za = in_agent_stride; a.lda = ld_in;           // A(m=k, kk=b)
  a.B = dz; a.B_zs = (long)N * J * ldb; a.B_za = (long)J * ldb; a.ldb = ldb;            // B(kk=b, n=j) = dz[j*ldb + b]
  a.C = theta + w_off; a.C_zs = (long)N * ldp; a.C_za = ldp; a.ldc = J;
  a.mask = mask; a.lr = lr;
  a.M = K; a.N = J; a.K = B; a.NA = N;
  a.sa = in_row_major ? 0.f : RC_W16_ACT_SCALE; a.sb = RC_F16_DZ_SCALE;   // activations (feature-major) x dz
  return in_row_major ? w_launch<false, true, WEPI_SGD>(a, S, stream) : w_launch<true, true, WEPI_SGD>(a, S, stream);
}

RCMARL_EXPORT int rcmarl_wide_head_value(const float* a2, const float* theta, const float* r_applied, float gamma,
                                         float* out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                                         void* stream) {
  if (!a2 || !theta || !out || !w_dims_ok(S, N, B, in_dim, hid, ldp, ldb)) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(B, WROWS), N, S), block(256);
  RCMARL_LAUNCH(k_whead_value, grid, block, 0, stream, a2, theta, r_applied, gamma, out, N, B, in_dim, hid, ldp, ldb);
  return rcmarl_check_launch();
}

// a2 (in) is overwritten by dz2; dz3: [S][N][ldb] scratch; grads: [S][N][rcmarl_wide_grad_size(hid)];
// losspart: [S][N][ceil(B / rows_per_chunk)]
RCMARL_EXPORT int rcmarl_wide_head_fit(float* a2, const float* theta, const float* y, float* dz3, float* grads,
                                       float* losspart, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                                       void* stream) {
  if (!a2 || !theta || !y || !dz3 || !grads || !losspart || !w_dims_ok(S, N, B, in_dim, hid, ldp, ldb))
    return RCMARL_ERR_ARG;
  const int nchunk = rc_ceil_div(B, WROWS);
  const dim3 gc(nchunk, N, S), gr(rc_ceil_div(hid + 1, 4), N, S), block(256);
  RCMARL_LAUNCH(k_whead_fit_cols, gc, block, 0, stream, (const float*)a2, theta, y, dz3, losspart, N, B, in_dim, hid,
                ldp, ldb, nchunk);
  RCMARL_LAUNCH((k_wrows<WROW_FIT>), gr, block, 0, stream, a2, (const float*)dz3, theta, (const int*)nullptr, grads, 0,
                N, B, in_dim, hid, ldp, ldb);
  return rcmarl_check_launch();
}

// grads[s][n][2*hid + 1 + j] = sum_b dz1[j][b]
RCMARL_EXPORT int rcmarl_wide_bias_grad(const float* dz1, float* grads, int S, int N, int B, int hid, int ldb,
                                        void* stream) {
  if (!dz1 || !grads || S <= 0 || N <= 0 || B <= 0 || hid <= 0 || ldb < B) return RCMARL_ERR_ARG;
  const dim3 gr(rc_ceil_div(hid, 4), N, S), block(256);
  RCMARL_LAUNCH((k_wrows<WROW_SUM>), gr, block, 0, stream, const_cast<float*>(dz1), (const float*)nullptr,
                (const float*)nullptr, (const int*)nullptr, grads, 2 * hid + 1, N, B, 0, hid, 0, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_wide_small_sgd(const float* grads, const float* losspart, float* theta, const int* mask,
                                        float* loss_out, int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                        void* stream) {
  if (!grads || !losspart || !theta || S <= 0 || N <= 0 || B <= 0 || hid <= 0) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_wsmall_sgd, dim3(N, S), dim3(256), 0, stream, grads, losspart, theta, mask, loss_out, N, B, in_dim,
                hid, ldp, rc_ceil_div(B, WROWS), lr);
  return rcmarl_check_launch();
}

// phi: features of the live net (layer-2 activations, [S][N*hid][ldb]).  Scratch: hmat [S][N][d+1][hid],
// hb [S][N][d+1], est [S][N][d+1][ldb], ebuf [S][N][ldb].  Output: grads[s][n][0..hid] = [sum_b e phi | sum_b e]
// for cooperative agents; agg_out (optional) [S][N][ldb].  agg_in != NULL: projection toward that aggregate only
// (K3), nbr/msg/hmat/hb/est unused.
static int wide_consensus_head_impl(const float* phi, const float* nparts, int n_parts, const float* theta, const float* msg, const int* nbr,
                                    const int* coop, const float* agg_in, float* hmat, float* hb, float* est,
                                    float* ebuf, float* grads, float* agg_out, int S, int N, int B, int in_dim,
                                    int hid, int ldp, int ldb, int d, int H, void* stream) {
  if (nparts != nullptr && n_parts <= 0) return RCMARL_ERR_ARG;
  if (!phi || !theta || !coop || !ebuf || !grads || !w_dims_ok(S, N, B, in_dim, hid, ldp, ldb)) return RCMARL_ERR_ARG;
  const dim3 gc(rc_ceil_div(B, WROWS), N, S), block(256);
  if (agg_in) {
    RCMARL_LAUNCH(k_wresidual, gc, block, 0, stream, phi, theta, agg_in, coop, ebuf, N, B, in_dim, hid, ldp, ldb);
  } else {
    if (!msg || !nbr || !hmat || !hb || !est || d <= 0 || H < 0 || d < 2 * H + 1) return RCMARL_ERR_ARG;
    if ((size_t)d * WSEL_ROWS * sizeof(float) > 64 * 1024) return RCMARL_ERR_UNSUPPORTED;
    RCMARL_LAUNCH(k_wgather_heads, dim3(d + 1, N, S), block, 0, stream, theta, msg, nbr, coop, hmat, hb, N, in_dim, hid,
                  ldp, d);
    WArgs a{};
    a.A = hmat; a.A_zs = (long)N * (d + 1) * hid; a.A_za = (long)(d + 1) * hid; a.lda = hid;       // A(m, k) k-contiguous
    a.B = phi; a.B_zs = (long)N * hid * ldb; a.B_za = (long)hid * ldb; a.ldb = ldb;
    a.C = est; a.C_zs = (long)N * (d + 1) * ldb; a.C_za = (long)(d + 1) * ldb; a.ldc = ldb;
    a.aux = hb; a.aux_zs = (long)N * (d + 1); a.aux_za = d + 1;
    a.M = d + 1; a.N = B; a.K = hid; a.NA = N;
    const int rc = w_launch<true, false, WEPI_BIAS>(a, S, stream);
    if (rc != RCMARL_OK) return rc;
    bool done = false;
#define RC_WSEL_CASE(DD, HH)                                                                                        \
    if (!done && d == DD && H == HH) {                                                                               \
      RCMARL_LAUNCH((k_wselect_net<DD, HH>), dim3(rc_ceil_div(B, WSEL_ROWS), N, S), dim3(WSEL_ROWS), 0, stream,      \
                    (const float*)est, phi, nparts, n_parts, coop, ebuf, agg_out, N, B, hid, ldb);                  \
      done = true;                                                                                                   \
    }
    RCMARL_SELNET_COMBOS(RC_WSEL_CASE)
#undef RC_WSEL_CASE
    if (!done)
      RCMARL_LAUNCH(k_wselect, dim3(rc_ceil_div(B, WSEL_ROWS), N, S), dim3(WSEL_ROWS), (size_t)d * WSEL_ROWS * sizeof(float),
                    stream, (const float*)est, phi, nparts, n_parts, coop, ebuf, agg_out, N, B, hid, ldb, d, H);
  }
  RCMARL_LAUNCH((k_wrows<WROW_DOT>), dim3(rc_ceil_div(hid + 1, 4), N, S), block, 0, stream, const_cast<float*>(phi),
                (const float*)ebuf, (const float*)nullptr, coop, grads, 0, N, B, in_dim, hid, ldp, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_wide_consensus_head(const float* phi, const float* theta, const float* msg, const int* nbr,
                                             const int* coop, const float* agg_in, float* hmat, float* hb, float* est,
                                             float* ebuf, float* grads, float* agg_out, int S, int N, int B, int in_dim,
                                             int hid, int ldp, int ldb, int d, int H, void* stream) {
  return wide_consensus_head_impl(phi, nullptr, 0, theta, msg, nbr, coop, agg_in, hmat, hb, est, ebuf, grads, agg_out, S, N, B, in_dim, hid,
                                  ldp, ldb, d, H, stream);
}
// the same with |phi|^2 of every replay row supplied as n_parts per-tile parts nparts[s][n][n_parts][ldb] (rcmarl_pk_forward2 writes
// them beside the fp32 activations): the selection pass then reads d + 1 estimates per row instead of d + 1 + hid values
RCMARL_EXPORT int rcmarl_wide_consensus_head_nrm(const float* phi, const float* nparts, int n_parts, const float* theta, const float* msg,
                                                 const int* nbr, const int* coop, float* hmat, float* hb, float* est, float* ebuf,
                                                 float* grads, float* agg_out, int S, int N, int B, int in_dim, int hid, int ldp,
                                                 int ldb, int d, int H, void* stream) {
  if (!nparts) return RCMARL_ERR_ARG;
  return wide_consensus_head_impl(phi, nparts, n_parts, theta, msg, nbr, coop, nullptr, hmat, hb, est, ebuf, grads, agg_out, S, N, B, in_dim,
                                  hid, ldp, ldb, d, H, stream);
}

RCMARL_EXPORT int rcmarl_wide_head_apply(const float* grads, float* theta, const int* coop, int S, int N, int B,
                                         int in_dim, int hid, int ldp, void* stream) {
  if (!grads || !theta || !coop || S <= 0 || N <= 0 || B <= 0 || hid <= 0) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_whead_apply, dim3(N, S), dim3(256), 0, stream, grads, theta, coop, N, B, in_dim, hid, ldp);
  return rcmarl_check_launch();
}
