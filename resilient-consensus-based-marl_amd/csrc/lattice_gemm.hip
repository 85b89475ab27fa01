// Layer-1 GEMMs on the bf16 matrix core, exact ("lattice bf16x3" path; see rcmarl_lattice.h for
// the arithmetic argument and the packed operand format).
//
//   rcmarl_lattice_encode              replay tensor X (fp32) -> integer lattice K, packed twice:
//                                      Kp  (rows = replay row b, reduction = feature)   forward  B operand
//                                      KTp (rows = feature,      reduction = b)         backward A operand
//   rcmarl_w1_split                    W1 of every agent -> three bf16 pieces of alpha_k*W1, packed
//                                      (rows = (agent,unit) column, reduction = feature) forward A operand
//   rcmarl_layer1_forward_lattice      a1t = lrelu(K W' + b1)         = rcmarl_layer1_forward
//   rcmarl_layer1_backward_sgd_lattice W1 -= lr * alpha_k * K^T dz1   = rcmarl_layer1_backward_sgd
//                                      (dz1 pieces are emitted packed by rcmarl_mid_fit_lattice)
//
// One GEMM kernel serves both: D[m][n] = sum_{pa,pb} sum_k A_pa[m][k] B_pb[n][k], operands in PK
// format, 256-thread workgroups (2x2 wavefronts), block tile (64 MT) x (64 NT), k-tile 32, two LDS
// stages filled by global_load_lds_dwordx4 (linear 1-KiB bursts: the swizzle lives in the packed
// format), one barrier per k-tile, v_mfma_f32_32x32x16_bf16.
//   forward : PA=3 (W' pieces), PB=1 (K),  tile 128 x 256  -> per k16 step 10 ds_read_b128 : 24 MFMA
//   backward: PA=1 (K^T),       PB=3 (dz), tile 256 x 128
// Workgroups are numbered so that all tiles of one seed run on ONE XCD (block b -> XCD b%8): the
// seed's small operand (K, 3 MB) stays in that XCD's L2 and the big one streams through once.
#include "rcmarl_lattice.h"
#include "rcmarl_lat_mainloop.h"
#include <type_traits>
#include <stdlib.h>

namespace {

__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }

// ---------------------------------------------------------------------------------------------
// encode: one workgroup = 32 replay rows x 128 features, transposed through LDS so that both packed
// images are written in 16-byte chunks.
__global__ __launch_bounds__(256) void k_lattice_encode(const float* __restrict__ x, long x_seed_stride,
                                                        const float* __restrict__ alpha, int B, int in_dim,
                                                        unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                        unsigned char* __restrict__ ktp, int ktp_rt, int ktp_kt,
                                                        int* __restrict__ flag, int kp_f16, int ktp_f16) {
  __shared__ unsigned short tileB[32][128 + 2], tileH[32][128 + 2];    // the integers as bf16 / as f16 bits (|K| <= 256: exact in both)
  const int s = blockIdx.z, b0 = blockIdx.y * 32, c0 = blockIdx.x * 128;
  const int t = threadIdx.x;
  {
    const int cl = t & 127, c = c0 + cl;
    const float al = c < in_dim ? alpha[c] : 1.f;
    bool bad = false;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int bl = (t >> 7) + 2 * i, b = b0 + bl;
      float kq = 0.f;
      if (b < B && c < in_dim) {
        const float xv = x[(long)s * x_seed_stride + (long)b * in_dim + c];
        kq = rintf(xv / al);
        // lattice property: x == alpha*K up to fp32 roundoff, |K| <= 256 (exact in bf16)
        if (!(fabsf(kq) <= 256.f) || !(fabsf(fmaf(kq, al, -xv)) <= 4.76837158e-7f * fabsf(xv))) bad = true;
      }
      tileB[bl][cl] = (unsigned short)rc_bf16_rne(kq);
      tileH[bl][cl] = (unsigned short)rc_f16_rne(kq);
    }
    if (bad) *flag = 1;
  }
  __syncthreads();
  // KTp block (row tile c0/128, k-tile b0/32): 128 rows x 4 chunks
  if (ktp != nullptr && (c0 >> 7) < ktp_rt && (b0 >> 5) < ktp_kt) {
    const unsigned short(*tile)[128 + 2] = ktp_f16 ? tileH : tileB;
    unsigned char* blk = ktp + (long)s * ktp_rt * ktp_kt * RC_PK_BLOCK + ((long)(c0 >> 7) * ktp_kt + (b0 >> 5)) * RC_PK_BLOCK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = t + 256 * q, r = idx >> 2, c4 = idx & 3;
      uint4 v;
      v.x = pack2(tile[8 * c4 + 0][r], tile[8 * c4 + 1][r]);
      v.y = pack2(tile[8 * c4 + 2][r], tile[8 * c4 + 3][r]);
      v.z = pack2(tile[8 * c4 + 4][r], tile[8 * c4 + 5][r]);
      v.w = pack2(tile[8 * c4 + 6][r], tile[8 * c4 + 7][r]);
      st_u4(blk + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), v);
    }
  }
  // Kp: rows b0..b0+31 of row tile b0/128, k-tiles c0/32 .. +3
  if (kp != nullptr && (b0 >> 7) < kp_rt) {
    const unsigned short(*tile)[128 + 2] = kp_f16 ? tileH : tileB;
    unsigned char* base = kp + (long)s * kp_rt * kp_kt * RC_PK_BLOCK + (long)(b0 >> 7) * kp_kt * RC_PK_BLOCK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = t + 256 * q, bl = idx >> 4, ch = idx & 15;
      const int kt = (c0 >> 5) + (ch >> 2), c4 = ch & 3, r = (b0 & 127) + bl;
      if (kt < kp_kt) {
        const unsigned short* src = &tile[bl][8 * ch];
        uint4 v;
        v.x = pack2(src[0], src[1]); v.y = pack2(src[2], src[3]);
        v.z = pack2(src[4], src[5]); v.w = pack2(src[6], src[7]);
        st_u4(base + (long)kt * RC_PK_BLOCK + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), v);
      }
    }
  }
}

// the pieces of eight consecutive-k values as one 16-byte chunk per piece plane (8 KiB apart); F16: of scale * w
template <bool F16>
__device__ __forceinline__ void store_pieces(unsigned char* dst, const float (&w)[8], float scale) {
  if constexpr (F16) {
    uint4 vh, vl;
    rc_split2h_pair(w[0] * scale, w[1] * scale, vh.x, vl.x);
    rc_split2h_pair(w[2] * scale, w[3] * scale, vh.y, vl.y);
    rc_split2h_pair(w[4] * scale, w[5] * scale, vh.z, vl.z);
    rc_split2h_pair(w[6] * scale, w[7] * scale, vh.w, vl.w);
    st_u4(dst, vh);
    st_u4(dst + RC_PK_BLOCK, vl);
  } else {
    uint4 vh, vm, vl;
    rc_split3_pair(w[0], w[1], vh.x, vm.x, vl.x);
    rc_split3_pair(w[2], w[3], vh.y, vm.y, vl.y);
    rc_split3_pair(w[4], w[5], vh.z, vm.z, vl.z);
    rc_split3_pair(w[6], w[7], vh.w, vm.w, vl.w);
    st_u4(dst, vh);
    st_u4(dst + RC_PK_BLOCK, vm);
    st_u4(dst + 2 * RC_PK_BLOCK, vl);
  }
}

// ---------------------------------------------------------------------------------------------
// W1 split: one workgroup = 128 (agent,unit) columns x 32 features -> three 8-KiB blocks (F16: two, of 2^10 alpha W1).
template <bool F16>
__global__ __launch_bounds__(256) void k_w1_split(const float* __restrict__ theta, const float* __restrict__ alpha,
                                                  unsigned char* __restrict__ wp, int N, int in_dim, int ldp,
                                                  int wp_rt, int wp_kt, int hid) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, c4 = t & 3;               // four adjacent lanes = the four 16-byte chunks of one 64-byte packed row: a store
  const int ncols = N * hid;                          // instruction touches 16 rows x 64 B instead of 64 rows x 16 B (a quarter of the
  constexpr int NP = F16 ? 2 : 3;                      // cache lines; the store path is paced by lines touched)
  if (F16) rc_f16_saturate();
  unsigned char* blk = wp + (long)s * wp_rt * wp_kt * NP * RC_PK_BLOCK + ((long)rt * wp_kt + kt) * NP * RC_PK_BLOCK;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = (t >> 2) + 64 * q;
    const int col = rt * 128 + r;
    const bool col_ok = col < ncols;
    const int ag = col_ok ? col / hid : 0, j = col - ag * hid;
    const float* th = theta + ((long)s * N + ag) * ldp + j;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kt * 32 + 8 * c4 + e;
      const bool ok = col_ok && k < in_dim;
      w[e] = ok ? th[(long)k * hid] * alpha[k] : 0.f;
    }
    store_pieces<F16>(blk + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), w, RC_F16_W_SCALE);
  }
}

// ---------------------------------------------------------------------------------------------
// dz pack: fp32 feature-major dz[S][rows][ldb] (rows = (agent,unit)) -> three exact bf16 pieces in PK form
// (reduction = replay row), zero beyond B / beyond the last row: the backward operand of a wide net, whose dz1 comes
// out of a dense GEMM (wide_kernels.hip) instead of mid_fit's fused epilogue.  One workgroup = 128 rows x 32 rows b.
template <bool F16>
__global__ __launch_bounds__(256) void k_dz_pack(const float* __restrict__ dz, unsigned char* __restrict__ dzp, int nrows,
                                                 int B, int ldb, int dzp_rt, int dzp_kt) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, c4 = t & 3;             // four adjacent lanes = the 128 B (32 k) of one row: coalesced both ways
  constexpr int NP = F16 ? 2 : 3;
  if (F16) rc_f16_saturate();
  unsigned char* blk = dzp + (long)s * dzp_rt * dzp_kt * NP * RC_PK_BLOCK + ((long)rt * dzp_kt + kt) * NP * RC_PK_BLOCK;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = (t >> 2) + 64 * q;
    const int row = rt * 128 + r;
    const bool row_ok = row < nrows;
    const float* src = dz + ((long)s * nrows + (row_ok ? row : 0)) * ldb + kt * 32 + 8 * c4;
    float w[8];
    if (row_ok && kt * 32 + 8 * c4 + 8 <= B) {         // ldb is a multiple of 64 floats: 16-B aligned
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (row_ok && kt * 32 + 8 * c4 + e < B) ? src[e] : 0.f;
    }
    store_pieces<F16>(blk + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), w, RC_F16_DZ_SCALE);
  }
}

// k_dz_pack + the row sums of dz (gb1 of a wide net = sum over replay rows of dz1, rcmarl_wide_bias_grad) in ONE pass over dz: one
// workgroup = 128 rows x ALL k-tiles (so a row's sum is formed by one thread quad in a fixed order: deterministic, no atomics).
// sums[((s * N + n) * sums_ld) + sums_off + j] for row n * hid + j.
template <bool F16>
__global__ __launch_bounds__(256) void k_dz_pack_rowsum(const float* __restrict__ dz, unsigned char* __restrict__ dzp,
                                                        float* __restrict__ sums, int sums_ld, int sums_off, int hid, int nrows,
                                                        int B, int ldb, int dzp_rt, int dzp_kt, int kts) {
  const int s = blockIdx.y, rt = blockIdx.x;
  const int t = threadIdx.x, c4 = t & 3;
  constexpr int NP = F16 ? 2 : 3;
  if (F16) rc_f16_saturate();
  float acc[2] = {0.f, 0.f};
  for (int kt = 0; kt < kts; ++kt) {
    unsigned char* blk = dzp + (long)s * dzp_rt * dzp_kt * NP * RC_PK_BLOCK + ((long)rt * dzp_kt + kt) * NP * RC_PK_BLOCK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (t >> 2) + 64 * q;
      const int row = rt * 128 + r;
      const bool row_ok = row < nrows;
      const float* src = dz + ((long)s * nrows + (row_ok ? row : 0)) * ldb + kt * 32 + 8 * c4;
      float w[8];
      if (row_ok && kt * 32 + 8 * c4 + 8 <= B) {         // ldb is a multiple of 64 floats: 16-B aligned
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (row_ok && kt * 32 + 8 * c4 + e < B) ? src[e] : 0.f;
      }
      acc[q] += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      store_pieces<F16>(blk + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), w, RC_F16_DZ_SCALE);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float v = acc[q];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    const int row = rt * 128 + (t >> 2) + 64 * q;
    if (c4 == 0 && row < nrows) {
      const int n = row / hid, j = row - n * hid;
      sums[((long)s * (nrows / hid) + n) * sums_ld + sums_off + j] = v;
    }
  }
}

// ---- forward: A = W' pieces (rows = (agent,unit) columns), B = K (rows = replay rows) ----------
// 128 x 256 block tile, two 40-KiB stages, two workgroups per CU.  W8: eight wavefronts of 64 x 64 (four per SIMD) instead
// of four of 64 x 128: measured 749 / 988 us against 776 / 1037 at the two cfg-4 shapes -> the forward's default.
// PK (wide networks, hid a multiple of 128, two-piece f16 form; dense_pk.hip consumes them): instead of the fp32 activations the
// epilogue writes what the NEXT GEMMs read -- the activations as two f16 pieces of 2^6 a1 in packed form in BOTH orientations
// (a1_bk: rows = replay row, reduction = unit, for layer 2's forward; a1_kb: rows = unit, reduction = replay row, for layer 2's weight
// gradient) and the sign bits of a1 (s1: one 32-bit word per (unit, 32 replay rows), for LeakyReLU' in the backward pass).
struct LatPkOut {
  unsigned char* a1_bk; int bk_rt;      // [S][N][bk_rt][hid/32][2][8 KiB]
  unsigned char* a1_kb; int kb_kt;      // [S][N][hid/128][kb_kt][2][8 KiB]
  unsigned* s1; int s1_ld;              // [S][N*hid][s1_ld]
  int* ovf;                             // set to 1 when an activation leaves the f16 range of 2^6 a1 (|a1| > 1015): the pieces saturate
};

template <bool W8, bool F16, bool PK = false>
__global__ RC_LAT_OCC(W8 ? 512 : 256, W8 ? 4 : 2)
void k_lat_forward(const unsigned char* __restrict__ wp, int wp_rt, int wp_kt, const unsigned char* __restrict__ kp, int kp_rt,
                   int kp_kt, const float* __restrict__ theta, float* __restrict__ a1t, int S, int N, int B, int in_dim, int ldp,
                   int ldb, int mtiles, int ntiles, int hid, const LatPkOut pk) {
  constexpr int PA = F16 ? 2 : 3, PB = 1, MT = 2, NT = W8 ? 2 : 4, WM = 2, WN = W8 ? 4 : 2;
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  RCMARL_DYN_SMEM(unsigned char, lds);
  int s, w;
  lat_decode(blockIdx.x, mtiles * ntiles, S, s, w);
#ifdef RC_LAT_STAGGER
  if (blockIdx.x >= 256 && blockIdx.x < 512) for (int i = 0; i < RC_LAT_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int bn = w % ntiles, bm = w / ntiles;                       // n fastest
  LatOperands op;
  op.a = wp + (long)s * wp_rt * wp_kt * (PA * RC_PK_BLOCK); op.a_kt = wp_kt; op.art0 = bm * C::ART;
  op.b = kp + (long)s * kp_rt * kp_kt * (PB * RC_PK_BLOCK); op.b_kt = kp_kt; op.brt0 = bn * C::BRT;
  rc_f32x16 acc[MT][NT];
#if defined(RC_LAT_KNOCK) && RC_LAT_KNOCK == 2              // measurement builds only (tools/build_variant.py): no k-loop
  lat_mainloop<PA, PB, MT, NT, WM, WN, false, F16>(op, ldb == 12345 ? 1 : 0, lds, acc);
#else
  lat_mainloop<PA, PB, MT, NT, WM, WN, false, F16>(op, (in_dim + 31) >> 5, lds, acc);
#endif
  // epilogue: a1t[col][b] = lrelu(z + b1[col])
  const int ncols = N * hid;
  const float* theta_s = theta + (long)s * N * ldp;
  float* a1t_s = a1t + (long)s * ncols * ldb;
  __syncthreads();
  float* bias = reinterpret_cast<float*>(lds);
  if (threadIdx.x < C::BM) {
    const int col = bm * C::BM + threadIdx.x;
    const int ag = col / hid;
    bias[threadIdx.x] = col < ncols ? theta_s[(long)ag * ldp + in_dim * hid + (col - ag * hid)] : 0.f;
  }
  __syncthreads();
  // Straight-line stores: the lane's 32 bias values come out of LDS in 8 ds_read_b128 up front (not one dependent
  // ds_read_b32 per element), row bases are wave-uniform (SGPR) and the lane part of the address is ONE 32-bit
  // offset, full tiles take no per-element predicate.  (The first version of this epilogue -- a branch, an LDS round
  // trip and a 64-bit multiply per element -- cost as much as the k-loop of a 512-deep GEMM: 176 of 832 us.)
  // (Round 5, built and measured, bit-identical, NOT faster: the four lanes of a quad transposing each 4 x 4 block in registers
  // (two DPP exchanges) so that a lane stores four consecutive replay rows of one column as one global_store_dwordx4 -- 16 stores of
  // eight full cache lines per wavefront tile instead of 64 of two: 496 / 696 us against 473 / 683, profiles/r05s_*; and the
  // operand-swapped kernel whose accumulators are born that way but whose 16-byte stores cover 32 rows: 876 us, profiles/r05d_*.
  // The NON-TEMPORAL hint of these stores is worth 16 %: plain stores 535 / 716 us against 460 / 675 -- they push the operand panels
  // out of the L2 (profiles/r05w_*).  The backward is the other way round: its theta / Wp stores as non-temporal ones cost +5..10 %
  // (profiles/r05x_*).)
  const int lane = threadIdx.x & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wm = wave / WN, wn = wave % WN;
  float bv[MT][16];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + wm * 32 * MT + 32 * mt + 8 * q + 4 * half);
      bv[mt][4 * q] = b4.x; bv[mt][4 * q + 1] = b4.y; bv[mt][4 * q + 2] = b4.z; bv[mt][4 * q + 3] = b4.w;
    }
  if constexpr (PK) {
    // one 128-column tile = 128 units of ONE agent (hid % 128 == 0); all of it is written, replay rows beyond B included
    // (finite values nobody weighs: the consumers' other operand is zero there)
    static_assert(F16, "packed activations exist in the two-piece f16 form");
    rc_f16_saturate();
    const int l31 = lane & 31;
    const int agent = (bm * C::BM) / hid, u0 = bm * C::BM - agent * hid, JK = hid >> 5, JT = hid >> 7;
    const long ag_lin = (long)s * N + agent;
    unsigned char* scratch = lds + 1024 + wave * 4096;              // (the bias values sit in the first 512 bytes)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n0 = bn * C::BN + wn * 32 * NT + 32 * nt, n = n0 + l31;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ublk = u0 + wm * 32 * MT + 32 * mt;               // first unit of this 32 x 32 block inside the agent
        unsigned ph[8], pl[8];
        unsigned myw = 0;
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float o[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * j + e;
            const float z = fmaf(acc[mt][nt][r], RC_F16_W_UNSCALE, bv[mt][r]);
            o[e] = fmaxf(z, RC_LEAK * z);
            if (pk.s1 != nullptr) {
              const unsigned long long bal = rc_ballot(o[e] > 0.f);
              const unsigned wsel = half ? (unsigned)(bal >> 32) : (unsigned)bal;
              if (l31 == r) myw = wsel;
            }
          }
          amax = rc_amax3(amax, o[0], o[1]);
          rc_split2h_pair(o[0] * RC_F16_ACT_SCALE, o[1] * RC_F16_ACT_SCALE, ph[j], pl[j]);
        }
        if (pk.ovf != nullptr && !(amax * RC_F16_ACT_SCALE <= 65000.f)) *pk.ovf = 1;
        if (pk.s1 != nullptr && l31 < 16) {
          const int col = bm * C::BM + wm * 32 * MT + 32 * mt + 8 * (l31 >> 2) + (l31 & 3) + 4 * half;
          pk.s1[((long)s * ncols + col) * pk.s1_ld + (n0 >> 5)] = myw;
        }
        if (pk.a1_bk != nullptr) {
          unsigned pc[2][4][2];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) { pc[0][q][dw] = ph[2 * q + dw]; pc[1][q][dw] = pl[2 * q + dw]; }
          unsigned char* rowp = pk.a1_bk + ((ag_lin * pk.bk_rt + (n >> 7)) * JK + (ublk >> 5)) * (2 * RC_PK_BLOCK) + (n & 127) * 64;
          pk_emit_rows_from_lanes<2>(pc, rowp, (n >> 2) & 3, (n >> 7) < pk.bk_rt);
        }
        if (pk.a1_kb != nullptr) {
          unsigned short h16[2][16];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            h16[0][2 * j] = (unsigned short)(ph[j] & 0xffffu); h16[0][2 * j + 1] = (unsigned short)(ph[j] >> 16);
            h16[1][2 * j] = (unsigned short)(pl[j] & 0xffffu); h16[1][2 * j + 1] = (unsigned short)(pl[j] >> 16);
          }
          if ((n0 >> 5) < pk.kb_kt) {                               // (wave-uniform)
            unsigned char* blk = pk.a1_kb + ((ag_lin * JT + (ublk >> 7)) * pk.kb_kt + (n0 >> 5)) * (2 * RC_PK_BLOCK);
            pk_emit_rows_from_regs<2>(h16, scratch, blk, ublk & 127);
          }
        }
      }
    }
    return;
  }
  const bool full_m = (bm + 1) * C::BM <= ncols;                   // workgroup-uniform
  auto store_tile = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = bn * C::BN + wn * 32 * NT + 32 * nt + (lane & 31);
#if defined(RC_LAT_KNOCK) && RC_LAT_KNOCK == 1              // measurement builds only: no stores
      if (n < B && ldb == 12345) {
#else
      if (n < B) {
#endif
        // byte offset of the lane inside its row block: 32 bits (one seed's a1t is < 4 GiB) -> saddr + voffset stores
        const unsigned lane_byte = ((unsigned)(4 * half) * (unsigned)ldb + (unsigned)n) * 4u;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m0 = bm * C::BM + wm * 32 * MT + 32 * mt;      // wave-uniform
          unsigned char* __restrict__ rowbase = reinterpret_cast<unsigned char*>(a1t_s + (long)m0 * ldb);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float z = F16 ? fmaf(acc[mt][nt][r], RC_F16_W_UNSCALE, bv[mt][r]) : acc[mt][nt][r] + bv[mt][r];
            const float o = fmaxf(z, RC_LEAK * z);                 // == rc_lrelu(z) bit for bit (0 < leak < 1), 2 ops not 3
            float* __restrict__ dst = reinterpret_cast<float*>(rowbase + (long)dr * ldb * 4 + lane_byte);
            if (FULL || m0 + dr + 4 * half < ncols) RC_NT_STORE(dst, o);
          }
        }
      }
    }
  };
  if (full_m) store_tile(std::true_type{}); else store_tile(std::false_type{});
}

#ifdef RC_LAT_PROTO   // measurement builds only: the round-5 forward shells that did not win (tools/prototypes/lattice_forward_shells.inc)
#define RC_LAT_PROTO_KERNELS
#include "../../tools/prototypes/lattice_forward_shells.inc"
#undef RC_LAT_PROTO_KERNELS
#else
#define RC_TSB(slot) ((void)0)
#endif

// ---- backward: A = K^T (rows = features), B = dz1 pieces (rows = (agent,unit) columns) ----------
// 256 x 128 block tile.  Default: four wavefronts of 128 x 64 with the spread LDS-DMA issue (875 -> 824 us in a block);
// W8 (eight wavefronts of 64 x 64): 780 / 1103 against 756 / 1103 us -- the alternative.
// M128 (four-wavefront form only): 128 x 128 block tiles for networks of at most 128 inputs (BASELINE configs[2]'s critic: 64 agents,
// 128 inputs) -- the 256-row tile would spend half its matrix work on rows beyond the input width.
template <bool W8, bool DZ16, bool WP16, bool M128 = false>
__global__ RC_LAT_OCC(W8 ? 512 : 256, W8 ? 4 : 2)
void k_lat_backward_sgd(const unsigned char* __restrict__ ktp, int ktp_rt, int ktp_kt, const unsigned char* __restrict__ dzp,
                        int dzp_rt, int dzp_kt, const float* __restrict__ alpha, float* __restrict__ theta,
                        const int* __restrict__ mask, int S, int N, int B, int in_dim, int ldp, float lr, int mtiles, int ntiles,
                        unsigned char* __restrict__ wp_out, int wp_rt, int wp_kt, int hid) {
  static_assert(!(W8 && M128), "the 128-row tile exists in the four-wavefront form");
  constexpr int PA = 1, PB = DZ16 ? 2 : 3, MT = W8 ? 2 : (M128 ? 2 : 4), NT = 2, WM = W8 ? 4 : 2, WN = 2;
  constexpr int WNP = WP16 ? 2 : 3;                                 // pieces of the forward operand written by the epilogue
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  RCMARL_DYN_SMEM(unsigned char, lds);
  int s, w;
  lat_decode(blockIdx.x, mtiles * ntiles, S, s, w);
#ifdef RC_LAT_STAGGER
  if (blockIdx.x >= 256 && blockIdx.x < 512) for (int i = 0; i < RC_LAT_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int bm = w % mtiles, bn = w / mtiles;                      // m fastest: neighbours share the dz panel
  LatOperands op;
  op.a = ktp + (long)s * ktp_rt * ktp_kt * (PA * RC_PK_BLOCK); op.a_kt = ktp_kt; op.art0 = bm * C::ART;
  op.b = dzp + (long)s * dzp_rt * dzp_kt * (PB * RC_PK_BLOCK); op.b_kt = dzp_kt; op.brt0 = bn * C::BRT;
  rc_f32x16 acc[MT][NT];
  RC_TSB(0);
#if defined(RC_LAT_KNOCK) && RC_LAT_KNOCK == 2
  lat_mainloop<PA, PB, MT, NT, WM, WN, !W8, DZ16>(op, ldp == 12345 ? 1 : 0, lds, acc);
#else
  lat_mainloop<PA, PB, MT, NT, WM, WN, !W8, DZ16>(op, (B + 31) >> 5, lds, acc);
#endif
  RC_TSB(1);
  // epilogue: W1[k][col] -= lr * alpha_k * acc; optionally the forward operand of the NEXT step is produced here
  // too (wp_out: bf16x3 pieces of alpha_k * W1_new, exactly what rcmarl_w1_split would write), so the local fit
  // needs no separate split pass.  A lane holds 4 consecutive k per (m-tile, register group) = half a 16-byte chunk.
  const int ncols = N * hid;
  if (WP16) rc_f16_saturate();
  __syncthreads();
  float* al = reinterpret_cast<float*>(lds);
  if (threadIdx.x < C::BM) {
    const int k = bm * C::BM + threadIdx.x;
    al[threadIdx.x] = k < in_dim ? alpha[k] : 0.f;
  }
  __syncthreads();
  // Two phases per n-tile: ALL 64 old weights of the lane are requested first (independent loads, one wait), then
  // the updates are computed and stored.  Written element by element (load, update, store the same address) the
  // compiler had to wait for every store before the next load -- possible alias -- i.e. 128 serial memory round trips
  // per lane, as long as the whole k-loop of the workgroup.
  const int lane = threadIdx.x & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wm = wave / WN, wn = wave % WN;
  const bool full_k = (bm + 1) * C::BM <= in_dim;                       // workgroup-uniform
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int cl = wn * 32 * NT + 32 * nt + (lane & 31);                 // column within the 128-wide tile
    const int col = bn * C::BN + cl;
#if defined(RC_LAT_KNOCK) && RC_LAT_KNOCK == 1
    const bool act = col < ncols && ldp == 12345;
#else
    const bool act = col < ncols;                                        // (the same for both lanes of a column: lane ^ 32)
#endif
    const int colc = act ? col : 0;
    const int ag = colc / hid, j = colc - ag * hid;
    const bool upd = act && (mask == nullptr || mask[ag]);
    // the lane's first row: k = bm*BM + wm*32*MT + 4*half; rows of (mt, gq, e) follow at uniform distances
    const int k0 = bm * C::BM + wm * 32 * MT + 4 * half;
    float* th = theta + ((long)s * N + ag) * ldp + j + (long)k0 * hid;
    // The forward operand of the next step: a lane's four consecutive k of one (mt, gq) are HALF a 16-byte chunk of the packed row
    // (the other half sits in lane ^ 32).  Round 5: the two lanes exchange halves (one v_permlane32_swap per dword) so that each
    // stores WHOLE chunks -- lane half 0 the even gq, half 1 the odd ones -- as 16-byte stores: half the store instructions and half
    // the cache lines touched per byte (the store path is paced by lines touched per instruction, profiles/r05d_*, r05f_*): 550 -> 500
    // and 802 -> 733 us at the cfg-4 shapes, bit-identical (profiles/r05m_*).  (Also built: theta itself as float4 accesses -- the
    // four lanes of a quad transpose their 4 x 4 old / new weights in registers by two DPP exchanges so that a lane holds one k and
    // four consecutive columns: bit-identical, 515-523 / 754 us, i.e. SLOWER than the dword accesses; profiles/r05n_*.  And a second
    // exchange, lane ^ 16 (v_permlane16_swap), after which a store covers 16 rows x 64 B instead of 32 rows x 32 B: bit-identical,
    // no further gain, 497 / 733 us; profiles/r05r_*.  What the operand still costs: 22 / 41 us of the launch, profiles/r05q_*.)
    unsigned char* wrow = wp_out == nullptr ? nullptr
        : wp_out + (long)s * wp_rt * wp_kt * (WNP * RC_PK_BLOCK) + (long)(colc >> 7) * wp_kt * (WNP * RC_PK_BLOCK) + (colc & 127) * 64;
    const int sw = (colc >> 2) & 3;
    float wold[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int dk = 32 * mt + 8 * (q >> 2) + (q & 3);               // uniform
        wold[mt][q] = (act && (full_k || k0 + dk < in_dim)) ? th[(long)dk * hid] : 0.f;
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int kt = bm * (C::BM / 32) + wm * MT + mt;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {                                   // gq = 2 gp, 2 gp + 1
        unsigned pc[2][WNP][2];                                          // [gq & 1][piece][dword]
#pragma unroll
        for (int g1 = 0; g1 < 2; ++g1) {
          const int gq = 2 * gp + g1;
          const float4 a4 = *reinterpret_cast<const float4*>(al + wm * 32 * MT + 32 * mt + 8 * gq + 4 * half);
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
          float wn4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int dk = 32 * mt + 8 * gq + e;
            float w = wold[mt][4 * gq + e];
            if (upd && (full_k || k0 + dk < in_dim)) {
              w = w - lr * (av[e] * acc[mt][nt][4 * gq + e]);
              th[(long)dk * hid] = w;
            }
            wn4[e] = w * av[e];
          }
          if constexpr (WP16) {
            rc_split2h_pair(wn4[0] * RC_F16_W_SCALE, wn4[1] * RC_F16_W_SCALE, pc[g1][0][0], pc[g1][1][0]);
            rc_split2h_pair(wn4[2] * RC_F16_W_SCALE, wn4[3] * RC_F16_W_SCALE, pc[g1][0][1], pc[g1][1][1]);
          } else {
            rc_split3_pair(wn4[0], wn4[1], pc[g1][0][0], pc[g1][1][0], pc[g1][2][0]);
            rc_split3_pair(wn4[2], wn4[3], pc[g1][0][1], pc[g1][1][1], pc[g1][2][1]);
          }
        }
        if (wp_out != nullptr && kt < wp_kt) {                            // (workgroup-uniform)
#pragma unroll
          for (int p = 0; p < WNP; ++p)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) rc_swap_halves(pc[0][p][dw], pc[1][p][dw]);
          // now: half 0 holds chunk gq = 2 gp (own k 0..3 in pc[0], the partner's k 4..7 in pc[1]); half 1 chunk 2 gp + 1 (the
          // partner's k 0..3 in pc[0], own k 4..7 in pc[1])
          if (act) {
            unsigned char* q = wrow + (long)kt * (WNP * RC_PK_BLOCK) + (((2 * gp + half) ^ sw) << 4);
#pragma unroll
            for (int p = 0; p < WNP; ++p)
            {
              uint4 v;
              v.x = pc[0][p][0]; v.y = pc[0][p][1]; v.z = pc[1][p][0]; v.w = pc[1][p][1];
              st_u4(q + p * RC_PK_BLOCK, v);
            }
          }
        }
      }
    }
    RC_TSB(2 + nt);
  }
#ifdef RC_LAT_TS
  RC_WAIT_VMEM();
  RC_TSB(4);
#endif
}

// RCMARL_LAT_W8=0 / 1 forces the four- / eight-wavefront form of both kernels (read at every call: tests switch it);
// default: forward eight, backward four (see the kernels' comments).
int lat_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
bool lat_w8(bool forward) { const char* e = getenv("RCMARL_LAT_W8"); return e ? atoi(e) != 0 : forward; }

template <bool W8, bool F16, bool PK = false>
int launch_forward(unsigned nblocks, void* stream, const unsigned char* wp, int wp_rt, int wp_kt, const unsigned char* kp, int kp_rt,
                   int kp_kt, const float* theta, float* a1t, int S, int N, int B, int in_dim, int ldp, int ldb, int mtiles,
                   int ntiles, int hid, const LatPkOut pk = LatPkOut{}) {
  const size_t smem = (size_t)2 * LatCfg<F16 ? 2 : 3, 1, 2, 4>::STAGE_BYTES;
  static const bool ok = rc_want_lds(k_lat_forward<W8, F16, PK>, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_lat_forward<W8, F16, PK>), dim3(nblocks), dim3(W8 ? 512 : 256), smem, stream, wp, wp_rt, wp_kt, kp, kp_rt, kp_kt,
                theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, hid, pk);
  return rcmarl_check_launch();
}

#ifdef RC_LAT_PROTO
#define RC_LAT_PROTO_LAUNCHERS
#include "../../tools/prototypes/lattice_forward_shells.inc"
#undef RC_LAT_PROTO_LAUNCHERS
#endif

template <bool W8, bool DZ16, bool WP16, bool M128 = false>
int launch_backward(unsigned nblocks, void* stream, const unsigned char* ktp, int ktp_rt, int ktp_kt, const unsigned char* dzp,
                    int dzp_rt, int dzp_kt, const float* alpha, float* theta, const int* mask, int S, int N, int B, int in_dim,
                    int ldp, float lr, int mtiles, int ntiles, unsigned char* wp_out, int wp_rt, int wp_kt, int hid) {
  const size_t smem = (size_t)2 * LatCfg<1, DZ16 ? 2 : 3, 4, 2>::STAGE_BYTES;
  static const bool ok = rc_want_lds(k_lat_backward_sgd<W8, DZ16, WP16, M128>, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_lat_backward_sgd<W8, DZ16, WP16, M128>), dim3(nblocks), dim3(W8 ? 512 : 256), smem, stream, ktp, ktp_rt, ktp_kt, dzp,
                dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, DZ16 ? lr * RC_F16_DZ_UNSCALE : lr, mtiles, ntiles,
                wp_out, wp_rt, wp_kt, hid);
  return rcmarl_check_launch();
}

}  // namespace

#ifdef RC_LAT_PROTO
#define RC_LAT_PROTO_DUMP
#include "../../tools/prototypes/lattice_forward_shells.inc"
#undef RC_LAT_PROTO_DUMP
#endif

// 0: exact bf16x3 everywhere; bit 0: forward operand (W', and K of the forward image) as f16x2; bit 1: backward operand (dz1, and
// K^T) as f16x2.  Callers that decode packed buffers (tests) ask here; the buffers are always sized for three pieces.

RCMARL_EXPORT int rcmarl_lattice_encode(const float* x, long x_seed_stride, const float* alpha, int S, int B, int in_dim,
                                        void* kp, int kp_rt, int kp_kt, void* ktp, int ktp_rt, int ktp_kt, int* flag,
                                        void* stream) {
  if (!x || !alpha || !flag || S <= 0 || B <= 0 || in_dim <= 0 || (!kp && !ktp)) return RCMARL_ERR_ARG;
  const int b_pad = rc_ceil_div(B, 256) * 256;
  if (kp && (kp_rt * 128 < b_pad || kp_kt * 32 < in_dim)) return RCMARL_ERR_ARG;
  if (ktp && (ktp_rt * 128 < in_dim || ktp_kt * 32 < b_pad)) return RCMARL_ERR_ARG;
  // features are swept up to the larger of the two images' extents so every block the GEMMs read is written
  int c_ext = in_dim;
  if (kp) c_ext = kp_kt * 32 > c_ext ? kp_kt * 32 : c_ext;
  if (ktp) c_ext = ktp_rt * 128 > c_ext ? ktp_rt * 128 : c_ext;
  const dim3 grid(rc_ceil_div(c_ext, 128), b_pad / 32, S), block(256);
  const int mode = rc_lat_f16_mode();
  rc_form_set(kp, mode & 1);
  rc_form_set(ktp, (mode >> 1) & 1);
  RCMARL_LAUNCH(k_lattice_encode, grid, block, 0, stream, x, x_seed_stride, alpha, B, in_dim, (unsigned char*)kp, kp_rt,
                kp_kt, (unsigned char*)ktp, ktp_rt, ktp_kt, flag, mode & 1, (mode >> 1) & 1);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_w1_split(const float* theta, const float* alpha, void* wp, int S, int N, int in_dim, int hid,
                                  int ldp, int wp_rt, int wp_kt, void* stream) {
  if (!theta || !alpha || !wp || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63) || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid <= 0) return RCMARL_ERR_ARG;
  if ((long)wp_rt * 128 < (long)N * hid || wp_kt * 32 < in_dim) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(in_dim, 32), rc_ceil_div(N * hid, 128), S), block(256);
  rc_form_set(wp, rc_lat_f16_mode() & 1);
  if (rc_lat_f16_mode() & 1) {
    RCMARL_LAUNCH(k_w1_split<true>, grid, block, 0, stream, theta, alpha, (unsigned char*)wp, N, in_dim, ldp, wp_rt, wp_kt, hid);
  } else {
    RCMARL_LAUNCH(k_w1_split<false>, grid, block, 0, stream, theta, alpha, (unsigned char*)wp, N, in_dim, ldp, wp_rt, wp_kt, hid);
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_lattice_pack_dz(const float* dz, void* dzp, int S, int N, int B, int hid, int ldb, int dzp_rt,
                                         int dzp_kt, void* stream) {
  if (!dz || !dzp || S <= 0 || N <= 0 || B <= 0 || hid <= 0 || ldb < B || (ldb & 3) ||
      (reinterpret_cast<uintptr_t>(dz) & 15))
    return RCMARL_ERR_ARG;
  const int rts = rc_ceil_div(N * hid, 128), kts = rc_ceil_div(B, 32);
  if (dzp_rt < rts || dzp_kt < kts) return RCMARL_ERR_ARG;
  rc_form_set(dzp, (rc_lat_f16_mode() >> 1) & 1);
  if (rc_lat_f16_mode() & 2) {
    RCMARL_LAUNCH(k_dz_pack<true>, dim3(kts, rts, S), dim3(256), 0, stream, dz, (unsigned char*)dzp, N * hid, B, ldb, dzp_rt, dzp_kt);
  } else {
    RCMARL_LAUNCH(k_dz_pack<false>, dim3(kts, rts, S), dim3(256), 0, stream, dz, (unsigned char*)dzp, N * hid, B, ldb, dzp_rt, dzp_kt);
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_lattice_pack_dz_rowsum(const float* dz, void* dzp, float* sums, int sums_ld, int sums_off, int S, int N,
                                                int B, int hid, int ldb, int dzp_rt, int dzp_kt, void* stream) {
  if (!dz || !dzp || !sums || S <= 0 || N <= 0 || B <= 0 || hid <= 0 || ldb < B || (ldb & 3) || sums_ld < sums_off + hid ||
      sums_off < 0 || (reinterpret_cast<uintptr_t>(dz) & 15))
    return RCMARL_ERR_ARG;
  const int rts = rc_ceil_div(N * hid, 128), kts = rc_ceil_div(B, 32);
  if (dzp_rt < rts || dzp_kt < kts) return RCMARL_ERR_ARG;
  rc_form_set(dzp, (rc_lat_f16_mode() >> 1) & 1);
  if (rc_lat_f16_mode() & 2) {
    RCMARL_LAUNCH(k_dz_pack_rowsum<true>, dim3(rts, S), dim3(256), 0, stream, dz, (unsigned char*)dzp, sums, sums_ld, sums_off, hid,
                  N * hid, B, ldb, dzp_rt, dzp_kt, kts);
  } else {
    RCMARL_LAUNCH(k_dz_pack_rowsum<false>, dim3(rts, S), dim3(256), 0, stream, dz, (unsigned char*)dzp, sums, sums_ld, sums_off, hid,
                  N * hid, B, ldb, dzp_rt, dzp_kt, kts);
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_layer1_forward_lattice(const void* kp, int kp_rt, int kp_kt, const void* wp, int wp_rt,
                                                int wp_kt, const float* theta, float* a1t, int S, int N, int B,
                                                int in_dim, int hid, int ldp, int ldb, void* stream) {
  if (!kp || !wp || !theta || !a1t || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || (ldp & 63) || (ldb & 63) || ldb < B ||
      ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid <= 0) return RCMARL_ERR_ARG;
  const int mtiles = rc_ceil_div(N * hid, 128), ntiles = rc_ceil_div(B, 256), ktiles = rc_ceil_div(in_dim, 32);
  if (wp_rt < mtiles || kp_rt < 2 * ntiles || wp_kt < ktiles || kp_kt < ktiles) return RCMARL_ERR_ARG;
  const unsigned nb = (unsigned)(S * mtiles * ntiles);
  const bool w8 = lat_w8(true), f16 = rc_lat_f16_mode() & 1;
  if (!rc_form_ok(kp, f16) || !rc_form_ok(wp, f16)) return RCMARL_ERR_ARG;       // written in the other operand form
#ifdef RC_LAT_PROTO
#define RC_LAT_PROTO_DISPATCH
#include "../../tools/prototypes/lattice_forward_shells.inc"
#undef RC_LAT_PROTO_DISPATCH
#endif
#define RC_FWD(W8, F16)                                                                                                       \
  launch_forward<W8, F16>(nb, stream, (const unsigned char*)wp, wp_rt, wp_kt, (const unsigned char*)kp, kp_rt, kp_kt, theta, \
                          a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, hid)
  return f16 ? (w8 ? RC_FWD(true, true) : RC_FWD(false, true)) : (w8 ? RC_FWD(true, false) : RC_FWD(false, false));
#undef RC_FWD
}

// The same GEMM for a WIDE network (hid a multiple of 128) whose next layers run on packed operands (dense_pk.hip): the epilogue
// writes the activations as two f16 pieces of 2^6 a1 in packed form -- a1_bk [S][N][bk_rt][hid/32][2][8 KiB] (rows = replay row,
// reduction = unit), a1_kb [S][N][hid/128][kb_kt][2][8 KiB] (rows = unit, reduction = replay row) -- and their sign bits
// s1 [S][N*hid][s1_ld] (bit b & 31 of word b >> 5), each optional, INSTEAD of the fp32 activations.  ovf_flag (optional): one int the
// kernel sets to 1 when an activation leaves the f16 range of the packed form (the pieces saturate).  Two-piece f16 form only
// (RCMARL_ERR_UNSUPPORTED otherwise).  Replaces model(x) of the first Dense layer, agents/resilient_CAC_agents.py:95-97,114,118.
RCMARL_EXPORT int rcmarl_layer1_forward_lattice_pk(const void* kp, int kp_rt, int kp_kt, const void* wp, int wp_rt, int wp_kt,
                                                   const float* theta, void* a1_bk, int bk_rt, void* a1_kb, int kb_kt,
                                                   unsigned* s1, int s1_ld, int* ovf_flag, int S, int N, int B, int in_dim, int hid,
                                                   int ldp, void* stream) {
  if (!kp || !wp || !theta || (!a1_bk && !a1_kb && !s1) || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || hid <= 0 || (ldp & 63) ||
      ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid & 127) return RCMARL_ERR_UNSUPPORTED;
  const int mtiles = N * hid / 128, ntiles = rc_ceil_div(B, 256), ktiles = rc_ceil_div(in_dim, 32);
  if (wp_rt < mtiles || kp_rt < 2 * ntiles || wp_kt < ktiles || kp_kt < ktiles) return RCMARL_ERR_ARG;
  if ((a1_bk && bk_rt < 2 * ntiles) || (a1_kb && kb_kt < 8 * ntiles) || (s1 && s1_ld < 8 * ntiles)) return RCMARL_ERR_ARG;
  if (!(rc_lat_f16_mode() & 1)) return RCMARL_ERR_UNSUPPORTED;
  if (!rc_form_ok(kp, true) || !rc_form_ok(wp, true)) return RCMARL_ERR_ARG;
  const unsigned nb = (unsigned)(S * mtiles * ntiles);
  LatPkOut pk;
  pk.a1_bk = (unsigned char*)a1_bk; pk.bk_rt = bk_rt; pk.a1_kb = (unsigned char*)a1_kb; pk.kb_kt = kb_kt; pk.s1 = s1; pk.s1_ld = s1_ld;
  pk.ovf = ovf_flag;
  return launch_forward<true, true, true>(nb, stream, (const unsigned char*)wp, wp_rt, wp_kt, (const unsigned char*)kp, kp_rt, kp_kt, theta,
                                          nullptr, S, N, B, in_dim, ldp, 64, mtiles, ntiles, hid, pk);
}

static int backward_sgd_lattice_impl(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt,
                                                     int dzp_kt, const float* alpha, float* theta, const int* mask,
                                                     int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                                     void* wp_out, int wp_rt, int wp_kt, void* stream) {
  if (!ktp || !dzp || !alpha || !theta || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || (ldp & 63) ||
      ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid <= 0) return RCMARL_ERR_ARG;
  const bool w8 = lat_w8(false);
  const int m128_env = lat_env_int("RCMARL_LAT_M128", 1);           // 1: 128-row tiles up to 128 inputs; 0: never; 2: always (measurement)
  const bool m128 = !w8 && ((in_dim <= 128 && m128_env != 0) || m128_env == 2);
  const int mtiles = m128 ? rc_ceil_div(in_dim, 128) : rc_ceil_div(in_dim, 256);
  const int ntiles = rc_ceil_div(N * hid, 128), ktiles = rc_ceil_div(B, 32);
  if (ktp_rt < (m128 ? mtiles : 2 * mtiles) || dzp_rt < ntiles || ktp_kt < ktiles || dzp_kt < ktiles) return RCMARL_ERR_ARG;
  if (wp_out && ((long)wp_rt * 128 < (long)ntiles * 128 || wp_kt < rc_ceil_div(in_dim, 32))) return RCMARL_ERR_ARG;
  const unsigned nb = (unsigned)(S * mtiles * ntiles);
  const int mode = rc_lat_f16_mode();
  if (!rc_form_ok(ktp, (mode >> 1) & 1) || !rc_form_ok(dzp, (mode >> 1) & 1)) return RCMARL_ERR_ARG;   // written in the other operand form
  rc_form_set(wp_out, mode & 1);
#define RC_BWD(W8, DZ16, WP16)                                                                                              \
  launch_backward<W8, DZ16, WP16>(nb, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt, (const unsigned char*)dzp, dzp_rt, \
                                  dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,                     \
                                  (unsigned char*)wp_out, wp_rt, wp_kt, hid)
  // at most 128 inputs: 128-row tiles (RCMARL_LAT_M128=0 keeps the 256-row tile)
  if (m128) {
#define RC_BWD128(DZ16, WP16)                                                                                                  \
  launch_backward<false, DZ16, WP16, true>(nb, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt, (const unsigned char*)dzp,  \
                                           dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,       \
                                           (unsigned char*)wp_out, wp_rt, wp_kt, hid)
    switch (mode) {
      case 0: return RC_BWD128(false, false);
      case 1: return RC_BWD128(false, true);
      case 2: return RC_BWD128(true, false);
      default: return RC_BWD128(true, true);
    }
#undef RC_BWD128
  }
  switch (mode * 2 + (w8 ? 1 : 0)) {
    case 0: return RC_BWD(false, false, false);
    case 1: return RC_BWD(true, false, false);
    case 2: return RC_BWD(false, false, true);
    case 3: return RC_BWD(true, false, true);
    case 4: return RC_BWD(false, true, false);
    case 5: return RC_BWD(true, true, false);
    case 6: return RC_BWD(false, true, true);
    default: return RC_BWD(true, true, true);
  }
#undef RC_BWD
}

RCMARL_EXPORT int rcmarl_layer1_backward_sgd_lattice(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt,
                                                     int dzp_kt, const float* alpha, float* theta, const int* mask,
                                                     int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                                     void* wp_out, int wp_rt, int wp_kt, void* stream) {
  return backward_sgd_lattice_impl(ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, hid, ldp, lr, wp_out,
                                   wp_rt, wp_kt, stream);
}
