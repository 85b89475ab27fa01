// Dense layers of a WIDE network (hid a multiple of 128; BASELINE configs[4]: the 512-unit critic) on PRE-SPLIT packed operands.
//
// wide_kernels.hip runs the per-agent GEMMs of layer 2 on the 16-bit matrix core too, but splits its fp32 operands into f16 pieces
// while it stages them (global fp32 -> registers -> pieces -> LDS): the matrix pipe sat 28-37 % busy behind that loader
// (profiles/r05_pmc_kbench_wide_counters.json).  Here every operand reaches its GEMM already as packed f16 pieces ("PK" blocks of
// rcmarl_lattice.h), written in that form by the epilogue of the kernel that PRODUCES it, and the GEMMs run the LDS-DMA k-loop of the
// layer-1 lattice kernels (rcmarl_lat_mainloop.h).  One full-batch SGD step of fit() (agents/resilient_CAC_agents.py:118) on a
// (in -> J -> J -> 1) network, J = hid, per (seed, agent):
//
//   rcmarl_layer1_forward_lattice_pk  a1 = lrelu(x W1 + b1)        -> a1_bk [rows b][red unit] + a1_kb [rows unit][red b] (2 f16 pieces
//                                     (lattice_gemm.hip)              of 2^6 a1) + s1 (sign bits of a1)
//   rcmarl_pk_pack_w2                 W2, W3                       -> w2t [rows j][red k] (2^10 W2), w2w3 [rows k][red j]
//                                                                     (2^10 W2[k][j] W3[j]), rs[k] = sum_j W2[k][j] W3[j]
//   rcmarl_pk_forward2                z2 = a1 W2 + b2, a2 = lrelu  -> mask_bj [rows b][red j] (f16 1 / 0), mask_jb [rows j][red b]
//                                                                     (0xffff / 0), vpart[tile][b] = sum_{j in tile} a2[j][b] W3[j]
//                                                                     (and, for the estimate consensus, fp32 a2 feature-major)
//   rcmarl_pk_head                    v = sum vpart + b3           -> dz3 = 2 (v - y) / B, its pieces dzv = {dz3, 0.1 dz3} x {h, l}, loss
//   rcmarl_pk_backward_data           dz1 = lrelu'(a1) W2 dz2      -> dzp (the lattice backward GEMM's operand, 2 pieces of 2^8 dz1)
//                                                                     + per-tile sums of dz1 (gb1)
//   rcmarl_pk_backward_w2             W2 -= lr a1^T dz2            -> theta; per-tile parts of gW3, q[j] = sum_b lrelu'(z2) dz3
//   rcmarl_pk_small_sgd               b1, b2, W3, b3 -= lr grad
//
// What makes this possible without ever storing a2 or dz2: LeakyReLU' takes two values, so with m = [z2 > 0] and g = 0.1 + 0.9 m
//   dz2[j][b]  = W3[j] dz3[b] g[j][b]
//   dz1[k][b]  = lrelu'(a1) dz3[b] (0.1 rs[k] + 0.9 sum_j (W2[k][j] W3[j]) m[j][b])        -- the mask is ONE exact f16 piece: 2 passes
//   gW2[k][j]  = W3[j] G[k][j],   G[k][j] = sum_b a1[k][b] (g[j][b] dz3[b])                -- the second operand is formed in the k-loop:
//                                                                                             bitwise select between the pieces of dz3 and
//                                                                                             of 0.1 dz3 under the 0xffff / 0 mask
//   gW3[j]     = sum_b a2[j][b] dz3[b] = sum_k W2[k][j] G[k][j] + b2[j] q[j]               -- a2 = g (a1 W2 + b2), so the head's gradient
//                                                                                             is a column sum in the W2 epilogue
//   gb2[j]     = W3[j] q[j]
// Operands: activations 2^6, weights 2^10, dz 2^8 (rcmarl_lattice.h); products l*l are dropped where both sides have two pieces
// (2^-22 of the product), conversions saturate.  Results differ from the fp32 chain of wide_kernels.hip by rounding order only
// (tests/wide_checks.py holds both to the same oracle bars).
#include "rcmarl_lattice.h"
#include "rcmarl_lat_mainloop.h"
#include <type_traits>

namespace {

// Block tiles of the three GEMMs: 256 x 256 with eight wavefronts of 128 x 64 when hid is a multiple of 256 (one workgroup per CU;
// half the operand bytes per matrix instruction of the 128 x 128 form: at 128 x 128 the launches ran at 7-11 TB/s of LDS-DMA
// traffic out of the L2 with the matrix pipe half idle, profiles/r06a_kbench_pk.txt), else 128 x 128 with four of 64 x 64.
struct PkSmall { static constexpr int MT = 2, NT = 2, WM = 2, WN = 2, BM = 128, BN = 128, THREADS = 256; };
struct PkBig { static constexpr int MT = 4, NT = 2, WM = 2, WN = 4, BM = 256, BN = 256, THREADS = 512; };
__host__ __device__ static inline int pk_tile(int hid) { return (hid & 255) == 0 ? 256 : 128; }
constexpr float PK_LEAK_COMP = 0.9f;                      // g = RC_LEAK + PK_LEAK_COMP * [z > 0]

// workgroup id -> (z = seed * N + agent, tile q): all tiles of an agent on ONE XCD (workgroups go round the eight XCDs by id)
__device__ __forceinline__ void pk_decode(int g, int per, int nz, int& z, int& q) {
  if ((nz & 7) == 0) {
    const int qq = g >> 3;
    z = (g & 7) + 8 * (qq / per);
    q = qq % per;
  } else {
    z = g / per;
    q = g - z * per;
  }
}

__device__ __forceinline__ void pk_store8(unsigned char* dst, const float (&w)[8], float scale) {
  uint4 vh, vl;
  rc_split2h_pair(w[0] * scale, w[1] * scale, vh.x, vl.x);
  rc_split2h_pair(w[2] * scale, w[3] * scale, vh.y, vl.y);
  rc_split2h_pair(w[4] * scale, w[5] * scale, vh.z, vl.z);
  rc_split2h_pair(w[6] * scale, w[7] * scale, vh.w, vl.w);
  st_u4(dst, vh);
  st_u4(dst + RC_PK_BLOCK, vl);
}

// ---------------------------------------------------------------------------------------------
// W2 (hid x hid, row-major [k][j]) and W3 of every (seed, agent) -> the two packed weight operands + rs.
// grid (hid / 128, Z): workgroup = 128 rows of BOTH outputs, all k-tiles (a row sum is formed by one thread quad in a fixed order).
__global__ __launch_bounds__(256) void k_pk_pack_w2(const float* __restrict__ theta, int ldp, int o_W2, int o_W3, int hid,
                                                    unsigned char* __restrict__ w2t, unsigned char* __restrict__ w2w3,
                                                    float* __restrict__ rs, int* __restrict__ ovf) {
  const int z = blockIdx.y, rt = blockIdx.x, t = threadIdx.x, c4 = t & 3;
  const int JT = hid >> 7, JK = hid >> 5;
  rc_f16_saturate();
  const float* __restrict__ W2 = theta + (long)z * ldp + o_W2;
  const float* __restrict__ W3 = theta + (long)z * ldp + o_W3;
  float acc[2] = {0.f, 0.f};
  float amax = 0.f;
  for (int kt = 0; kt < JK; ++kt) {
    unsigned char* blk_t = w2t + (((long)z * JT + rt) * JK + kt) * (2 * RC_PK_BLOCK);
    unsigned char* blk_w = w2w3 + (((long)z * JT + rt) * JK + kt) * (2 * RC_PK_BLOCK);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (t >> 2) + 64 * q, row = rt * 128 + r, c0 = kt * 32 + 8 * c4;
      // w2w3: row = k, reduction = j (contiguous in memory)
      const float4 lo = *reinterpret_cast<const float4*>(W2 + (long)row * hid + c0), hi = *reinterpret_cast<const float4*>(W2 + (long)row * hid + c0 + 4);
      const float4 w3a = *reinterpret_cast<const float4*>(W3 + c0), w3b = *reinterpret_cast<const float4*>(W3 + c0 + 4);
      const float w[8] = {lo.x * w3a.x, lo.y * w3a.y, lo.z * w3a.z, lo.w * w3a.w, hi.x * w3b.x, hi.y * w3b.y, hi.z * w3b.z, hi.w * w3b.w};
      acc[q] += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      amax = rc_amax3(rc_amax3(rc_amax3(rc_amax3(amax, w[0], w[1]), w[2], w[3]), w[4], w[5]), w[6], w[7]);
      pk_store8(blk_w + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), w, RC_F16_W_SCALE);
      // w2t: row = j, reduction = k (stride hid in memory; the 64 rows of a pass are consecutive j: coalesced per k)
      float u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = W2[(long)(c0 + e) * hid + row];
      amax = rc_amax3(rc_amax3(rc_amax3(rc_amax3(amax, u[0], u[1]), u[2], u[3]), u[4], u[5]), u[6], u[7]);
      pk_store8(blk_t + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4), u, RC_F16_W_SCALE);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float v = acc[q];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (c4 == 0) rs[(long)z * hid + rt * 128 + (t >> 2) + 64 * q] = v;
  }
  if (ovf != nullptr && !(amax * RC_F16_W_SCALE <= 65000.f)) *ovf = 1;     // (|W2| or |W2 W3| > 63: the pieces saturate; NaN counts)
}

// ---------------------------------------------------------------------------------------------
// layer 2 forward: tile [m = unit j][n = replay row b] = sum_k W2[k][j] a1[k][b]
struct PkFwdArgs {
  const unsigned char* w2t;                 // [Z][JT][JK][2][8 KiB]
  const unsigned char* a1bk; int bk_rt;     // [Z][bk_rt][JK][2][8 KiB]
  const float* theta; int ldp, o_b2, o_W3;
  float* a2; int ldb;                       // fp32 feature-major [Z * hid][ldb], or NULL
  unsigned char* mask_bj; int mbj_rt;       // [Z][mbj_rt][JK][8 KiB], f16 1.0 / 0, or NULL
  unsigned char* mask_jb; int mjb_kt;       // [Z][JT][mjb_kt][8 KiB], 0xffff / 0, or NULL
  float* vpart;                             // [Z][JT][ldb], or NULL
  float* npart;                             // [Z][JT][ldb], or NULL: sum_{j in tile} a2[j][b]^2 (|phi|^2 of the projection step)
  int Z, B, hid, ntb;                       // ntb = ceil(B / 128)
};

template <class T>
__global__ RC_LAT_OCC(T::THREADS, 2) void k_pk_forward2(const PkFwdArgs a) {
  constexpr int MT = T::MT, NT = T::NT, WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN;
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int JT = a.hid >> 7, JK = a.hid >> 5, MTL = a.hid / BM;
  int z, q;
  pk_decode(blockIdx.x, MTL * a.ntb, a.Z, z, q);
  const int bm = q % MTL, bn = q / MTL;                   // m fastest: the workgroups that share an a1 panel run side by side
  LatOperands op;
  op.a = a.w2t + (long)z * JT * JK * (2 * RC_PK_BLOCK); op.a_kt = JK; op.art0 = bm * (BM / 128);
  op.b = a.a1bk + (long)z * a.bk_rt * JK * (2 * RC_PK_BLOCK); op.b_kt = JK; op.brt0 = bn * (BN / 128);
  rc_f32x16 acc[MT][NT];
  lat_mainloop<2, 2, MT, NT, WM, WN, false, true, true>(op, JK, lds, acc);
  __syncthreads();
  float* sb = reinterpret_cast<float*>(lds);              // b2, W3 of the tile's units; the row halves' parts of v
  float* sw3 = sb + BM;
  float* sv = sw3 + BM;
  float* sn = sv + WM * BN;
  const float* __restrict__ th = a.theta + (long)z * a.ldp;
  if (threadIdx.x < BM) {
    sb[threadIdx.x] = th[a.o_b2 + bm * BM + threadIdx.x];
    sw3[threadIdx.x] = th[a.o_W3 + bm * BM + threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wm = wave / WN, wn = wave % WN;
  unsigned char* scratch = lds + 8192 + wave * 2048;
  constexpr float UNSCALE = RC_F16_W_UNSCALE * RC_F16_ACT_UNSCALE;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = bn * BN + wn * 32 * NT + 32 * nt, n = n0 + l31;
    float vp = 0.f, np2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int j0 = wm * 32 * MT + 32 * mt;              // first unit of this 32 x 32 block inside the tile
      unsigned pc[1][4][2];
      unsigned short m16[1][16];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 b4 = *reinterpret_cast<const float4*>(sb + j0 + 8 * qq + 4 * half);
        const float4 w4 = *reinterpret_cast<const float4*>(sw3 + j0 + 8 * qq + 4 * half);
        const float bq[4] = {b4.x, b4.y, b4.z, b4.w}, wq[4] = {w4.x, w4.y, w4.z, w4.w};
        bool pos[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qq + e;
          const float zz = fmaf(acc[mt][nt][r], UNSCALE, bq[e]);
          const float o = fmaxf(zz, RC_LEAK * zz);
          pos[e] = o > 0.f;
          vp = fmaf(o, wq[e], vp);
          np2 = fmaf(o, o, np2);
          if (a.a2 != nullptr && n < a.B)
            RC_NT_STORE(a.a2 + ((long)z * a.hid + bm * BM + j0 + 8 * qq + 4 * half + e) * a.ldb + n, o);
          m16[0][r] = pos[e] ? (unsigned short)0xffffu : (unsigned short)0u;
        }
        pc[0][qq][0] = (pos[0] ? 0x3c00u : 0u) | (pos[1] ? 0x3c000000u : 0u);
        pc[0][qq][1] = (pos[2] ? 0x3c00u : 0u) | (pos[3] ? 0x3c000000u : 0u);
      }
      if (a.mask_bj != nullptr) {
        unsigned char* rowp = a.mask_bj + (((long)z * a.mbj_rt + (n >> 7)) * JK + ((bm * BM + j0) >> 5)) * RC_PK_BLOCK + (n & 127) * 64;
        pk_emit_rows_from_lanes<1>(pc, rowp, (n >> 2) & 3, (n >> 7) < a.mbj_rt);
      }
      if (a.mask_jb != nullptr && (n0 >> 5) < a.mjb_kt) {
        const int jg = bm * BM + j0;                       // the block's first unit inside the agent
        unsigned char* blk = a.mask_jb + (((long)z * JT + (jg >> 7)) * a.mjb_kt + (n0 >> 5)) * RC_PK_BLOCK;
        pk_emit_rows_from_regs<1>(m16, scratch, blk, jg & 127);
      }
    }
    if (a.vpart != nullptr) {
      vp += __shfl_xor(vp, 32);
      if (half == 0) sv[wm * BN + wn * 32 * NT + 32 * nt + l31] = vp;
    }
    if (a.npart != nullptr) {
      np2 += __shfl_xor(np2, 32);
      if (half == 0) sn[wm * BN + wn * 32 * NT + 32 * nt + l31] = np2;
    }
  }
  if (a.npart != nullptr) {
    __syncthreads();
    if (threadIdx.x < BN) {
      const int n = bn * BN + threadIdx.x;
      float v = sn[threadIdx.x];
#pragma unroll
      for (int w = 1; w < WM; ++w) v += sn[w * BN + threadIdx.x];
      if (n < a.ldb) a.npart[((long)z * MTL + bm) * a.ldb + n] = v;
    }
  }
  if (a.vpart != nullptr) {
    __syncthreads();
    if (threadIdx.x < BN) {
      const int n = bn * BN + threadIdx.x;
      float v = sv[threadIdx.x];
#pragma unroll
      for (int w = 1; w < WM; ++w) v += sv[w * BN + threadIdx.x];
      if (n < a.ldb) a.vpart[((long)z * MTL + bm) * a.ldb + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// the head: v[b] = sum_t vpart[t][b] + b3.  mode 0: out = v; 1: out = aux + gamma v (TD target); 2 (fit): out = dz3 = 2 (v - aux) / B,
// dzv[z][4][Bp] = f16 pieces {h(2^8 dz3), l(2^8 dz3), h(2^8 0.1 dz3), l(2^8 0.1 dz3)} (zero from B on: the weight-gradient GEMM's
// reduction runs over whole 32-row tiles), loss parts per 256 rows.
__global__ __launch_bounds__(256) void k_pk_head(const float* __restrict__ vpart, const float* __restrict__ theta, int ldp, int o_b3, int JT,
                                                 const float* __restrict__ aux, float gamma, int mode, float* __restrict__ out,
                                                 unsigned short* __restrict__ dzv, int Bp, float* __restrict__ losspart, int nchunk,
                                                 int B, int ldb) {
  __shared__ float red[4];
  const int z = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
  const bool valid = b < B;
  float v = 0.f;
  if (valid) {
    for (int t = 0; t < JT; ++t) v += vpart[((long)z * JT + t) * ldb + b];
    v += theta[(long)z * ldp + o_b3];
  }
  const long o = (long)z * ldb + b;
  if (mode != 2) {
    if (valid) out[o] = mode == 1 ? aux[o] + gamma * v : v;
    return;
  }
  rc_f16_saturate();
  const float diff = valid ? v - aux[o] : 0.f;
  const float dz3 = (2.0f * diff) / (float)B;
  if (valid) out[o] = dz3;
  if (b < Bp) {
    unsigned h, l;
    rc_split2h_pair(dz3 * RC_F16_DZ_SCALE, (RC_LEAK * dz3) * RC_F16_DZ_SCALE, h, l);
    unsigned short* d = dzv + (long)z * 4 * Bp + b;
    d[0] = (unsigned short)(h & 0xffffu);
    d[Bp] = (unsigned short)(l & 0xffffu);
    d[2 * (long)Bp] = (unsigned short)(h >> 16);
    d[3 * (long)Bp] = (unsigned short)(l >> 16);
  }
  const float sq = rc_wave_sum(diff * diff);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0 && (int)blockIdx.x < nchunk) losspart[(long)z * nchunk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
// backward-data, operand roles swapped: tile [m = replay row b][n = unit k] = sum_j m[j][b] (W2[k][j] W3[j]) -- a lane then holds four
// consecutive replay rows of ONE unit per register group, which is the row layout of the lattice backward's dz operand.
struct PkBdArgs {
  const unsigned char* mask_bj; int mbj_rt;
  const unsigned char* w2w3;                // [Z][JT][JK][2][8 KiB]
  const float* rs;                          // [Z][hid]
  const unsigned* s1; int s1_ld;            // [Z * hid][s1_ld]
  const float* dz3; int ldb;                // [Z][ldb]
  unsigned char* dzp; int dzp_rt, dzp_kt;   // per seed [dzp_rt][dzp_kt][2][8 KiB], rows = agent * hid + unit
  float* gb1part;                           // [Z][ntb][hid]
  int* ovf;                                 // set when |2^8 dz1| leaves the f16 range
  int Z, N, B, hid, ntb;
};

template <class T>
__global__ RC_LAT_OCC(T::THREADS, 2) void k_pk_backward_data(const PkBdArgs a) {
  constexpr int MT = T::MT, NT = T::NT, WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN;
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int JT = a.hid >> 7, JK = a.hid >> 5, NTL = a.hid / BN;
  int z, q;
  pk_decode(blockIdx.x, NTL * a.ntb, a.Z, z, q);
  const int bn = q % NTL, bm = q / NTL;                   // n (unit tile) fastest: neighbours share the mask panel
  LatOperands op;
  op.a = a.mask_bj + (long)z * a.mbj_rt * JK * RC_PK_BLOCK; op.a_kt = JK; op.art0 = bm * (BM / 128);
  op.b = a.w2w3 + (long)z * JT * JK * (2 * RC_PK_BLOCK); op.b_kt = JK; op.brt0 = bn * (BN / 128);
  rc_f32x16 acc[MT][NT];
  lat_mainloop<1, 2, MT, NT, WM, WN, false, true>(op, JK, lds, acc);
  __syncthreads();
  float* sdz = reinterpret_cast<float*>(lds);             // dz3 of the tile's replay rows; the row halves' parts of gb1
  float* sg = sdz + BM;
  if (threadIdx.x < BM) {
    const int b = bm * BM + threadIdx.x;
    sdz[threadIdx.x] = b < a.B ? a.dz3[(long)z * a.ldb + b] : 0.f;
  }
  __syncthreads();
  rc_f16_saturate();
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wm = wave / WN, wn = wave % WN;
  const int s = z / a.N, agent = z - s * a.N;
  unsigned char* dzp_s = a.dzp + (long)s * a.dzp_rt * a.dzp_kt * (2 * RC_PK_BLOCK);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int kc = bn * BN + wn * 32 * NT + 32 * nt + l31; // the lane's unit
    const float t01 = RC_LEAK * a.rs[(long)z * a.hid + kc];
    const int prow = agent * a.hid + kc;                  // its row in the packed dz operand
    float gsum = 0.f, amax = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int bl = wm * 32 * MT + 32 * mt, b0 = bm * BM + bl;  // first replay row of this 32 x 32 block (a multiple of 32)
      const unsigned word = a.s1[((long)z * a.hid + kc) * a.s1_ld + (b0 >> 5)];
      unsigned pc[2][4][2];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 d4 = *reinterpret_cast<const float4*>(sdz + bl + 8 * qq + 4 * half);
        const float dq[4] = {d4.x, d4.y, d4.z, d4.w};
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qq + e;
          const float g1 = ((word >> (8 * qq + 4 * half + e)) & 1u) ? 1.f : RC_LEAK;
          const float tt = fmaf(PK_LEAK_COMP, acc[mt][nt][r] * RC_F16_W_UNSCALE, t01);
          const float dv = (dq[e] * tt) * g1;
          gsum += dv;
          x[e] = dv * RC_F16_DZ_SCALE;
        }
        amax = rc_amax3(rc_amax3(amax, x[0], x[1]), x[2], x[3]);
        rc_split2h_pair(x[0], x[1], pc[0][qq][0], pc[1][qq][0]);
        rc_split2h_pair(x[2], x[3], pc[0][qq][1], pc[1][qq][1]);
      }
      unsigned char* rowp = dzp_s + ((long)(prow >> 7) * a.dzp_kt + (b0 >> 5)) * (2 * RC_PK_BLOCK) + (prow & 127) * 64;
      pk_emit_rows_from_lanes<2>(pc, rowp, (prow >> 2) & 3, (b0 >> 5) < a.dzp_kt);
    }
    gsum += __shfl_xor(gsum, 32);
    if (half == 0) sg[wm * BN + wn * 32 * NT + 32 * nt + l31] = gsum;
    if (a.ovf != nullptr && !(amax <= 65000.f)) *a.ovf = 1;
  }
  __syncthreads();
  if (threadIdx.x < BN) {
    float v = sg[threadIdx.x];
#pragma unroll
    for (int w = 1; w < WM; ++w) v += sg[w * BN + threadIdx.x];
    a.gb1part[((long)z * a.ntb + bm) * a.hid + bn * BN + threadIdx.x] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient + SGD of W2: tile [m = unit k][n = unit j] = sum_b a1[k][b] (g[j][b] dz3[b]).  The second operand does not exist
// in memory: the LDS stage holds the 0xffff / 0 mask (one piece), and per k16 step the fragment's two pieces are SELECTED bitwise
// between the pieces of dz3 and of 0.1 dz3 (eight consecutive replay rows, the same for every row of the fragment: four 16-byte loads
// a lane, one k-tile ahead).  q[j] = sum_b g dz3 falls out of the same fragments (v_dot2 against ones) in the wavefronts of the
// first row tile.
struct PkBwArgs {
  const unsigned char* a1kb; int kb_kt;     // [Z][JT][kb_kt][2][8 KiB]
  const unsigned char* mask_jb; int mjb_kt; // [Z][JT][mjb_kt][8 KiB]
  const unsigned short* dzv; int Bp;        // [Z][4][Bp]
  float* theta; int ldp, o_W2, o_W3;
  const int* mask; float lr;
  float* gw3part;                           // [Z][JT][hid]
  float* qout;                              // [Z][hid]
  int Z, N, B, hid;
};

__device__ __forceinline__ uint4 pk_bfi(const uint4& m, const uint4& x, const uint4& y) {
  uint4 r;
  r.x = (m.x & x.x) | (~m.x & y.x); r.y = (m.y & x.y) | (~m.y & y.y);
  r.z = (m.z & x.z) | (~m.z & y.z); r.w = (m.w & x.w) | (~m.w & y.w);
  return r;
}
// sum of the eight f16 values of a fragment, fp32 accumulate
__device__ __forceinline__ float pk_sum8(const uint4& f, float acc) {
#ifdef RCMARL_EMU
  const unsigned w[4] = {f.x, f.y, f.z, f.w};
  for (int i = 0; i < 4; ++i) acc += rc_f16_to_f32(w[i] & 0xffffu) + rc_f16_to_f32(w[i] >> 16);
  return acc;
#else
  const rc_h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rc_h2, f.x), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rc_h2, f.y), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rc_h2, f.z), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rc_h2, f.w), one, acc, false);
  return acc;
#endif
}

template <class T>
__global__ RC_LAT_OCC(T::THREADS, 2) void k_pk_backward_w2(const PkBwArgs a) {
  constexpr int PA = 2, PB = 1, MT = T::MT, NT = T::NT, WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN;
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int JT = a.hid >> 7, MTL = a.hid / BM, NTL = a.hid / BN;
  int z, q;
  pk_decode(blockIdx.x, MTL * NTL, a.Z, z, q);
  const int agent = z % a.N;
  if (a.mask != nullptr && !a.mask[agent]) return;        // workgroup-uniform
  const int bm = q % MTL, bn = q / MTL;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wm = wave / WN, wn = wave % WN;
  const bool do_q = bm == 0 && wm == 0;                   // wave-uniform
  rc_f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
  float qs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) qs[nt] = 0.f;
  {
    const unsigned char* pa = a.a1kb + ((long)z * JT + bm * (BM / 128)) * a.kb_kt * (PA * RC_PK_BLOCK);
    const unsigned char* pb = a.mask_jb + ((long)z * JT + bn * (BN / 128)) * a.mjb_kt * (PB * RC_PK_BLOCK);
    const unsigned char* gsrc[C::GLDS];
    int gstep[C::GLDS];
#pragma unroll
    for (int i = 0; i < C::GLDS; ++i) {
      const int qi = wave + C::NWV * i;
      if (qi < C::A_KB) {
        const int seg = qi / (PA * 8), off = qi - seg * (PA * 8);
        gsrc[i] = pa + (long)seg * a.kb_kt * (PA * RC_PK_BLOCK) + off * 1024; gstep[i] = PA * RC_PK_BLOCK;
      } else {
        const int q2 = qi - C::A_KB, seg = q2 / (PB * 8), off = q2 - seg * (PB * 8);
        gsrc[i] = pb + (long)seg * a.mjb_kt * (PB * RC_PK_BLOCK) + off * 1024; gstep[i] = PB * RC_PK_BLOCK;
      }
    }
    const unsigned lane16 = lane * 16;
    const rc_lds_t lds0 = rc_lds_addr(lds) + wave * 1024;
    auto stage = [&](int buf, int t) {
      const rc_lds_t dst = lds0 + buf * C::STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < C::GLDS; ++i) RC_GLDS16S(gsrc[i] + (long)t * gstep[i], lane16, dst + i * (C::NWV * 1024));
    };
    const unsigned short* dzb = a.dzv + (long)z * 4 * a.Bp + 8 * half;
    uint4 dn[2][4], dv[2][4];
    auto load_dz = [&](int t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 4; ++p) dn[ks][p] = *reinterpret_cast<const uint4*>(dzb + (long)p * a.Bp + 32 * t + 16 * ks);
    };
    const int sw = (l31 >> 2) & 3;
    const int co0 = ((0 + half) ^ sw) << 4, co1 = ((2 + half) ^ sw) << 4;
    int offA[MT], offB[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wm * 32 * MT + 32 * mt + l31;
      offA[mt] = (row >> 7) * PA * RC_PK_BLOCK + (row & 127) * 64;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = wn * 32 * NT + 32 * nt + l31;
      offB[nt] = C::A_KB * 1024 + (row >> 7) * PB * RC_PK_BLOCK + (row & 127) * 64;
    }
    const int n_ktiles = (a.B + 31) >> 5;
    stage(0, 0);
    load_dz(0);
    for (int t = 0; t < n_ktiles; ++t) {
      const int cur = t & 1;
      RC_WAIT_VMEM();
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 4; ++p) dv[ks][p] = dn[ks][p];
      if (t + 1 < n_ktiles) {
        stage(cur ^ 1, t + 1);
        load_dz(t + 1);
      }
      const unsigned char* st = lds + cur * C::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int co = ks == 0 ? co0 : co1;
        uint4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          ah[mt] = ld_u4(st + offA[mt] + co);
          al[mt] = ld_u4(st + offA[mt] + RC_PK_BLOCK + co);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const uint4 m = ld_u4(st + offB[nt] + co);
          bh[nt] = pk_bfi(m, dv[ks][0], dv[ks][2]);
          bl[nt] = pk_bfi(m, dv[ks][1], dv[ks][3]);
          if (do_q) qs[nt] = pk_sum8(bl[nt], pk_sum8(bh[nt], qs[nt]));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = rc_mfma_f16(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = rc_mfma_f16(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = rc_mfma_f16(ah[mt], bh[nt], acc[mt][nt]);
      }
    }
  }
  __syncthreads();
  float* sw3 = reinterpret_cast<float*>(lds);             // W3 of the tile's columns; the row halves' parts of gW3
  float* sgw = sw3 + BN;
  float* __restrict__ th = a.theta + (long)z * a.ldp;
  if (threadIdx.x < BN) sw3[threadIdx.x] = th[a.o_W3 + bn * BN + threadIdx.x];
  __syncthreads();
  constexpr float UNSCALE = RC_F16_ACT_UNSCALE * RC_F16_DZ_UNSCALE;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int jl = wn * 32 * NT + 32 * nt + l31, j = bn * BN + jl;
    const float w3 = sw3[jl];
    float gsum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float* __restrict__ wp = th + a.o_W2 + (long)(bm * BM + wm * 32 * MT + 32 * mt + 4 * half) * a.hid + j;
      float wold[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) wold[r] = wp[(long)(8 * (r >> 2) + (r & 3)) * a.hid];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float G = acc[mt][nt][r] * UNSCALE;
        gsum = fmaf(wold[r], G, gsum);
        wp[(long)(8 * (r >> 2) + (r & 3)) * a.hid] = wold[r] - a.lr * (w3 * G);
      }
    }
    gsum += __shfl_xor(gsum, 32);
    if (half == 0) sgw[wm * BN + jl] = gsum;
    if (do_q) {
      const float qv = qs[nt] + __shfl_xor(qs[nt], 32);
      if (half == 0) a.qout[(long)z * a.hid + j] = qv * RC_F16_DZ_UNSCALE;
    }
  }
  __syncthreads();
  if (threadIdx.x < BN) {
    float v = sgw[threadIdx.x];
#pragma unroll
    for (int w = 1; w < WM; ++w) v += sgw[w * BN + threadIdx.x];
    a.gw3part[((long)z * MTL + bm) * a.hid + bn * BN + threadIdx.x] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// b1, b2, W3, b3 -= lr * grad (masked agents only); loss_out[z] = sum(diff^2) / B
__global__ __launch_bounds__(256) void k_pk_small_sgd(const float* __restrict__ gw3part, int JT, const float* __restrict__ qv,
                                                      const float* __restrict__ gb1part, int ntb, const float* __restrict__ dz3,
                                                      int ldb, const float* __restrict__ losspart, int nchunk,
                                                      float* __restrict__ theta, int ldp, int o_b1, int o_b2, int o_W3, int o_b3,
                                                      const int* __restrict__ mask, float* __restrict__ loss_out, int N, int B, int hid,
                                                      float lr) {
  __shared__ float red[256];
  const int z = blockIdx.x, t = threadIdx.x;
  if (loss_out != nullptr && t == 0) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += losspart[(long)z * nchunk + c];
    loss_out[z] = sum / (float)B;
  }
  if (mask != nullptr && !mask[z % N]) return;
  float g3 = 0.f;
  for (int b = t; b < B; b += 256) g3 += dz3[(long)z * ldb + b];
  red[t] = g3;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  float* __restrict__ th = theta + (long)z * ldp;
  for (int e = t; e < hid; e += 256) {
    const float qq = qv[(long)z * hid + e], w3 = th[o_W3 + e], b2 = th[o_b2 + e];
    float gw3 = 0.f, gb1 = 0.f;
    for (int tt = 0; tt < JT; ++tt) gw3 += gw3part[((long)z * JT + tt) * hid + e];
    gw3 = fmaf(b2, qq, gw3);
    for (int tt = 0; tt < ntb; ++tt) gb1 += gb1part[((long)z * ntb + tt) * hid + e];
    th[o_W3 + e] = w3 - lr * gw3;
    th[o_b2 + e] = b2 - lr * (w3 * qq);
    th[o_b1 + e] = th[o_b1 + e] - lr * gb1;
  }
  if (t == 0) th[o_b3] = th[o_b3] - lr * red[0];
}

static inline bool pk_dims_ok(int S, int N, int B, int hid, int ldp) { return S > 0 && N > 0 && B > 0 && hid > 0 && ldp > 0; }
template <class T> static inline size_t pk_smem(int pa, int pb) { return (size_t)2 * (pa * (T::BM / 128) + pb * (T::BN / 128)) * RC_PK_BLOCK; }

template <class T> static int pk_launch_forward2(const PkFwdArgs& a, void* stream) {
  const size_t smem = pk_smem<T>(2, 2);
  static const bool ok = rc_want_lds(k_pk_forward2<T>, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_pk_forward2<T>), dim3((unsigned)(a.Z * (a.hid / T::BM) * a.ntb)), dim3(T::THREADS), smem, stream, a);
  return rcmarl_check_launch();
}
template <class T> static int pk_launch_backward_data(const PkBdArgs& a, void* stream) {
  const size_t smem = pk_smem<T>(1, 2);
  static const bool ok = rc_want_lds(k_pk_backward_data<T>, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_pk_backward_data<T>), dim3((unsigned)(a.Z * (a.hid / T::BN) * a.ntb)), dim3(T::THREADS), smem, stream, a);
  return rcmarl_check_launch();
}
template <class T> static int pk_launch_backward_w2(const PkBwArgs& a, void* stream) {
  const size_t smem = pk_smem<T>(2, 1);
  static const bool ok = rc_want_lds(k_pk_backward_w2<T>, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH((k_pk_backward_w2<T>), dim3((unsigned)(a.Z * (a.hid / T::BM) * (a.hid / T::BN))), dim3(T::THREADS), smem, stream, a);
  return rcmarl_check_launch();
}

}  // namespace

// Sizes a caller allocates (bytes / elements), for B replay rows and `hid` units (B padded to 256 rows):
//   a1_bk, a1_kb: S*N * (Bp/128) * (hid/32) * 2 * 8192 bytes each       mask_bj, mask_jb: half of that
//   w2t, w2w3:    S*N * (hid/128) * (hid/32) * 2 * 8192 bytes each       s1: S*N*hid * (Bp/32) uint32
//   dzv: S*N * 4 * Bp uint16     vpart: S*N * (hid/128) * ldb fp32      gw3part: S*N * (hid/128) * hid    gb1part: S*N * ceil(B/128) * hid
// (the part buffers are used up to hid/T resp. ceil(B/T) entries, T = the block tile side: 256 when hid % 256 == 0, else 128)
RCMARL_EXPORT int rcmarl_pk_parts(int hid) { return hid > 0 && (hid & 127) == 0 ? hid / pk_tile(hid) : 0; }
RCMARL_EXPORT int rcmarl_pk_supported(int hid) { return hid > 0 && (hid & 127) == 0 && (rc_lat_f16_mode() & 3) == 3; }

// theta[s][n] -> w2t, w2w3, rs   (o_W2 = in_dim*hid + hid, o_W3 = o_W2 + hid*hid + hid: the Keras row of rcmarl_common.h)
RCMARL_EXPORT int rcmarl_pk_pack_w2(const float* theta, void* w2t, void* w2w3, float* rs, int* ovf_flag, int S, int N, int in_dim, int hid,
                                    int ldp, void* stream) {
  if (!theta || !w2t || !w2w3 || !rs || !pk_dims_ok(S, N, 1, hid, ldp) || in_dim <= 0 || (ldp & 3)) return RCMARL_ERR_ARG;
  if ((hid & 127) || ((in_dim * hid + hid) & 3)) return RCMARL_ERR_UNSUPPORTED;
  const NetGeom g = make_geom(in_dim, hid, 1);
  RCMARL_LAUNCH(k_pk_pack_w2, dim3(hid >> 7, S * N), dim3(256), 0, stream, theta, ldp, g.o_W2, g.o_W3, hid, (unsigned char*)w2t,
                (unsigned char*)w2w3, rs, ovf_flag);
  return rcmarl_check_launch();
}

// layer 2 forward of every (seed, agent) from the packed operands.  Outputs, each optional: a2 (fp32 feature-major [S][N*hid][ldb]),
// mask_bj / mask_jb (the fit's LeakyReLU masks, packed), vpart ([S][N][hid/128][ldb], parts of the head's value).
// Replaces model(x) of the second Dense layer, agents/resilient_CAC_agents.py:95-97,114,118.
RCMARL_EXPORT int rcmarl_pk_forward2(const void* w2t, const void* a1_bk, int bk_rt, const float* theta, float* a2, void* mask_bj,
                                     int mbj_rt, void* mask_jb, int mjb_kt, float* vpart, float* npart, int S, int N, int B, int in_dim,
                                     int hid, int ldp, int ldb, void* stream) {
  if (!w2t || !a1_bk || !theta || (!a2 && !mask_bj && !mask_jb && !vpart && !npart) || !pk_dims_ok(S, N, B, hid, ldp) || in_dim <= 0 || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid & 127) return RCMARL_ERR_UNSUPPORTED;
  const int T = pk_tile(hid), ntb = rc_ceil_div(B, T), rts = ntb * (T / 128);
  if (bk_rt < rts || (mask_bj && mbj_rt < rts) || (mask_jb && mjb_kt < 4 * rts)) return RCMARL_ERR_ARG;
  const NetGeom g = make_geom(in_dim, hid, 1);
  PkFwdArgs a{};
  a.w2t = (const unsigned char*)w2t; a.a1bk = (const unsigned char*)a1_bk; a.bk_rt = bk_rt;
  a.theta = theta; a.ldp = ldp; a.o_b2 = g.o_b2; a.o_W3 = g.o_W3;
  a.a2 = a2; a.ldb = ldb; a.mask_bj = (unsigned char*)mask_bj; a.mbj_rt = mbj_rt; a.mask_jb = (unsigned char*)mask_jb; a.mjb_kt = mjb_kt;
  a.vpart = vpart; a.npart = npart; a.Z = S * N; a.B = B; a.hid = hid; a.ntb = ntb;
  return T == 256 ? pk_launch_forward2<PkBig>(a, stream) : pk_launch_forward2<PkSmall>(a, stream);
}

// mode 0: out = V; 1: out = aux + gamma V (aux = applied reward: the TD target, :114-115); 2: the MSE head of fit() (aux = y):
// out = dz3, dzv, losspart ([S][N][ceil(B/256)]).  Bp: B rounded up to 256.
RCMARL_EXPORT int rcmarl_pk_head(const float* vpart, const float* theta, const float* aux, float gamma, int mode, float* out,
                                 void* dzv, float* losspart, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream) {
  if (!vpart || !theta || !out || mode < 0 || mode > 2 || (mode >= 1 && !aux) || (mode == 2 && (!dzv || !losspart)) ||
      !pk_dims_ok(S, N, B, hid, ldp) || in_dim <= 0 || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid & 127) return RCMARL_ERR_UNSUPPORTED;
  const int Bp = rc_ceil_div(B, 256) * 256;
  const NetGeom g = make_geom(in_dim, hid, 1);
  RCMARL_LAUNCH(k_pk_head, dim3(Bp / 256, S * N), dim3(256), 0, stream, vpart, theta, ldp, g.o_b3, hid / pk_tile(hid), aux, gamma, mode, out,
                (unsigned short*)dzv, Bp, losspart, rc_ceil_div(B, 256), B, ldb);
  return rcmarl_check_launch();
}

// dz1 of every (seed, agent) straight into the lattice backward GEMM's packed operand dzp (two f16 pieces of 2^8 dz1, rows =
// agent * hid + unit; rcmarl_layer1_backward_sgd_lattice reads it) + gb1part[s][n][tile][unit] (sums of dz1 over the 128 rows of a tile).
RCMARL_EXPORT int rcmarl_pk_backward_data(const void* mask_bj, int mbj_rt, const void* w2w3, const float* rs, const unsigned* s1,
                                          int s1_ld, const float* dz3, void* dzp, int dzp_rt, int dzp_kt, float* gb1part, int* ovf_flag,
                                          int S, int N, int B, int hid, int ldb, void* stream) {
  if (!mask_bj || !w2w3 || !rs || !s1 || !dz3 || !dzp || !gb1part || !pk_dims_ok(S, N, B, hid, 1) || ldb < B) return RCMARL_ERR_ARG;
  if ((hid & 127) || !(rc_lat_f16_mode() & 2)) return RCMARL_ERR_UNSUPPORTED;
  const int T = pk_tile(hid), ntb = rc_ceil_div(B, T), rts = ntb * (T / 128);
  if (mbj_rt < rts || s1_ld < 4 * rts || dzp_rt < N * (hid >> 7) || dzp_kt < 4 * rts) return RCMARL_ERR_ARG;
  PkBdArgs a{};
  a.mask_bj = (const unsigned char*)mask_bj; a.mbj_rt = mbj_rt; a.w2w3 = (const unsigned char*)w2w3; a.rs = rs; a.s1 = s1; a.s1_ld = s1_ld;
  a.dz3 = dz3; a.ldb = ldb; a.dzp = (unsigned char*)dzp; a.dzp_rt = dzp_rt; a.dzp_kt = dzp_kt; a.gb1part = gb1part; a.ovf = ovf_flag;
  a.Z = S * N; a.N = N; a.B = B; a.hid = hid; a.ntb = ntb;
  rc_form_set(dzp, 1);
  return T == 256 ? pk_launch_backward_data<PkBig>(a, stream) : pk_launch_backward_data<PkSmall>(a, stream);
}

// W2 -= lr * gW2 for the agents with mask[n] != 0 (NULL: all); gw3part [S][N][hid/128][hid], q [S][N][hid] for rcmarl_pk_small_sgd
RCMARL_EXPORT int rcmarl_pk_backward_w2(const void* a1_kb, int kb_kt, const void* mask_jb, int mjb_kt, const void* dzv, float* theta,
                                        const int* mask, float* gw3part, float* q, int S, int N, int B, int in_dim, int hid, int ldp,
                                        float lr, void* stream) {
  if (!a1_kb || !mask_jb || !dzv || !theta || !gw3part || !q || !pk_dims_ok(S, N, B, hid, ldp) || in_dim <= 0) return RCMARL_ERR_ARG;
  if (hid & 127) return RCMARL_ERR_UNSUPPORTED;
  const int kts = rc_ceil_div(B, 32);
  if (kb_kt < kts || mjb_kt < kts) return RCMARL_ERR_ARG;
  const NetGeom g = make_geom(in_dim, hid, 1);
  PkBwArgs a{};
  a.a1kb = (const unsigned char*)a1_kb; a.kb_kt = kb_kt; a.mask_jb = (const unsigned char*)mask_jb; a.mjb_kt = mjb_kt;
  a.dzv = (const unsigned short*)dzv; a.Bp = rc_ceil_div(B, 256) * 256; a.theta = theta; a.ldp = ldp; a.o_W2 = g.o_W2; a.o_W3 = g.o_W3;
  a.mask = mask; a.lr = lr; a.gw3part = gw3part; a.qout = q; a.Z = S * N; a.N = N; a.B = B; a.hid = hid;
  return pk_tile(hid) == 256 ? pk_launch_backward_w2<PkBig>(a, stream) : pk_launch_backward_w2<PkSmall>(a, stream);
}

// b1, b2, W3, b3 -= lr * grad from the parts the two backward GEMMs and the head left; loss_out [S][N] (optional)
RCMARL_EXPORT int rcmarl_pk_small_sgd(const float* gw3part, const float* q, const float* gb1part, const float* dz3, const float* losspart,
                                      float* theta, const int* mask, float* loss_out, int S, int N, int B, int in_dim, int hid, int ldp,
                                      int ldb, float lr, void* stream) {
  if (!gw3part || !q || !gb1part || !dz3 || !losspart || !theta || !pk_dims_ok(S, N, B, hid, ldp) || in_dim <= 0 || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid & 127) return RCMARL_ERR_UNSUPPORTED;
  const NetGeom g = make_geom(in_dim, hid, 1);
  RCMARL_LAUNCH(k_pk_small_sgd, dim3(S * N), dim3(256), 0, stream, gw3part, hid / pk_tile(hid), q, gb1part, rc_ceil_div(B, pk_tile(hid)), dz3, ldb, losspart,
                rc_ceil_div(B, 256), theta, ldp, g.o_b1, g.o_b2, g.o_W3, g.o_b3, mask, loss_out, N, B, hid, lr);
  return rcmarl_check_launch();
}
