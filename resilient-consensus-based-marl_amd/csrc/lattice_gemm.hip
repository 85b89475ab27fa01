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
#include <type_traits>
#include <stdlib.h>

namespace {

__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st_u4(unsigned char* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }

// ---------------------------------------------------------------------------------------------
// encode: one workgroup = 32 replay rows x 128 features, transposed through LDS so that both packed
// images are written in 16-byte chunks.
__global__ __launch_bounds__(256) void k_lattice_encode(const float* __restrict__ x, long x_seed_stride,
                                                        const float* __restrict__ alpha, int B, int in_dim,
                                                        unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                        unsigned char* __restrict__ ktp, int ktp_rt, int ktp_kt,
                                                        int* __restrict__ flag) {
  __shared__ unsigned short tile[32][128 + 2];
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
      tile[bl][cl] = (unsigned short)rc_bf16_rne(kq);
    }
    if (bad) *flag = 1;
  }
  __syncthreads();
  // KTp block (row tile c0/128, k-tile b0/32): 128 rows x 4 chunks
  if (ktp != nullptr && (c0 >> 7) < ktp_rt && (b0 >> 5) < ktp_kt) {
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

// ---------------------------------------------------------------------------------------------
// W1 split: one workgroup = 128 (agent,unit) columns x 32 features -> three 8-KiB blocks.
__global__ __launch_bounds__(256) void k_w1_split(const float* __restrict__ theta, const float* __restrict__ alpha,
                                                  unsigned char* __restrict__ wp, int N, int in_dim, int ldp,
                                                  int wp_rt, int wp_kt, int hid) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, r = t & 127;
  const int col = rt * 128 + r, ncols = N * hid;
  const bool col_ok = col < ncols;
  const int ag = col_ok ? col / hid : 0, j = col - ag * hid;
  const float* th = theta + ((long)s * N + ag) * ldp + j;
  unsigned char* blk = wp + (long)s * wp_rt * wp_kt * 3 * RC_PK_BLOCK + ((long)rt * wp_kt + kt) * 3 * RC_PK_BLOCK;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c4 = (t >> 7) + 2 * q;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kt * 32 + 8 * c4 + e;
      const bool ok = col_ok && k < in_dim;
      w[e] = ok ? th[(long)k * hid] * alpha[k] : 0.f;
    }
    uint4 vh, vm, vl;
    rc_split3_pair(w[0], w[1], vh.x, vm.x, vl.x);
    rc_split3_pair(w[2], w[3], vh.y, vm.y, vl.y);
    rc_split3_pair(w[4], w[5], vh.z, vm.z, vl.z);
    rc_split3_pair(w[6], w[7], vh.w, vm.w, vl.w);
    const int o = r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4);
    st_u4(blk + o, vh);
    st_u4(blk + RC_PK_BLOCK + o, vm);
    st_u4(blk + 2 * RC_PK_BLOCK + o, vl);
  }
}

// ---------------------------------------------------------------------------------------------
// dz pack: fp32 feature-major dz[S][rows][ldb] (rows = (agent,unit)) -> three exact bf16 pieces in PK form
// (reduction = replay row), zero beyond B / beyond the last row: the backward operand of a wide net, whose dz1 comes
// out of a dense GEMM (wide_kernels.hip) instead of mid_fit's fused epilogue.  One workgroup = 128 rows x 32 rows b.
__global__ __launch_bounds__(256) void k_dz_pack(const float* __restrict__ dz, unsigned char* __restrict__ dzp, int nrows,
                                                 int B, int ldb, int dzp_rt, int dzp_kt) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, c4 = t & 3;             // four adjacent lanes = the 128 B (32 k) of one row: coalesced both ways
  unsigned char* blk = dzp + (long)s * dzp_rt * dzp_kt * 3 * RC_PK_BLOCK + ((long)rt * dzp_kt + kt) * 3 * RC_PK_BLOCK;
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
    uint4 vh, vm, vl;
    rc_split3_pair(w[0], w[1], vh.x, vm.x, vl.x);
    rc_split3_pair(w[2], w[3], vh.y, vm.y, vl.y);
    rc_split3_pair(w[4], w[5], vh.z, vm.z, vl.z);
    rc_split3_pair(w[6], w[7], vh.w, vm.w, vl.w);
    const int o = r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4);
    st_u4(blk + o, vh);
    st_u4(blk + RC_PK_BLOCK + o, vm);
    st_u4(blk + 2 * RC_PK_BLOCK + o, vl);
  }
}

// ---------------------------------------------------------------------------------------------
// WM x WN wavefronts per workgroup (default 2 x 2), each owning MT x NT accumulator blocks of 32 x 32
template <int PA, int PB, int MT, int NT, int WM = 2, int WN = 2> struct LatCfg {
  static constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN, NWV = WM * WN;
  static_assert(BM % 128 == 0 && BN % 128 == 0, "block tile sides are multiples of 128");
  static constexpr int ART = BM / 128, BRT = BN / 128;              // 128-row tiles per block side
  static constexpr int A_KB = ART * PA * 8, B_KB = BRT * PB * 8;    // KiB per k-tile stage
  static constexpr int STAGE_KB = A_KB + B_KB, STAGE_BYTES = STAGE_KB * 1024;
  static constexpr int GLDS = STAGE_KB / NWV;                       // 1-KiB bursts per wavefront per stage
  static_assert(STAGE_KB % NWV == 0, "stage splits evenly over the wavefronts");
};

struct LatOperands {
  const unsigned char* a; const unsigned char* b;   // seed base of each packed operand
  int a_kt, b_kt;                                   // allocated k-tiles (block stride along the row-tile axis)
  int art0, brt0;                                   // first 128-row tile of this workgroup on each side
  int pf = 0;                                       // > 0: touch the 3-piece operand's k-tile t+pf (L2 prefetch)
};

// DBG (measurement aid, RCMARL_LAT_DBG; results are WRONG for DBG != 0): bit 0 = no LDS-DMA after the prologue,
// bit 1 = no vmcnt wait / barrier, bit 2 = fragments read from LDS once (k-loop = matrix core only)
// SPREAD: the LDS-DMA bursts of the next k-tile are issued one at a time BETWEEN the matrix-core instructions of the first
// half of this k-tile (measured: every 2nd MFMA 717-733 us, every 3rd 729-736, every 4th 746 on the backward) instead of back to back right after the barrier (an LDS-DMA instruction blocks the wavefront's
// issue for 60-180 cycles; in a burst those add up while no MFMA of this wavefront is in flight).
template <int PA, int PB, int MT, int NT, int NSTAGE, int DBG = 0, int WM = 2, int WN = 2, bool SPREAD = false>
__device__ __forceinline__ void lat_mainloop(const LatOperands& op, int n_ktiles, unsigned char* lds,
                                             rc_f32x16 (&acc)[MT][NT]) {
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  static_assert(NSTAGE == 2 || NSTAGE == 3, "LDS ring depth");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  // wave-uniform source of each of this wavefront's bursts at k-tile 0, and its per-k-tile advance
  const unsigned char* gsrc[C::GLDS];
  int gstep[C::GLDS];
#pragma unroll
  for (int i = 0; i < C::GLDS; ++i) {
    const int q = wave + C::NWV * i;
    if (q < C::A_KB) {
      const int seg = q / (PA * 8), off = q - seg * (PA * 8);
      gsrc[i] = op.a + ((long)(op.art0 + seg) * op.a_kt) * (PA * RC_PK_BLOCK) + off * 1024;
      gstep[i] = PA * RC_PK_BLOCK;
    } else {
      const int q2 = q - C::A_KB;
      const int seg = q2 / (PB * 8), off = q2 - seg * (PB * 8);
      gsrc[i] = op.b + ((long)(op.brt0 + seg) * op.b_kt) * (PB * RC_PK_BLOCK) + off * 1024;
      gstep[i] = PB * RC_PK_BLOCK;
    }
  }
  const unsigned lane16 = lane * 16;
  const rc_lds_t lds0 = rc_lds_addr(lds) + wave * 1024;
  auto stage = [&](int buf, int t) {
    const rc_lds_t dst = lds0 + buf * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::GLDS; ++i) RC_GLDS16S(gsrc[i] + (long)t * gstep[i], lane16, dst + i * (C::NWV * 1024));
  };
  auto stage_one = [&](int buf, int t, int i) {
    RC_GLDS16S(gsrc[i] + (long)t * gstep[i], lane16, lds0 + buf * C::STAGE_BYTES + i * (C::NWV * 1024));
  };
  constexpr int N_MFMA = 2 * PA * PB * MT * NT;                 // matrix-core instructions of a wavefront per k-tile
#ifdef RC_LAT_SPREAD_EVERY
  constexpr int EVERY = RC_LAT_SPREAD_EVERY;                    // (tuning builds)
#else
  constexpr int EVERY = (N_MFMA / 2) / C::GLDS > 0 ? (N_MFMA / 2) / C::GLDS : 1;     // all bursts within the first half
#endif
  static_assert(!SPREAD || (NSTAGE == 2 && EVERY * C::GLDS <= N_MFMA), "spread issue: two stages, every burst has a slot");

  // fragment addresses: row = lane&31 (+ tile offsets), chunk = 2*kstep + lane>>5, XOR (row>>2)&3
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

  // L2 prefetch (op.pf > 0, NSTAGE 2): the 3-piece operand is streamed from HBM once per workgroup, and the LDS ring
  // gives its loads only one k-tile of lead.  Lanes 0..191 of the workgroup each touch one 128-B line of k-tile
  // t + pf (24 KiB, contiguous in the packed format) with a plain load whose result is never used; the DMA of that
  // tile then finds its lines in L2.  The touch is the YOUNGEST memory instruction of its wavefront when the next
  // k-tile starts, so the wait before the barrier is vmcnt(1) there: it never waits for the prefetch itself.
  constexpr bool pf_ok = PA == 3 ? C::ART == 1 : (PB == 3 && C::BRT == 1);     // the 3-piece side is one 128-row tile wide
  const int pfd = (NSTAGE == 2 && pf_ok) ? op.pf : 0;
  const bool pf_wave = wave < 3;
  const unsigned char* pf_src = (PA == 3 ? op.a + (long)op.art0 * op.a_kt * (3 * RC_PK_BLOCK)
                                         : op.b + (long)op.brt0 * op.b_kt * (3 * RC_PK_BLOCK)) + threadIdx.x * 128;
  float pfv = 0.f;
  bool pf_flying = false;

  // ring of NSTAGE stages: tile t+NSTAGE-1 is requested right after the barrier that retires tile t-1
  stage(0, 0);
  if (NSTAGE == 3 && n_ktiles > 1) stage(1, 1);
  int cur = 0;                                        // t % NSTAGE
  for (int t = 0; t < n_ktiles; ++t) {
    // this wavefront's bursts of tile t have landed (NSTAGE 3: tile t+1's may still be in flight) ...
    if (!(DBG & 2) || t == 0) {
      if (NSTAGE == 3 && t + 1 < n_ktiles) RC_WAIT_VMEM_N(C::GLDS);
      else if (pf_flying && pf_wave) RC_WAIT_VMEM_N(1);
      else RC_WAIT_VMEM();
      __syncthreads();              // ... and everybody's; all reads of the buffer refilled next are done
    }
    const bool more = t + NSTAGE - 1 < n_ktiles && !(DBG & 1);
    const int nb = cur == 0 ? NSTAGE - 1 : cur - 1;             // (t + NSTAGE - 1) % NSTAGE
    if (more && !SPREAD) stage(nb, t + NSTAGE - 1);
    pf_flying = pfd > 0 && t + pfd < n_ktiles;
#ifndef RCMARL_EMU
    if (pf_flying && pf_wave)
      asm volatile("global_load_dword %0, %1, off" : "=v"(pfv) : "v"(pf_src + (long)(t + pfd) * (3 * RC_PK_BLOCK)) : "memory");
#endif
    const unsigned char* st = lds + ((DBG & 4) ? 0 : cur) * C::STAGE_BYTES;
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ks == 0 ? co0 : co1;
      uint4 af[MT][PA], bf[NT][PB];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < PA; ++p) af[mt][p] = ld_u4(st + offA[mt] + p * RC_PK_BLOCK + co);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < PB; ++p) bf[nt][p] = ld_u4(st + offB[nt] + p * RC_PK_BLOCK + co);
      // smallest pieces first; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int pa = PA - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = PB - 1; pb >= 0; --pb)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = rc_mfma_bf16(af[mt][pa], bf[nt][pb], acc[mt][nt]);
              if constexpr (SPREAD) {
                const int o = ks * (PA * PB * MT * NT) + (((PA - 1 - pa) * PB + (PB - 1 - pb)) * MT + mt) * NT + nt;
                if (o % EVERY == EVERY - 1 && o / EVERY < C::GLDS) {
                  RC_SCHED_FENCE();
                  if (more) stage_one(nb, t + NSTAGE - 1, o / EVERY);
                  RC_SCHED_FENCE();
                }
              }
            }
    }
  }
#ifndef RCMARL_EMU
  if (pfd > 0) {                    // the register of the last touch must not be reused while that load is in flight
    RC_WAIT_VMEM();
    asm volatile("" ::"v"(pfv));
  }
#endif
}

// The backward GEMM's k-loop with its THREE-PIECE operand (dz, the B side) loaded straight from global memory into
// registers: the packed image already lies in fragment order (the LDS stage was a verbatim copy of it), so a lane's
// fragment of (row tile, k-tile, piece, k16 step) is one 16-byte load.  LDS then carries only the one-piece operand:
// 16 KiB of LDS-DMA + 32 KiB of fragment reads per workgroup and k-tile instead of 40 + 80 (the
// long k-loop of the backward is limited by operand delivery into the CU, not by the matrix core).  The two wavefronts
// that share a dz row block both load it (the second hits L1/L2).  Fragments of k-tile t+1 are requested into the
// registers a k16 step has just freed, i.e. half a k-tile to a full k-tile ahead of their use; they are PLAIN loads
// (hipcc counts them itself; the asm LDS-DMA it cannot see only makes its vmcnt waits conservative).
template <int MT, int NT, int WM, int WN>
__device__ __forceinline__ void lat_mainloop_bdirect(const LatOperands& op, int n_ktiles, unsigned char* lds,
                                                     rc_f32x16 (&acc)[MT][NT]) {
  constexpr int PB = 3, NWV = WM * WN, BM = 32 * MT * WM, ART = BM / 128, A_KB = ART * 8, GLDS = A_KB / NWV;
  constexpr int STAGE_BYTES = A_KB * 1024;
  static_assert(BM % 128 == 0 && A_KB % NWV == 0 && 32 * NT * WN == 128, "tile shape of the B-direct k-loop");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
  const unsigned char* gsrc[GLDS];
#pragma unroll
  for (int i = 0; i < GLDS; ++i) {
    const int q = wave + NWV * i, seg = q / 8, off = q - seg * 8;
    gsrc[i] = op.a + ((long)(op.art0 + seg) * op.a_kt) * RC_PK_BLOCK + off * 1024;
  }
  const unsigned lane16 = lane * 16;
  const rc_lds_t lds0 = rc_lds_addr(lds) + wave * 1024;
  auto stage = [&](int buf, int t) {
    const rc_lds_t dst = lds0 + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < GLDS; ++i) RC_GLDS16S(gsrc[i] + (long)t * RC_PK_BLOCK, lane16, dst + i * (NWV * 1024));
  };
  const int sw = (l31 >> 2) & 3;
  const int co0 = ((0 + half) ^ sw) << 4, co1 = ((2 + half) ^ sw) << 4;
  int offA[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = wm * 32 * MT + 32 * mt + l31;
    offA[mt] = (row >> 7) * RC_PK_BLOCK + (row & 127) * 64;
  }
  // this lane's rows of the dz tile (one 128-row tile wide); k-tile t, piece p, k16 step ks at  t*3*8 KiB + p*8 KiB + co_ks
  const unsigned char* gB = op.b + (long)op.brt0 * op.b_kt * (PB * RC_PK_BLOCK);
  unsigned offB[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) offB[nt] = (unsigned)(wn * 32 * NT + 32 * nt + l31) * 64;
  uint4 b0[NT][PB], b1[NT][PB];
  auto loadB = [&](uint4 (&b)[NT][PB], int t, int co) {
    const unsigned char* base = gB + (long)t * (PB * RC_PK_BLOCK) + co;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int p = 0; p < PB; ++p) b[nt][p] = ld_u4(base + p * RC_PK_BLOCK + offB[nt]);
  };
  auto kstep = [&](const unsigned char* st, int co, const uint4 (&b)[NT][PB]) {
    uint4 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = ld_u4(st + offA[mt] + co);
#pragma unroll
    for (int pb = PB - 1; pb >= 0; --pb)            // smallest pieces first, as lat_mainloop: the same fp32 sums
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = rc_mfma_bf16(af[mt], b[nt][pb], acc[mt][nt]);
  };
  stage(0, 0);
  loadB(b0, 0, co0);
  loadB(b1, 0, co1);
  RC_SCHED_FENCE();
  for (int t = 0; t < n_ktiles; ++t) {
    const int cur = t & 1;
    // the A bursts of tile t are older than the (at most) 2 x NT x PB fragment loads issued after them
    RC_WAIT_VMEM_N(2 * NT * PB);
    __syncthreads();                                  // everybody's bursts landed; all reads of the other stage are done
    const bool more = t + 1 < n_ktiles;
    if (more) stage(cur ^ 1, t + 1);
    const unsigned char* st = lds + cur * STAGE_BYTES;
    kstep(st, co0, b0);
    RC_SCHED_FENCE();
    if (more) loadB(b0, t + 1, co0);
    RC_SCHED_FENCE();
    kstep(st, co1, b1);
    RC_SCHED_FENCE();
    if (more) loadB(b1, t + 1, co1);
    RC_SCHED_FENCE();
  }
}

// The same k-loop on a ring of FOUR half-stages (k16 each, 20 KiB; same 80 KiB of LDS): the loads of half-stage
// h+3 are requested while half-stage h is computed, i.e. 1.5 k-tiles of lead instead of 1, and the wait before a
// barrier is a COUNTED vmcnt that leaves the two youngest half-stages in flight.  A half-stage takes the 32-byte
// half of every 64-byte packed row that holds logical chunks {2h, 2h+1} (physical half h ^ bit3(row), the
// format's XOR swizzle); the LDS image is [128 rows][32 B] per block with the two chunks placed so that the
// fragment read  row*32 + ((lane>>5) ^ bit3(row))*16  is bank-conflict free (the placement is made on the SOURCE
// address of the LDS-DMA, whose destination is lane-linear).
template <int PA, int PB, int MT, int NT>
__device__ __forceinline__ void lat_mainloop_half(const LatOperands& op, int n_ktiles, unsigned char* lds,
                                                  rc_f32x16 (&acc)[MT][NT]) {
  typedef LatCfg<PA, PB, MT, NT> C;
  constexpr int HALF_BYTES = C::STAGE_BYTES / 2, HBLK = RC_PK_BLOCK / 2;   // 4 KiB per (row tile, piece) per half
  constexpr int GL = C::STAGE_KB / 2 / 4;                                  // bursts per wavefront per half-stage
  static_assert((C::STAGE_KB / 2) % 4 == 0, "half-stage splits evenly over 4 wavefronts");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
  // burst q (1 KiB = 32 rows x 32 B of one block): block index q/4, rows 32*(q%4)..; lane -> (row, slot u)
  const unsigned char* gsrc[GL];
  int gstep[GL];
  unsigned voff0[GL], voff1[GL];            // per-lane source offsets for h = 0 / h = 1
#pragma unroll
  for (int i = 0; i < GL; ++i) {
    const int q = wave + 4 * i;
    const int blk = q >> 2, r = 32 * (q & 3) + (lane >> 1), u = lane & 1;
    const int b3 = (r >> 3) & 1, b2 = (r >> 2) & 1;
    const unsigned sub = (unsigned)((u ^ b2 ^ b3) << 4);
    voff0[i] = r * 64 + ((0 ^ b3) << 5) + sub;
    voff1[i] = r * 64 + ((1 ^ b3) << 5) + sub;
    if (blk < C::ART * PA) {
      const int seg = blk / PA, pc = blk - seg * PA;
      gsrc[i] = op.a + ((long)(op.art0 + seg) * op.a_kt) * (PA * RC_PK_BLOCK) + pc * RC_PK_BLOCK;
      gstep[i] = PA * RC_PK_BLOCK;
    } else {
      const int b2i = blk - C::ART * PA;
      const int seg = b2i / PB, pc = b2i - seg * PB;
      gsrc[i] = op.b + ((long)(op.brt0 + seg) * op.b_kt) * (PB * RC_PK_BLOCK) + pc * RC_PK_BLOCK;
      gstep[i] = PB * RC_PK_BLOCK;
    }
  }
  const rc_lds_t lds0 = rc_lds_addr(lds) + wave * 1024;
  auto stage = [&](int slot, int hs) {                         // half-stage hs = 2*t + h
    const rc_lds_t dst = lds0 + slot * HALF_BYTES;
    const int t = hs >> 1;
    if (hs & 1) {
#pragma unroll
      for (int i = 0; i < GL; ++i) RC_GLDS16S(gsrc[i] + (long)t * gstep[i], voff1[i], dst + i * 4096);
    } else {
#pragma unroll
      for (int i = 0; i < GL; ++i) RC_GLDS16S(gsrc[i] + (long)t * gstep[i], voff0[i], dst + i * 4096);
    }
  };
  const int b3l = (l31 >> 3) & 1;
  const int co = (half ^ b3l) << 4;
  int offA[MT], offB[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = wm * 32 * MT + 32 * mt + l31;
    offA[mt] = (row >> 7) * PA * HBLK + (row & 127) * 32 + co;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wn * 32 * NT + 32 * nt + l31;
    offB[nt] = C::ART * PA * HBLK + (row >> 7) * PB * HBLK + (row & 127) * 32 + co;
  }
  const int total = 2 * n_ktiles;
  stage(0, 0);
  if (total > 1) stage(1, 1);
  if (total > 2) stage(2, 2);
  int cur = 0;
  for (int hs = 0; hs < total; ++hs) {
    if (hs + 2 < total) RC_WAIT_VMEM_N(2 * GL);
    else if (hs + 1 < total) RC_WAIT_VMEM_N(GL);
    else RC_WAIT_VMEM();
    __syncthreads();
    if (hs + 3 < total) stage((cur + 3) & 3, hs + 3);
    const unsigned char* st = lds + cur * HALF_BYTES;
    cur = (cur + 1) & 3;
    uint4 af[MT][PA], bf[NT][PB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < PA; ++p) af[mt][p] = ld_u4(st + offA[mt] + p * HBLK);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int p = 0; p < PB; ++p) bf[nt][p] = ld_u4(st + offB[nt] + p * HBLK);
#pragma unroll
    for (int pa = PA - 1; pa >= 0; --pa)
#pragma unroll
      for (int pb = PB - 1; pb >= 0; --pb)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = rc_mfma_bf16(af[mt][pa], bf[nt][pb], acc[mt][nt]);
  }
}

// workgroup id -> (seed, tile w within the seed); all tiles of a seed on one XCD when S % 8 == 0
__device__ __forceinline__ void lat_decode(int g, int per_seed, int S, int& seed, int& w) {
  if ((S & 7) == 0) {
    const int xcd = g & 7, q = g >> 3;
    seed = xcd + 8 * (q / per_seed);
    w = q % per_seed;
  } else {
    seed = g / per_seed;
    w = g - seed * per_seed;
  }
}

// De-phasing aid: the two workgroups that share a CU start together and would otherwise run their
// (memory-bound) prologue/epilogue and their (matrix-core-bound) k-loop in lockstep.  First-wave workgroups
// whose bit `bit` of (id/8) is set sleep `n` x ~4 us once, so the pair drifts half a tile apart and one's
// epilogue overlaps the other's k-loop.  Pure scheduling: no effect on results.
__device__ __forceinline__ void lat_stagger(int bit, int n) {
  if (bit >= 0 && blockIdx.x < 512u && (((blockIdx.x >> 3) >> bit) & 1u)) rc_sleep(n);
}

// Static priority for every other workgroup (bit `bit` of id/8): the two workgroups that share a SIMD otherwise
// interleave their MFMAs round-robin, finish their k-tile bursts together and then both sit in the barrier/load
// phase with the matrix pipe idle.  With one of them at s_setprio 1 its burst runs first and the other's fills the
// gap: the pair ping-pongs instead of marching in step.  Pure scheduling hint: no effect on results.
__device__ __forceinline__ void lat_prio(int bit) {
  if (bit >= 0 && (((blockIdx.x >> 3) >> bit) & 1u)) rc_setprio1();
}

// ---- forward: A = W' pieces (rows = (agent,unit) columns), B = K (rows = replay rows) ----------
// W8: eight wavefronts per workgroup (2 x 4, each 64 x 64) on the SAME 128 x 256 block tile and LDS stages: 64 instead of
// 128 accumulator registers per wavefront, i.e. four instead of two wavefronts per SIMD at two workgroups per CU
template <int NSTAGE, int DBG = 0, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256, NSTAGE == 3 ? 1 : 2) void k_lat_forward(const unsigned char* __restrict__ wp, int wp_rt, int wp_kt,
                                                        const unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                        const float* __restrict__ theta, float* __restrict__ a1t, int S,
                                                        int N, int B, int in_dim, int ldp, int ldb, int mtiles,
                                                        int ntiles, int dbg_same_tile, int stg_bit, int stg_n, int hid) {
  constexpr int PA = 3, PB = 1, MT = 2, NT = W8 ? 2 : 4, WM = 2, WN = W8 ? 4 : 2;
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  RCMARL_DYN_SMEM(unsigned char, lds);
  lat_stagger((stg_bit & 0xff) - 1, stg_n);
  lat_prio(((stg_bit >> 8) & 0xff) - 1);
  const int cw = dbg_same_tile >> 8;
  dbg_same_tile &= 0xff;
  // persistent form (RCMARL_LAT_PERSIST): the grid is one resident wave of workgroups, each walks tiles g, g + grid, ..
  // (g & 7 stays the XCD, so a seed's tiles stay on one XCD); otherwise the grid covers the tiles and this runs once
  for (int g = blockIdx.x; g < S * mtiles * ntiles; g += gridDim.x) {
  int s, w;
  lat_decode(g, mtiles * ntiles, S, s, w);
  // Tile order inside a seed: the n-tiles are walked in chunks of `cw` (dbg_same_tile bits 8..), m-major inside a
  // chunk, n fastest.  The workgroups resident on an XCD (64) then share ONE chunk of the replay operand (cw x 256 KiB
  // at 512 inputs) plus a sliding window of W' panels -- inside the 4-MiB L2 -- instead of all n-tiles (3 MiB) plus
  // five W' panels (PMC: 2.6 GB fetched per launch for 0.3 GB of operands with the plain n-fastest order).
  int bn, bm;
  if (cw <= 0 || cw >= ntiles) { bn = w % ntiles; bm = w / ntiles; }
  else {
    const int per_chunk = mtiles * cw, c = w / per_chunk, r = w - c * per_chunk;
    const int wc = min(cw, ntiles - c * cw);
    bm = r / wc; bn = c * cw + (r - bm * wc);
  }
  LatOperands op;
  op.a = wp + (long)s * wp_rt * wp_kt * (PA * RC_PK_BLOCK); op.a_kt = wp_kt; op.art0 = bm * C::ART;
  op.b = kp + (long)s * kp_rt * kp_kt * (PB * RC_PK_BLOCK); op.b_kt = kp_kt; op.brt0 = bn * C::BRT;
  op.pf = (stg_bit >> 24) & 0xf;
  if (dbg_same_tile) {              // measurement aid only (RCMARL_LAT_SAMETILE=1): every workgroup streams ONE
    op.a = wp; op.art0 = 0;         // panel pair, i.e. the k-loop with a perfectly cached memory system
    op.b = kp; op.brt0 = 0;
  }
  rc_f32x16 acc[MT][NT];
  if constexpr (NSTAGE == 4) lat_mainloop_half<PA, PB, MT, NT>(op, (in_dim + 31) >> 5, lds, acc);
  else if constexpr (NSTAGE == 6) lat_mainloop<PA, PB, MT, NT, 2, DBG, WM, WN, true>(op, (in_dim + 31) >> 5, lds, acc);
  else lat_mainloop<PA, PB, MT, NT, NSTAGE, DBG, WM, WN>(op, (in_dim + 31) >> 5, lds, acc);
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
  const bool full_m = (bm + 1) * C::BM <= ncols;                   // workgroup-uniform
  auto store_tile = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = bn * C::BN + wn * 32 * NT + 32 * nt + (lane & 31);
      if (n < B) {
        // byte offset of the lane inside its row block: 32 bits (one seed's a1t is < 4 GiB) -> saddr + voffset stores
        const unsigned lane_byte = ((unsigned)(4 * half) * (unsigned)ldb + (unsigned)n) * 4u;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m0 = bm * C::BM + wm * 32 * MT + 32 * mt;      // wave-uniform
          unsigned char* __restrict__ rowbase = reinterpret_cast<unsigned char*>(a1t_s + (long)m0 * ldb);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float z = acc[mt][nt][r] + bv[mt][r];
            const float o = fmaxf(z, RC_LEAK * z);                 // == rc_lrelu(z) bit for bit (0 < leak < 1), 2 ops not 3
            float* __restrict__ dst = reinterpret_cast<float*>(rowbase + (long)dr * ldb * 4 + lane_byte);
            if (DBG & 8) { if (o == 12345.678f) *dst = o; }        // DBG 8: no epilogue stores
            else if (FULL || m0 + dr + 4 * half < ncols) RC_NT_STORE(dst, o);
          }
        }
      }
    }
  };
  if (full_m) store_tile(std::true_type{}); else store_tile(std::false_type{});
  __syncthreads();                  // (persistent form: the bias words share LDS with the next tile's first stage)
  }
}

// ---- backward: A = K^T (rows = features), B = dz1 pieces (rows = (agent,unit) columns) ----------
// (explicit work-group size + waves per SIMD: with __launch_bounds__(512, 2) hipcc allots 129 registers to the W8 form,
// one too many for the four wavefronts per SIMD that two 8-wavefront workgroups per CU need)
#ifdef RCMARL_EMU
#define RC_LAT_OCC(threads, waves)
#else
#define RC_LAT_OCC(threads, waves) __attribute__((amdgpu_flat_work_group_size(threads, threads), amdgpu_waves_per_eu(waves)))
#endif
// NSTAGE 7 = two stages, EIGHT wavefronts of 128 x 64 on a 256 x 256 tile (one workgroup per CU): the one-piece operand's
// stage is shared by twice the dz columns, 64 instead of 80 KiB of LDS-DMA per 256 x 256 x 32 of work
template <int NSTAGE, int DBG = 0, bool W8 = false>
__global__ RC_LAT_OCC((W8 || NSTAGE == 7 || NSTAGE == 8) ? 512 : 256, NSTAGE == 3 ? 1 : (W8 ? 4 : 2))
void k_lat_backward_sgd(const unsigned char* __restrict__ ktp, int ktp_rt, int ktp_kt,
                                                             const unsigned char* __restrict__ dzp, int dzp_rt, int dzp_kt,
                                                             const float* __restrict__ alpha, float* __restrict__ theta,
                                                             const int* __restrict__ mask, int S, int N, int B,
                                                             int in_dim, int ldp, float lr, int mtiles, int ntiles,
                                                             unsigned char* __restrict__ wp_out, int wp_rt, int wp_kt,
                                                             int stg_bit, int stg_n, int hid, int wp_fit) {
  // NSTAGE 8 = two stages, eight wavefronts of 128 x 64 on a 512 x 128 tile: the whole input width of the critic in one workgroup
  // (the three-piece dz panel is read once instead of once per 256 input rows)
  constexpr int PA = 1, PB = 3, MT = W8 ? 2 : 4, NT = 2, WM = (W8 || NSTAGE == 8) ? 4 : 2, WN = NSTAGE == 7 ? 4 : 2;
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
  RCMARL_DYN_SMEM(unsigned char, lds);
  lat_stagger((stg_bit & 0xff) - 1, stg_n);
  lat_prio(((stg_bit >> 8) & 0xff) - 1);
  for (int g = blockIdx.x; g < S * mtiles * ntiles; g += gridDim.x) {     // persistent form: see k_lat_forward
  int s, w;
  lat_decode(g, mtiles * ntiles, S, s, w);
  const int bm = w % mtiles, bn = w / mtiles;                      // m fastest: neighbours share the dz panel
  LatOperands op;
  op.a = ktp + (long)s * ktp_rt * ktp_kt * (PA * RC_PK_BLOCK); op.a_kt = ktp_kt; op.art0 = bm * C::ART;
  op.b = dzp + (long)s * dzp_rt * dzp_kt * (PB * RC_PK_BLOCK); op.b_kt = dzp_kt; op.brt0 = bn * C::BRT;
  op.pf = (stg_bit >> 24) & 0xf;
  if (DBG & 8) {                    // measurement aid: every workgroup streams the same panel pair (all L2 hits)
    op.a = ktp; op.art0 = 0; op.b = dzp; op.brt0 = 0;
  }
  rc_f32x16 acc[MT][NT];
  if constexpr (NSTAGE == 4) lat_mainloop_half<PA, PB, MT, NT>(op, (B + 31) >> 5, lds, acc);
  else if constexpr (NSTAGE == 5) lat_mainloop_bdirect<MT, NT, WM, WN>(op, (B + 31) >> 5, lds, acc);
  else if constexpr (NSTAGE == 6) lat_mainloop<PA, PB, MT, NT, 2, DBG, WM, WN, true>(op, (B + 31) >> 5, lds, acc);
  else if constexpr (NSTAGE == 7 || NSTAGE == 8) lat_mainloop<PA, PB, MT, NT, 2, DBG, WM, WN>(op, (B + 31) >> 5, lds, acc);
  else lat_mainloop<PA, PB, MT, NT, NSTAGE, DBG, WM, WN>(op, (B + 31) >> 5, lds, acc);
  // epilogue: W1[k][col] -= lr * alpha_k * acc; optionally the forward operand of the NEXT step is produced here
  // too (wp_out: bf16x3 pieces of alpha_k * W1_new, exactly what rcmarl_w1_split would write), so the local fit
  // needs no separate split pass.  A lane holds 4 consecutive k per (m-tile, register group) = half a 16-byte chunk.
  const int ncols = N * hid;
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
    if (col < ncols) {
      const int ag = col / hid, j = col - ag * hid;
      const bool upd = mask == nullptr || mask[ag];
      // the lane's first row: k = bm*BM + wm*32*MT + 4*half; rows of (mt, gq, e) follow at uniform distances
      const int k0 = bm * C::BM + wm * 32 * MT + 4 * half;
      float* th = theta + ((long)s * N + ag) * ldp + j + (long)k0 * hid;
      // row of this column in the forward operand: natural order (column tile bn, row cl), or the fit order of the
      // fused local-fit kernel (wp_fit; rcmarl_lattice.h)
      const int wr = wp_fit ? rc_fit_row(ag, j) : col;
      unsigned char* wrow = wp_out == nullptr ? nullptr
          : wp_out + (long)s * wp_rt * wp_kt * (3 * RC_PK_BLOCK) + (long)(wr >> 7) * wp_kt * (3 * RC_PK_BLOCK) + (wr & 127) * 64 + half * 8;
      const int sw = (wr >> 2) & 3;
      float wold[MT][16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int dk = 32 * mt + 8 * (q >> 2) + (q & 3);               // uniform
          wold[mt][q] = (full_k || k0 + dk < in_dim) ? th[(long)dk * hid] : 0.f;
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int kt = bm * (C::BM / 32) + wm * MT + mt;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
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
          if (wrow != nullptr && kt < wp_kt) {
            unsigned h0, m0, l0, h1, m1, l1;
            rc_split3_pair(wn4[0], wn4[1], h0, m0, l0);
            rc_split3_pair(wn4[2], wn4[3], h1, m1, l1);
            unsigned char* q = wrow + (long)kt * (3 * RC_PK_BLOCK) + ((gq ^ sw) << 4);
            *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(q + RC_PK_BLOCK) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(q + 2 * RC_PK_BLOCK) = make_uint2(l0, l1);
          }
        }
      }
    }
  }
  __syncthreads();                  // (persistent form: the alpha words share LDS with the next tile's first stage)
  }
}

template <class K>
bool lat_want_lds(K kernel, size_t smem) {
  return rc_want_lds(kernel, smem);
}

int lat_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// packed scheduling knobs handed to the kernels: bits 0-7 = stagger bit + 1 (0 = off), bits 8.. = priority bit + 1
int lat_stagger_bit() {
  static int v = ((lat_env_int("RCMARL_LAT_PF", 0) & 0xf) << 24) | (((lat_env_int("RCMARL_LAT_PRIO_BIT", -1) + 1) & 0xff) << 8) |
                 ((lat_env_int("RCMARL_LAT_STAGGER_BIT", -1) + 1) & 0xff);
  return v;
}
// RCMARL_LAT_PERSIST=k (tuning knob, default 0 = one workgroup per tile): launch k resident waves of workgroups
// (k x 256 CUs x workgroups per CU) and let each walk several tiles
int lat_grid(int tiles, int ns) {
  static const int k = lat_env_int("RCMARL_LAT_PERSIST", 0);
  const int resident = rc_persistent_grid(256 * (ns == 3 ? 1 : 2) * k);
  return (k > 0 && resident < tiles) ? resident : tiles;
}
int lat_stagger_n() { static int v = lat_env_int("RCMARL_LAT_STAGGER_N", 3); return v; }
// Eight wavefronts of 64 x 64 per workgroup instead of four of 64 x 128 / 128 x 64 (four instead of two wavefronts per
// SIMD on the same tiles and LDS stages).  Measured at cfg 4: forward 749 / 988 us against 776 / 1037 (critic / TR shape),
// backward 780 / 1103 against 756 / 1103 -- twice the occupancy buys 3-5 % on one kernel and nothing on the other: the
// k-loops are bound by operand delivery into the CU, not by latency.  Default: forward on, backward off;
// RCMARL_LAT_W8=0 / 1 forces both (read at every call).
bool lat_w8(bool forward) { const char* e = getenv("RCMARL_LAT_W8"); return e ? atoi(e) != 0 : forward; }

// LDS ring of the lattice GEMMs: 2 full k32 stages (80 KiB, two workgroups per CU), 3 (120 KiB, one workgroup per
// CU, a tile more of load lead; measured slower) or 4 HALF stages (80 KiB, 1.5 tiles of lead, counted vmcnt).
// RCMARL_LAT_STAGES is a tuning knob, read once.
int lat_stages() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RCMARL_LAT_STAGES");
    v = e ? atoi(e) : 2;
    if (v != 2 && v != 3 && v != 4) v = 2;
  }
  return v;
}
// backward only: RCMARL_LAT_BDIRECT=1 loads the three-piece dz fragments global -> registers (lat_mainloop_bdirect)
bool lat_bdirect() { return lat_env_int("RCMARL_LAT_BDIRECT", 0) != 0; }      // (read per call: tests switch it)
// RCMARL_LAT_SPREAD: bit 0 = forward, bit 1 = backward: LDS-DMA bursts issued between the matrix-core instructions (lat_mainloop).
// Default 2: measured -2 % on the backward at both cfg-4 shapes, +1..8 % on the forward; bit-identical either way.
int lat_spread() { return lat_env_int("RCMARL_LAT_SPREAD", 2); }

}  // namespace

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
  RCMARL_LAUNCH(k_lattice_encode, grid, block, 0, stream, x, x_seed_stride, alpha, B, in_dim, (unsigned char*)kp, kp_rt,
                kp_kt, (unsigned char*)ktp, ktp_rt, ktp_kt, flag);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_w1_split(const float* theta, const float* alpha, void* wp, int S, int N, int in_dim, int hid,
                                  int ldp, int wp_rt, int wp_kt, void* stream) {
  if (!theta || !alpha || !wp || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63) || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid <= 0) return RCMARL_ERR_ARG;
  if ((long)wp_rt * 128 < (long)N * hid || wp_kt * 32 < in_dim) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(in_dim, 32), rc_ceil_div(N * hid, 128), S), block(256);
  RCMARL_LAUNCH(k_w1_split, grid, block, 0, stream, theta, alpha, (unsigned char*)wp, N, in_dim, ldp, wp_rt, wp_kt, hid);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_lattice_pack_dz(const float* dz, void* dzp, int S, int N, int B, int hid, int ldb, int dzp_rt,
                                         int dzp_kt, void* stream) {
  if (!dz || !dzp || S <= 0 || N <= 0 || B <= 0 || hid <= 0 || ldb < B || (ldb & 3) ||
      (reinterpret_cast<uintptr_t>(dz) & 15))
    return RCMARL_ERR_ARG;
  const int rts = rc_ceil_div(N * hid, 128), kts = rc_ceil_div(B, 32);
  if (dzp_rt < rts || dzp_kt < kts) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_dz_pack, dim3(kts, rts, S), dim3(256), 0, stream, dz, (unsigned char*)dzp, N * hid, B, ldb, dzp_rt, dzp_kt);
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
  const int ns = lat_stages();
  static const int dbg = (getenv("RCMARL_LAT_SAMETILE") ? (atoi(getenv("RCMARL_LAT_SAMETILE")) & 0xff) : 0) |
                         (lat_env_int("RCMARL_LAT_NCHUNK", 0) << 8);
  const size_t smem = (size_t)(ns == 4 ? 2 : ns) * LatCfg<3, 1, 2, 4>::STAGE_BYTES;
  const dim3 grid((unsigned)lat_grid(S * mtiles * ntiles, ns)), block(256);
  static const int dbgm = getenv("RCMARL_LAT_DBG") ? atoi(getenv("RCMARL_LAT_DBG")) : 0;
  if (ns == 2 && dbgm != 0) {
#define RC_DBG_CASE(M)                                                                                               \
    if (dbgm == M) {                                                                                                 \
      if (!lat_want_lds(k_lat_forward<2, M>, smem)) return RCMARL_ERR_LAUNCH;                                        \
      RCMARL_LAUNCH((k_lat_forward<2, M>), grid, block, smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,        \
                    (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, \
                    lat_stagger_bit(), lat_stagger_n(), hid);                                                        \
    }
    RC_DBG_CASE(1) RC_DBG_CASE(2) RC_DBG_CASE(3) RC_DBG_CASE(4) RC_DBG_CASE(5) RC_DBG_CASE(6) RC_DBG_CASE(7)
    RC_DBG_CASE(8) RC_DBG_CASE(9) RC_DBG_CASE(15)
#undef RC_DBG_CASE
    return rcmarl_check_launch();
  }
  if (ns == 2 && (lat_spread() & 1)) {
    static const bool ok = lat_want_lds(k_lat_forward<6, 0, true>, smem) && lat_want_lds(k_lat_forward<6>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    if (lat_w8(true)) {
      RCMARL_LAUNCH((k_lat_forward<6, 0, true>), grid, dim3(512), smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                    (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
    } else {
      RCMARL_LAUNCH((k_lat_forward<6>), grid, block, smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                    (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
    }
  } else if (ns == 2 && lat_w8(true)) {
    static const bool ok = lat_want_lds(k_lat_forward<2, 0, true>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_forward<2, 0, true>), grid, dim3(512), smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                  (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
  } else if (ns == 2) {
    static const bool ok = lat_want_lds(k_lat_forward<2>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_forward<2>), grid, block, smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                  (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
  } else if (ns == 4) {
    static const bool ok = lat_want_lds(k_lat_forward<4>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_forward<4>), grid, block, smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                  (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
  } else {
    static const bool ok = lat_want_lds(k_lat_forward<3>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_forward<3>), grid, block, smem, stream, (const unsigned char*)wp, wp_rt, wp_kt,
                  (const unsigned char*)kp, kp_rt, kp_kt, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, dbg, lat_stagger_bit(), lat_stagger_n(), hid);
  }
  return rcmarl_check_launch();
}

static int backward_sgd_lattice(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt,
                                                     int dzp_kt, const float* alpha, float* theta, const int* mask,
                                                     int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                                     void* wp_out, int wp_rt, int wp_kt, void* stream, int wp_fit) {
  if (!ktp || !dzp || !alpha || !theta || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || (ldp & 63) ||
      ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if (hid <= 0) return RCMARL_ERR_ARG;
  const int mtiles = rc_ceil_div(in_dim, 256), ntiles = rc_ceil_div(N * hid, 128), ktiles = rc_ceil_div(B, 32);
  if (ktp_rt < 2 * mtiles || dzp_rt < ntiles || ktp_kt < ktiles || dzp_kt < ktiles) return RCMARL_ERR_ARG;
  if (wp_fit && hid != 20) return RCMARL_ERR_UNSUPPORTED;
  if (wp_out && ((long)wp_rt * 128 < (wp_fit ? (long)rc_ceil_div(N, RC_FIT_AGENTS) * RC_FIT_ROWS : (long)ntiles * 128) ||
                 wp_kt < rc_ceil_div(in_dim, 32)))
    return RCMARL_ERR_ARG;
  const int ns = lat_stages();
  const size_t smem = (size_t)(ns == 4 ? 2 : ns) * LatCfg<1, 3, 4, 2>::STAGE_BYTES;
  const dim3 grid((unsigned)lat_grid(S * mtiles * ntiles, ns)), block(256);
  static const int dbgm = getenv("RCMARL_LAT_DBG") ? atoi(getenv("RCMARL_LAT_DBG")) : 0;
  if (ns == 2 && dbgm != 0 && dbgm <= 8) {
#define RC_DBG_CASE(M)                                                                                               \
    if (dbgm == M) {                                                                                                 \
      if (!lat_want_lds(k_lat_backward_sgd<2, M>, smem)) return RCMARL_ERR_LAUNCH;                                   \
      RCMARL_LAUNCH((k_lat_backward_sgd<2, M>), grid, block, smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt, \
                    (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles,  \
                    ntiles, (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);          \
    }
    RC_DBG_CASE(1) RC_DBG_CASE(2) RC_DBG_CASE(3) RC_DBG_CASE(7) RC_DBG_CASE(8)
#undef RC_DBG_CASE
    return rcmarl_check_launch();
  }
  if (lat_bdirect()) {
    const size_t smem5 = 2 * 16 * 1024;                 // two stages of the one-piece operand only
    static const bool ok = lat_want_lds(k_lat_backward_sgd<5>, smem5);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<5>), dim3((unsigned)(S * mtiles * ntiles)), block, smem5, stream, (const unsigned char*)ktp,
                  ktp_rt, ktp_kt, (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles,
                  ntiles, (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  } else if (ns == 2 && lat_env_int("RCMARL_LAT_WIDE", 0) != 0 && dzp_rt >= 2 * rc_ceil_div(N * hid, 256)) {
    const int ntiles2 = rc_ceil_div(N * hid, 256);
    const size_t smem7 = 2 * LatCfg<1, 3, 4, 2, 2, 4>::STAGE_BYTES;
    static const bool ok = lat_want_lds(k_lat_backward_sgd<7>, smem7);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<7>), dim3((unsigned)(S * mtiles * ntiles2)), dim3(512), smem7, stream,
                  (const unsigned char*)ktp, ktp_rt, ktp_kt, (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B,
                  in_dim, ldp, lr, mtiles, ntiles2, (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid,
                  wp_fit);
  } else if (ns == 2 && lat_env_int("RCMARL_LAT_TALL", 0) != 0 && ktp_rt >= 4 * rc_ceil_div(in_dim, 512)) {
    const int mtiles4 = rc_ceil_div(in_dim, 512);
    const size_t smem8 = 2 * LatCfg<1, 3, 4, 2, 4, 2>::STAGE_BYTES;
    static const bool ok = lat_want_lds(k_lat_backward_sgd<8>, smem8);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<8>), dim3((unsigned)(S * mtiles4 * ntiles)), dim3(512), smem8, stream,
                  (const unsigned char*)ktp, ktp_rt, ktp_kt, (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B,
                  in_dim, ldp, lr, mtiles4, ntiles, (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid,
                  wp_fit);
  } else if (ns == 2 && lat_w8(false)) {
    static const bool ok = lat_want_lds(k_lat_backward_sgd<2, 0, true>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<2, 0, true>), grid, dim3(512), smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt,
                  (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,
                  (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  } else if (ns == 2 && (lat_spread() & 2)) {
    static const bool ok = lat_want_lds(k_lat_backward_sgd<6>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<6>), grid, block, smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt,
                  (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,
                  (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  } else if (ns == 2) {
    static const bool ok = lat_want_lds(k_lat_backward_sgd<2>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<2>), grid, block, smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt,
                  (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,
                  (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  } else if (ns == 4) {
    static const bool ok = lat_want_lds(k_lat_backward_sgd<4>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<4>), grid, block, smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt,
                  (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,
                  (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  } else {
    static const bool ok = lat_want_lds(k_lat_backward_sgd<3>, smem);
    if (!ok) return RCMARL_ERR_LAUNCH;
    RCMARL_LAUNCH((k_lat_backward_sgd<3>), grid, block, smem, stream, (const unsigned char*)ktp, ktp_rt, ktp_kt,
                  (const unsigned char*)dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, ldp, lr, mtiles, ntiles,
                  (unsigned char*)wp_out, wp_rt, wp_kt, lat_stagger_bit(), lat_stagger_n(), hid, wp_fit);
  }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_layer1_backward_sgd_lattice(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt,
                                                     int dzp_kt, const float* alpha, float* theta, const int* mask,
                                                     int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                                     void* wp_out, int wp_rt, int wp_kt, void* stream) {
  return backward_sgd_lattice(ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, hid, ldp, lr, wp_out,
                              wp_rt, wp_kt, stream, 0);
}

// the same step; wp_out receives the split of the updated W1 in FIT ORDER (the A operand of rcmarl_fit_fused_lattice)
RCMARL_EXPORT int rcmarl_layer1_backward_sgd_lattice_fit(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt,
                                                         int dzp_kt, const float* alpha, float* theta, const int* mask,
                                                         int S, int N, int B, int in_dim, int hid, int ldp, float lr,
                                                         void* wpf_out, int wpf_rt, int wpf_kt, void* stream) {
  return backward_sgd_lattice(ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, hid, ldp, lr, wpf_out,
                              wpf_rt, wpf_kt, stream, 1);
}
