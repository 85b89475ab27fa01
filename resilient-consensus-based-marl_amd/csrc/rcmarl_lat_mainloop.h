// The k-loop of the packed-operand ("PK", rcmarl_lattice.h) matrix-core GEMMs: two LDS stages filled by LDS-DMA, one barrier per
// k-tile, v_mfma_f32_32x32x16_{f16,bf16}.  Shared by the layer-1 lattice GEMMs (lattice_gemm.hip) and the dense layers of wide
// networks on pre-split operands (dense_pk.hip).  Include inside the translation unit's anonymous namespace users.
#pragma once
#include "rcmarl_lattice.h"

namespace {

__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st_u4(unsigned char* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// WM x WN wavefronts per workgroup, each owning MT x NT accumulator blocks of 32 x 32
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
};

// The k-loop: two LDS stages filled by LDS-DMA, one barrier per k-tile.
// SPREAD: the LDS-DMA bursts of the next k-tile are issued one at a time BETWEEN the matrix-core instructions of the first
// half of this k-tile instead of back to back right after the barrier (an LDS-DMA instruction blocks the wavefront's issue
// for 60-180 cycles; in a burst those add up while no MFMA of this wavefront is in flight).  Measured: -2..-4 % on the
// backward, +1..+8 % on the forward; bit-identical either way.
// (Round 2 built and measured, then round 3 removed: a 3-stage ring, a ring of four half-stages with counted vmcnt, the
// three-piece operand's fragments loaded global -> registers, 256 x 256 and 512 x 128 tiles with eight wavefronts,
// persistent workgroups, start staggers, static priorities, L2 prefetch touches -- all within -15..+0 % of this form;
// DESIGN.md section 5 keeps the numbers.  Round 5, on the forward (profiles/r05g_*): the second half of the wavefronts requesting
// the next stage BEHIND its matrix work instead of in front of it +5..+9 %, static priority for that half +4..+6 %.)
// DROP_LL (dense_pk.hip: both operands fp32-valued, two f16 pieces each): the product of the two LOW pieces (2^-22 of the result) is
// not formed -- three matrix passes per fp32 product instead of four.
template <int PA, int PB, int MT, int NT, int WM, int WN, bool SPREAD, bool F16, bool DROP_LL = false>
__device__ __forceinline__ void lat_mainloop(const LatOperands& op, int n_ktiles, unsigned char* lds,
                                             rc_f32x16 (&acc)[MT][NT]) {
  typedef LatCfg<PA, PB, MT, NT, WM, WN> C;
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
  static_assert(!DROP_LL || (PA == 2 && PB == 2 && !SPREAD), "DROP_LL: two pieces on both sides, plain issue order");
  constexpr int N_MFMA = 2 * PA * PB * MT * NT;                 // matrix-core instructions of a wavefront per k-tile
  constexpr int EVERY = (N_MFMA / 2) / C::GLDS > 0 ? (N_MFMA / 2) / C::GLDS : 1;     // all bursts within the first half
  static_assert(!SPREAD || EVERY * C::GLDS <= N_MFMA, "spread issue: every burst has a slot");

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

  stage(0, 0);
  for (int t = 0; t < n_ktiles; ++t) {
    const int cur = t & 1;
    RC_WAIT_VMEM();                 // this wavefront's bursts of tile t have landed ...
    __syncthreads();                // ... and everybody's; all reads of the buffer refilled next are done
    const bool more = t + 1 < n_ktiles;
    if (more && !SPREAD) stage(cur ^ 1, t + 1);
    const unsigned char* st = lds + cur * C::STAGE_BYTES;
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
        for (int pb = PB - 1; pb >= 0; --pb) {
          if (DROP_LL && pa == PA - 1 && pb == PB - 1) continue;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = F16 ? rc_mfma_f16(af[mt][pa], bf[nt][pb], acc[mt][nt]) : rc_mfma_bf16(af[mt][pa], bf[nt][pb], acc[mt][nt]);
              if constexpr (SPREAD) {
                const int o = ks * (PA * PB * MT * NT) + (((PA - 1 - pa) * PB + (PB - 1 - pb)) * MT + mt) * NT + nt;
                if (o % EVERY == EVERY - 1 && o / EVERY < C::GLDS) {
                  RC_SCHED_FENCE();
                  if (more) stage_one(cur ^ 1, t + 1, o / EVERY);
                  RC_SCHED_FENCE();
                }
              }
            }
        }
    }
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

// explicit work-group size + waves per SIMD (with __launch_bounds__(512, 2) hipcc allots 129 registers to an eight-wavefront
// form, one too many for the four wavefronts per SIMD that two such workgroups per CU need)
#ifdef RCMARL_EMU
#define RC_LAT_OCC(threads, waves)
#else
#define RC_LAT_OCC(threads, waves) __attribute__((amdgpu_flat_work_group_size(threads, threads), amdgpu_waves_per_eu(waves)))
#endif


// ---------------------------------------------------------------------------------------------
// Epilogue helpers: one 32 x 32 accumulator block of a wavefront (register r of lane (l31, half) = row 8 (r >> 2) + (r & 3) + 4 half,
// column l31 -- the D layout of the 32x32 MFMAs) written as PACKED 16-bit operand rows for the next GEMM, in either orientation.
//
// (1) PK rows = the block's COLUMNS, reduction index = its ROWS: a lane holds four consecutive reduction indices per register group,
// the two lanes of a pair (lane, lane ^ 32) exchange halves (one v_permlane32_swap per dword) and each stores whole 16-byte chunks
// -- the form of k_lat_backward_sgd's next-step operand.  pc[p][q][dw]: piece p of rows 8q + 4 half + (2 dw, 2 dw + 1), low / high
// half-word.  rowp: the lane's 64-byte row inside piece 0 of the (row-tile, k-tile) block; sw = (row >> 2) & 3.
template <int NP>
__device__ __forceinline__ void pk_emit_rows_from_lanes(unsigned (&pc)[NP][4][2], unsigned char* rowp, int sw, bool ok) {
  const int half = (threadIdx.x >> 5) & 1;
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) rc_swap_halves(pc[p][2 * gp][dw], pc[p][2 * gp + 1][dw]);
    if (ok) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint4 v;
        v.x = pc[p][2 * gp][0]; v.y = pc[p][2 * gp][1]; v.z = pc[p][2 * gp + 1][0]; v.w = pc[p][2 * gp + 1][1];
        st_u4(rowp + p * RC_PK_BLOCK + (((2 * gp + half) ^ sw) << 4), v);
      }
    }
  }
}
// (2) PK rows = the block's ROWS, reduction index = its COLUMNS (the lanes): transposed through a wavefront-private LDS scratch of
// NP x 2 KiB ([32 rows][32 columns] 16-bit per piece: 16 two-byte writes per piece, then every lane reads two whole 16-byte chunks per
// piece and stores them -- 16 rows x 64 B = whole cache lines per store instruction).  h16[p][r]: piece p of register r.
// blk: piece 0 of the (row-tile, k-tile) block; row0: the block's first row inside its 128-row tile.
template <int NP>
__device__ __forceinline__ void pk_emit_rows_from_regs(const unsigned short (&h16)[NP][16], unsigned char* scratch, unsigned char* blk,
                                                       int row0) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      reinterpret_cast<unsigned short*>(scratch + p * 2048)[(8 * (r >> 2) + (r & 3) + 4 * half) * 32 + l31] = h16[p][r];
  RC_WAVE_SYNC();
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = lane + 64 * i, row = idx >> 2, c = idx & 3, R = row0 + row;
      st_u4(blk + p * RC_PK_BLOCK + R * 64 + ((c ^ ((R >> 2) & 3)) << 4), ld_u4(scratch + p * 2048 + row * 64 + c * 16));
    }
  RC_WAVE_SYNC();
}

}  // namespace
