// NOT PART OF THE PRODUCT LIBRARY (round 4): csrc/wide_kernels.hip with k_wgemm16 as EIGHT wavefronts -- four consumers (fragments + MFMA),
// four producers (global fp32 -> f16 pieces -> LDS, loads three k-tiles ahead) -- and an XCD-aware workgroup mapping.  Measured at
// the cfg-5 layer-2 shapes (tools/kbench.py wide, 64 agents): 480 / 528 / 390 us against 453 / 532 / 390 us of the four-wavefront
// kernel that shipped: no gain.  SQ counters (profiles/r04y_sq_k_wgemm16_producer_consumer.json): matrix core 27 % busy, vector ALUs
// 35 %, LDS bank conflicts 0, wavefronts waiting 61 % of their cycles -- on operand delivery: 32 KiB of fp32 operands per 128 x 128 x 32
// tile step is 96 flop/byte, and the 3.2 GB a launch pulls through L2 arrive at ~6.5 TB/s whichever way the loads are issued.
// (only the k_wgemm16 section of the file is kept here; everything around it -- WArgs, w_loop_f32, w_epilogue, the entry points -- is
// csrc/wide_kernels.hip as committed)

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

template <bool KC>
__device__ __forceinline__ void w16_load(const float* __restrict__ P, int ld, int r0, int k0, int R, int K, float4 (&reg)[4]) {
  const int t = threadIdx.x & 255;
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
  const int t = threadIdx.x & 255;
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

// Eight wavefronts: 0-3 CONSUME (each a 64 x 64 part of the tile: fragments from LDS, 24 MFMAs per k-tile), 4-7 PRODUCE (global fp32
// -> registers -> f16 pieces -> LDS) with the loads of k-tile kt + 3 in flight while kt is consumed: a loader that shares the
// consumers' registers has room for one tile ahead only, and the k-loop then waits out a memory round trip per k-tile (the first
// cut of this kernel: 9.4k cycles per k-tile for 768 cycles of matrix work; profiles/r04y_*).
#ifdef RCMARL_EMU
#define RC_W16_OCC
#else
#define RC_W16_OCC __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(4)))
#endif
template <bool A_KC, bool B_KC, int EPI>
__global__ RC_W16_OCC void k_wgemm16(const WArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * W16_STAGE];
  __shared__ int s_ovf;
  static_assert(2 * W16_STAGE >= (int)(4 * WBK * WLD * sizeof(float)), "the fp32 loop's stages fit the same memory");
  // workgroup id -> ((seed, agent) z, m-tile, n-tile): all tiles of a (seed, agent) on ONE XCD (workgroups go round the eight XCDs by
  // id), m-tile fastest -- the workgroups that share a B panel run side by side and the agent's A matrix stays in that XCD's L2.
  // (x = n-tile fastest over all XCDs, the plain 3-D grid of the first cut, re-read the panels from beyond L2: 6.5 TB/s of operand
  // traffic at 27 % matrix-core occupancy, profiles/r04y_*.)
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
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool producer = wave >= 4;
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  rc_f16_saturate();
  if (threadIdx.x == 0) s_ovf = 0;
  const int nk = (a.K + W16_BK - 1) / W16_BK;
  rc_f32x16 acc[2][2];
  if (producer) {
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    float amax = 0.f;
    auto load = [&](float4 (&ra)[4], float4 (&rb)[4], int kt) {
      if (kt < nk) {
        w16_load<A_KC>(A, a.lda, m0, kt * W16_BK, a.M, a.K, ra);
        w16_load<B_KC>(Bp, a.ldb, n0, kt * W16_BK, a.N, a.K, rb);
      }
    };
    auto store = [&](const float4 (&ra)[4], const float4 (&rb)[4], int kt) {
      if (kt < nk) {
        unsigned char* __restrict__ st = sm + (kt & 1) * W16_STAGE;
        amax = w16_store<A_KC>(st, st + W16_PIECE, ra, a.sa, amax);
        amax = w16_store<B_KC>(st + 2 * W16_PIECE, st + 3 * W16_PIECE, rb, a.sb, amax);
      }
    };
    load(ra0, rb0, 0);
    load(ra1, rb1, 1);
    store(ra0, rb0, 0);
    load(ra0, rb0, 2);
    __syncthreads();                                            // stage 0 holds k-tile 0
    for (int kt = 0; kt < nk; kt += 2) {                        // consumers work on k-tile kt, then kt + 1
      store(ra1, rb1, kt + 1);
      load(ra1, rb1, kt + 3);
      __syncthreads();
      if (kt + 1 < nk) {
        store(ra0, rb0, kt + 2);
        load(ra0, rb0, kt + 4);
        __syncthreads();
      }
    }
    if (amax > RC_W16_RANGE) s_ovf = 1;                        // (NaN operands do not take this branch: they poison either loop alike)
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* __restrict__ st = sm + (kt & 1) * W16_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[j] = w16_frag<B_KC>(st + 2 * W16_PIECE, wn * 64 + 32 * j, ks);
          bl[j] = w16_frag<B_KC>(st + 3 * W16_PIECE, wn * 64 + 32 * j, ks);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 ah = w16_frag<A_KC>(st, wm * 64 + 32 * i, ks);
          const uint4 al = w16_frag<A_KC>(st + W16_PIECE, wm * 64 + 32 * i, ks);
          // smallest products first; consecutive MFMAs hit different accumulators
          acc[i][0] = rc_mfma_f16(al, bh[0], acc[i][0]);
          acc[i][1] = rc_mfma_f16(al, bh[1], acc[i][1]);
          acc[i][0] = rc_mfma_f16(ah, bl[0], acc[i][0]);
          acc[i][1] = rc_mfma_f16(ah, bl[1], acc[i][1]);
          acc[i][0] = rc_mfma_f16(ah, bh[0], acc[i][0]);
          acc[i][1] = rc_mfma_f16(ah, bh[1], acc[i][1]);
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  float unscale = 1.f / (a.sa * a.sb);
  if (s_ovf) {                                                // workgroup-uniform: out of the f16 range -> the fp32 loop, same tile
    __syncthreads();
    float* fa = reinterpret_cast<float*>(sm);
    w_loop_f32<A_KC, B_KC, true>(a, A, Bp, m0, n0, fa, fa + 2 * WBK * WLD, acc, !producer);
    unscale = 1.f;
  }
  if (!producer) w_epilogue<EPI>(a, s, ag, m0, n0, acc, unscale);
}

