// NOT PART OF THE PRODUCT LIBRARY (round 4): csrc/wide_kernels.hip with k_wgemm16 as 256 x 128 x 32 tiles, eight wavefronts, branch-free
// (clamped) loads two k-tiles ahead with counted waits, the split of the next tile interleaved with the matrix work by
// sched_group_barrier.  Measured (tools/kbench.py wide, 64 agents): 463 / 548 / 361 us against 453 / 532 / 390 us of the 128 x 128
// four-wavefront kernel that shipped -- the same, as were the plain 256 x 128 form (461 / 561 / 376) and its pipelined form without the
// interleave (452 / 551 / 376).  tools/_wide_probe-style scan of the forward (K = 32 ... 512): time = 130 us + 0.83 us per unit of K
// per launch, i.e. 4.6k cycles per k-tile and workgroup against 1.5k of matrix work, whatever the staging looks like.
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
// Tile: 256 (m) x 128 (n) x 32, EIGHT wavefronts of 64 x 64 (the first cut's 128 x 128 tile with four moved 32 KiB of fp32 operands
// per 3.1 MFLOP and was bound by operand delivery at 27 % matrix-core occupancy, section 5 Round 4 item 12; this one moves 48 KiB
// per 6.3 MFLOP).  Two 48-KiB stages (dynamic LDS), one workgroup per CU.
constexpr int W16_BM = 256, W16_BN = 128, W16_BK = 32, W16_NT = 512;
constexpr int W16_PA = W16_BM * W16_BK * 2, W16_PB = W16_BN * W16_BK * 2;      // bytes of one piece plane of A / of B, one stage
constexpr int W16_STAGE = 2 * W16_PA + 2 * W16_PB;
#define RC_W16_RANGE 65000.f

// global -> registers: this thread's float4s of a [ROWS x 32] operand tile.  BRANCH-FREE: a float4 outside the matrix is read from a
// clamped (valid) address and zeroed by its scale in w16_store -- a guarded load is a branch, and behind a branch hipcc's wait-count
// pass waits for ALL outstanding loads (vmcnt(0)) instead of the ones a store needs, which collapses any prefetch distance to one
// tile: the first cut of this kernel ran at bytes-in-flight / latency = 64 KiB per CU / ~3 us (section 5 Round 4 item 12).
template <bool KC, int ROWS>
__device__ __forceinline__ void w16_load(const float* __restrict__ P, int ld, int r0, int k0, int R, int K,
                                         float4 (&reg)[ROWS * 8 / W16_NT], unsigned& okmask) {
  const int t = threadIdx.x;
  okmask = 0u;
#pragma unroll
  for (int i = 0; i < ROWS * 8 / W16_NT; ++i) {
    const int idx = t + W16_NT * i;
    int r, k;
    if (KC) { r = r0 + (idx >> 3); k = k0 + (idx & 7) * 4; }
    else { k = k0 + idx / (ROWS / 4); r = r0 + (idx % (ROWS / 4)) * 4; }
    const bool ok = r < R && k < K;
    okmask |= ok ? (1u << i) : 0u;
    const int rc = min(r, R - (KC ? 1 : 4)), kc = min(k, K - (KC ? 4 : 1));        // (the caller guarantees whole float4s inside the matrix)
    // one (seed, agent) operand is < 2^32 bytes: uniform base + 32-bit lane offset, no 64-bit vector address arithmetic
    const unsigned off = KC ? (unsigned)rc * (unsigned)ld + (unsigned)kc : (unsigned)kc * (unsigned)ld + (unsigned)rc;
    reg[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(P) + 4u * off);
  }
}

__device__ __forceinline__ int w16_krow(int k, int w) { return k ^ (w & 3) ^ (4 * (w & 1)); }

// registers -> the two piece planes of one operand (ph, pl: ROWS * 64 bytes each); returns max |scaled value|
template <bool KC, int ROWS>
__device__ __forceinline__ float w16_store(unsigned char* __restrict__ ph, unsigned char* __restrict__ pl,
                                           const float4 (&reg)[ROWS * 8 / W16_NT], unsigned okmask, float scale, float amax) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < ROWS * 8 / W16_NT; ++i) {
    const int idx = t + W16_NT * i;
    const float sc = (okmask >> i) & 1u ? scale : 0.f;           // outside the matrix: zero (the value read is a finite matrix element)
    const float x0 = reg[i].x * sc, x1 = reg[i].y * sc, x2 = reg[i].z * sc, x3 = reg[i].w * sc;
    amax = rc_amax3(rc_amax3(amax, x0, x1), x2, x3);
    uint2 h, lo;
    rc_split2h_pair(x0, x1, h.x, lo.x);
    rc_split2h_pair(x2, x3, h.y, lo.y);
    int off;
    if (KC) {
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      off = r * 64 + (((k4 >> 3) ^ ((r >> 2) & 3)) << 4) + (k4 & 4) * 2;
    } else {
      const int k = idx / (ROWS / 4), r4 = (idx % (ROWS / 4)) * 4, w = r4 >> 4;
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

#ifdef RCMARL_EMU
#define RC_W16_OCC
#else
#define RC_W16_OCC __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(2)))
#endif
// Scheduling shape of one k-tile: the split + LDS write of the NEXT tile's registers is vector-ALU work, the matrix work of this tile
// runs on the matrix pipe -- of the SAME wavefronts, and all eight are in the same phase (one barrier per k-tile), so unless the two
// are interleaved instruction by instruction the pipes take turns (measured: 4.6k cycles per k-tile against 1.5k of matrix work).
// One MFMA, then up to seven vector / LDS instructions that fit under its 32 cycles.
#ifdef RCMARL_EMU
#define W16_INTERLEAVE() ((void)0)
#else
#define W16_INTERLEAVE()                                                                        \
  do {                                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) {                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* one MFMA */                        \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   /* up to two LDS reads (fragments) */ \
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   /* up to six vector ALU */            \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   /* up to one LDS write */             \
    }                                                                                            \
  } while (0)
#endif
template <bool A_KC, bool B_KC, int EPI>
__global__ RC_W16_OCC void k_wgemm16(const WArgs a) {
  RCMARL_DYN_SMEM(unsigned char, sm);                          // 2 * W16_STAGE bytes
  __shared__ int s_ovf;
  static_assert(2 * W16_STAGE >= (int)(2 * 4 * WBK * WLD * sizeof(float)), "the fp32 loop's stages (two half tiles) fit the same memory");
  const int z = blockIdx.z, s = z / a.NA, ag = z - s * a.NA;
  if (EPI == WEPI_SGD && a.mask && !a.mask[ag]) return;        // workgroup-uniform
  const int m0 = blockIdx.y * W16_BM, n0 = blockIdx.x * W16_BN;
  const float* __restrict__ A = a.A + s * a.A_zs + ag * a.A_za;
  const float* __restrict__ Bp = a.B + s * a.B_zs + ag * a.B_za;
  const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane((int)(t >> 6)), wm = w >> 1, wn = w & 1;
  rc_f16_saturate();
  if (t == 0) s_ovf = 0;
  rc_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Two register sets: the loads of k-tile kt + 2 are issued at the top of iteration kt and consumed (split + LDS write) at the end of
  // iteration kt + 1 -- two tiles (96 KiB per CU) in flight.  No branch anywhere in the loop: tiles past the last one are all-zero
  // operands (their float4s fail the bounds test), so an odd tile count simply runs one empty iteration.
  constexpr int NA4 = W16_BM * 8 / W16_NT, NB4 = W16_BN * 8 / W16_NT;
  float4 ra0[NA4], rb0[NB4], ra1[NA4], rb1[NB4];
  unsigned ma0, mb0, ma1, mb1;
  float amax = 0.f;
  const int nk = (a.K + W16_BK - 1) / W16_BK;
  auto load = [&](float4 (&ra)[NA4], float4 (&rb)[NB4], unsigned& ma, unsigned& mb, int kt) {
    w16_load<A_KC, W16_BM>(A, a.lda, m0, kt * W16_BK, a.M, a.K, ra, ma);
    w16_load<B_KC, W16_BN>(Bp, a.ldb, n0, kt * W16_BK, a.N, a.K, rb, mb);
  };
  auto store = [&](const float4 (&ra)[NA4], const float4 (&rb)[NB4], unsigned ma, unsigned mb, int kt) {
    unsigned char* __restrict__ st = sm + (kt & 1) * W16_STAGE;
    // the scales pass through an opaque (volatile) asm HERE: without it hipcc hoists the scaling and the first conversion of a
    // register set up to its loads -- across the matrix work -- and waits for the loads right where they were issued
    float sa = a.sa, sb = a.sb;
#ifndef RCMARL_EMU
    asm volatile("" : "+v"(sa), "+v"(sb));
#endif
    amax = w16_store<A_KC, W16_BM>(st, st + W16_PA, ra, ma, sa, amax);
    amax = w16_store<B_KC, W16_BN>(st + 2 * W16_PA, st + 2 * W16_PA + W16_PB, rb, mb, sb, amax);
  };
  auto compute = [&](int kt) {
    const unsigned char* __restrict__ st = sm + (kt & 1) * W16_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = w16_frag<A_KC>(st, wm * 64 + 32 * i, ks);
        al[i] = w16_frag<A_KC>(st + W16_PA, wm * 64 + 32 * i, ks);
        bh[i] = w16_frag<B_KC>(st + 2 * W16_PA, wn * 64 + 32 * i, ks);
        bl[i] = w16_frag<B_KC>(st + 2 * W16_PA + W16_PB, wn * 64 + 32 * i, ks);
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
  };
  load(ra0, rb0, ma0, mb0, 0);
  load(ra1, rb1, ma1, mb1, 1);
  store(ra0, rb0, ma0, mb0, 0);
  __syncthreads();
#pragma unroll 1
  for (int kt = 0; kt < nk; kt += 2) {
    load(ra0, rb0, ma0, mb0, kt + 2);
    RC_SCHED_FENCE();                                          // (hipcc sinks the loads below the matrix work otherwise: shorter live ranges, no prefetch)
    compute(kt);
    store(ra1, rb1, ma1, mb1, kt + 1);
    W16_INTERLEAVE();
    __syncthreads();
    load(ra1, rb1, ma1, mb1, kt + 3);
    RC_SCHED_FENCE();
    compute(kt + 1);
    store(ra0, rb0, ma0, mb0, kt + 2);
    W16_INTERLEAVE();
    __syncthreads();
  }
  if (amax > RC_W16_RANGE) s_ovf = 1;                         // (NaN operands do not take this branch: they poison either loop alike)
  __syncthreads();
  float unscale = 1.f / (a.sa * a.sb);
  const int mh = m0 + 128 * (w >> 2);                         // the 128-row half this wavefront's 64 x 64 block lies in
  if (s_ovf) {                                                // workgroup-uniform: out of the f16 range -> the fp32 loop, both half tiles side by side
    __syncthreads();
    float* fa = reinterpret_cast<float*>(sm) + (w >> 2) * (4 * WBK * WLD);
    w_loop_f32<A_KC, B_KC, true>(a, A, Bp, mh, n0, fa, fa + 2 * WBK * WLD, acc);
    unscale = 1.f;
  }
  w_epilogue<EPI>(a, s, ag, mh, n0, acc, unscale);
}

