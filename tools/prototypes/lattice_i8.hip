// NOT PART OF THE PRODUCT LIBRARY (moved out of csrc/ in round 4): a measured-slower prototype kept for the record -- DESIGN.md section 5,
// Round 3 item 3; profiles/r03e_int8_limb_forward_prototype.txt.  It built against csrc/rcmarl_lattice.h and exported three entry points
// (rcmarl_lattice_encode_i8, rcmarl_w1_split_i8, rcmarl_layer1_forward_i8) that no longer exist in include/rcmarl.h.
// Layer-1 FORWARD GEMM on the int8 matrix core ("int8 limbs"), a measured PROTOTYPE of VERDICT r02 item 1(d).
//
// The lattice path (rcmarl_lattice.h) multiplies the small-integer replay operand K (|K| <= 127 here: grids up to 64 x 64)
// by alpha_k * W1 split into three bf16 pieces: 3 passes of v_mfma_f32_32x32x16_bf16.  The int8 matrix core issues
// v_mfma_i32_32x32x32_i8 at the same 32 cycles per instruction with TWICE the k (measured 3.53 POP/s against 1.77 PFLOP/s
// chip-wide, profiles/r03a_mfma_peak_bf16_int8_clocks.txt), so four int8 "limbs" of the weight cost 4/2 = 2 bf16-pass
// equivalents instead of 3 -- if the operands can be delivered.  Arithmetic:
//     W'[c][k] = alpha_k * W1[k][c]            (fp32, as the bf16x3 path)
//     E_c      = exponent with max_k |W'[c][k]| < 2^E_c                      (one scale per W1 column, over ALL inputs)
//     Q[c][k]  = rint(W'[c][k] * 2^(30 - E_c))                               (|Q| <= 2^30; exact for |W'| >= 2^(E_c - 7),
//                                                                             absolute error <= 2^(E_c - 31) below that)
//     Q        = L0 + 2^8 L1 + 2^16 L2 + 2^24 L3,   L0..L2 in [-128, 127], |L3| <= 64         (balanced base-256 digits)
//     acc_l[c][b] = sum_k L_l[c][k] * K[b][k]                                (int32, EXACT: |acc_l| < 2^25)
//     z1[c][b] = 2^(E_c - 30) * (acc0 + 2^8 acc1 + 2^16 acc2 + 2^24 acc3)    (three fp32 roundings in all)
// i.e. the dot product is exact up to the quantisation of the small weights -- no worse than an fp32 fmaf chain, whose
// every step rounds.  What it costs: FOUR accumulator sets (the bf16 pieces share one fp32 accumulator because the hardware
// aligns their exponents; int32 limbs cannot), i.e. 64 x 64 wavefront tiles at 256 accumulator registers, one wavefront per
// SIMD, 128 x 128 workgroup tiles -- the same operand bytes per useful flop as the bf16x3 kernel's 128 x 256 tiles.
//
// Packed int8 format ("PK8"): as the bf16 format with bytes for elements -- 8-KiB blocks [R/128][KT64][NP][128 rows][64 k],
// the four 16-byte chunks of a 64-byte row XOR-swizzled by (row>>2)&3; an MFMA fragment (lane: row = lane&31, 16 consecutive
// k = chunk 2*kstep + lane>>5) is one conflict-free ds_read_b128.
//
// Entry points (include/rcmarl.h): rcmarl_lattice_encode_i8, rcmarl_w1_split_i8, rcmarl_layer1_forward_i8.  NOT wired into the
// engine: tools/kbench.py i8 times them against the bf16x3 kernels (DESIGN.md section 5 has the verdict).
#include "rcmarl_lattice.h"
#include <type_traits>

namespace {

#ifdef RCMARL_EMU
typedef intx16 rc_i32x16;
__device__ __forceinline__ rc_i32x16 rc_mfma_i8(uint4 a, uint4 b, rc_i32x16 c) { return __hipemu_mfma_i32_32x32x32_i8(a, b, c); }
#else
typedef int rc_i32x16 __attribute__((ext_vector_type(16)));
typedef int rc_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rc_i32x16 rc_mfma_i8(uint4 a, uint4 b, rc_i32x16 c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(rc_i32x4, a), __builtin_bit_cast(rc_i32x4, b), c, 0, 0, 0);
}
#endif

__device__ __forceinline__ uint4 ld_u4(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }

// ---- encode: replay rows -> K as int8, rows = replay row, reduction = feature.  One thread = one row x 32 features.
__global__ __launch_bounds__(256) void k_encode_i8(const float* __restrict__ x, long x_seed_stride, const float* __restrict__ alpha,
                                                   int B, int in_dim, unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                                   int* __restrict__ flag) {
  const int s = blockIdx.z, rt = blockIdx.y, kt = blockIdx.x;
  const int t = threadIdx.x, r = t >> 1, hf = t & 1;
  const int b = rt * 128 + r;
  unsigned char* blk = kp + ((long)s * kp_rt * kp_kt + (long)rt * kp_kt + kt) * RC_PK_BLOCK;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ch = 2 * hf + q;                          // 16-byte chunk of the 64-byte row
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = kt * 64 + 16 * ch + e;
      int kq = 0;
      if (b < B && c < in_dim) {
        const float xv = x[(long)s * x_seed_stride + (long)b * in_dim + c], al = alpha[c];
        const float kf = rintf(xv / al);
        if (!(fabsf(kf) <= 127.f) || !(fabsf(fmaf(kf, al, -xv)) <= 4.76837158e-7f * fabsf(xv))) bad = true;
        kq = (int)kf;
      }
      w[e >> 2] |= ((unsigned)kq & 0xffu) << (8 * (e & 3));
    }
    uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    *reinterpret_cast<uint4*>(blk + r * 64 + ((ch ^ ((r >> 2) & 3)) << 4)) = v;
  }
  if (bad) *flag = 1;
}

// ---- W1 -> four int8 limbs per weight + one scale per column.  One workgroup = 128 (agent, unit) columns, ALL inputs:
// pass 1 finds the column maxima, pass 2 (the same global reads, L2-warm) quantises and writes the limb planes.
__global__ __launch_bounds__(256) void k_w1_split_i8(const float* __restrict__ theta, const float* __restrict__ alpha,
                                                     unsigned char* __restrict__ wp, float* __restrict__ scale, int N, int in_dim,
                                                     int ldp, int wp_rt, int wp_kt, int hid) {
  __shared__ float smax[256];
  const int s = blockIdx.y, rt = blockIdx.x;
  const int t = threadIdx.x, r = t & 127, kh = t >> 7;
  const int col = rt * 128 + r, ncols = N * hid;
  const bool col_ok = col < ncols;
  const int ag = col_ok ? col / hid : 0, j = col - ag * hid;
  const float* th = theta + ((long)s * N + ag) * ldp + j;
  float mx = 0.f;
  if (col_ok)
    for (int k = kh; k < in_dim; k += 2) mx = fmaxf(mx, fabsf(th[(long)k * hid] * alpha[k]));
  smax[t] = mx;
  __syncthreads();
  mx = fmaxf(smax[r], smax[r + 128]);
  // 2^E > mx (E = 0 for an all-zero column); q = 2^(30 - E), 1/q = the scale the GEMM multiplies back
  int e2 = 0;
  if (mx > 0.f) { (void)frexpf(mx, &e2); }             // mx = f * 2^e2, 0.5 <= f < 1
  const float q = ldexpf(1.f, 30 - e2);
  if (col_ok && kh == 0) scale[(long)s * wp_rt * 128 + col] = ldexpf(1.f, e2 - 30);
  unsigned char* base = wp + ((long)s * wp_rt * wp_kt + (long)rt * wp_kt) * 4 * RC_PK_BLOCK;
  for (int kt = 0; kt < wp_kt; ++kt) {
#pragma unroll
    for (int qc = 0; qc < 2; ++qc) {
      const int ch = 2 * kh + qc;
      unsigned l0[4] = {0, 0, 0, 0}, l1[4] = {0, 0, 0, 0}, l2[4] = {0, 0, 0, 0}, l3[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = kt * 64 + 16 * ch + e;
        int Q = 0;
        if (col_ok && k < in_dim) Q = (int)rintf((th[(long)k * hid] * alpha[k]) * q);
        // balanced digits: d = ((Q + 128) & 255) - 128, Q <- (Q - d) >> 8  (exact: Q - d is a multiple of 256)
        const int d0 = ((Q + 128) & 255) - 128; Q = (Q - d0) >> 8;
        const int d1 = ((Q + 128) & 255) - 128; Q = (Q - d1) >> 8;
        const int d2 = ((Q + 128) & 255) - 128; Q = (Q - d2) >> 8;
        const unsigned sh = 8 * (e & 3);
        l0[e >> 2] |= ((unsigned)d0 & 0xffu) << sh; l1[e >> 2] |= ((unsigned)d1 & 0xffu) << sh;
        l2[e >> 2] |= ((unsigned)d2 & 0xffu) << sh; l3[e >> 2] |= ((unsigned)Q & 0xffu) << sh;
      }
      unsigned char* p = base + (long)kt * 4 * RC_PK_BLOCK + r * 64 + ((ch ^ ((r >> 2) & 3)) << 4);
      uint4 v;
      v.x = l0[0]; v.y = l0[1]; v.z = l0[2]; v.w = l0[3]; *reinterpret_cast<uint4*>(p) = v;
      v.x = l1[0]; v.y = l1[1]; v.z = l1[2]; v.w = l1[3]; *reinterpret_cast<uint4*>(p + RC_PK_BLOCK) = v;
      v.x = l2[0]; v.y = l2[1]; v.z = l2[2]; v.w = l2[3]; *reinterpret_cast<uint4*>(p + 2 * RC_PK_BLOCK) = v;
      v.x = l3[0]; v.y = l3[1]; v.z = l3[2]; v.w = l3[3]; *reinterpret_cast<uint4*>(p + 3 * RC_PK_BLOCK) = v;
    }
  }
}

// ---- forward GEMM: A = W limbs (rows = columns c of W1, 4 planes), B = K (rows = replay rows), k-tile 64
#ifdef RCMARL_EMU
#define RC_I8_OCC
#else
#define RC_I8_OCC __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1)))
#endif
__global__ RC_I8_OCC void k_lat_forward_i8(const unsigned char* __restrict__ wp, int wp_rt, int wp_kt,
                                           const unsigned char* __restrict__ kp, int kp_rt, int kp_kt,
                                           const float* __restrict__ scale, const float* __restrict__ theta,
                                           float* __restrict__ a1t, int S, int N, int B, int in_dim, int ldp, int ldb, int mtiles,
                                           int ntiles, int hid) {
  constexpr int NL = 4, MT = 2, NT = 2;                 // limbs; 32x32 blocks per wavefront (64 x 64), 2 x 2 wavefronts = 128 x 128
  constexpr int A_KB = NL * 8, B_KB = 8, STAGE_BYTES = (A_KB + B_KB) * 1024, GLDS = (A_KB + B_KB) / 4;
  RCMARL_DYN_SMEM(unsigned char, lds);
  const int per_seed = mtiles * ntiles;
  int s, w;
  if ((S & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; s = xcd + 8 * (q / per_seed); w = q % per_seed; }
  else { s = blockIdx.x / per_seed; w = blockIdx.x - s * per_seed; }
  const int bn = w % ntiles, bm = w / ntiles;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  rc_i32x16 acc[NL][MT][NT];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[l][mt][nt][q] = 0;
  // LDS-DMA bursts of this wavefront: 40 one-KiB bursts per stage, 10 per wavefront (A: 32 = 4 limb blocks of 8, B: 8)
  const unsigned char* a_base = wp + ((long)s * wp_rt * wp_kt + (long)bm * wp_kt) * NL * RC_PK_BLOCK;
  const unsigned char* b_base = kp + ((long)s * kp_rt * kp_kt + (long)bn * kp_kt) * RC_PK_BLOCK;
  const unsigned char* gsrc[GLDS];
  int gstep[GLDS];
#pragma unroll
  for (int i = 0; i < GLDS; ++i) {
    const int q = wave + 4 * i;
    if (q < A_KB) { gsrc[i] = a_base + q * 1024; gstep[i] = NL * RC_PK_BLOCK; }
    else { gsrc[i] = b_base + (q - A_KB) * 1024; gstep[i] = RC_PK_BLOCK; }
  }
  const unsigned lane16 = lane * 16;
  const rc_lds_t lds0 = rc_lds_addr(lds) + wave * 1024;
  auto stage = [&](int buf, int t) {
    const rc_lds_t dst = lds0 + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < GLDS; ++i) RC_GLDS16S(gsrc[i] + (long)t * gstep[i], lane16, dst + i * 4096);
  };
  const int sw = (l31 >> 2) & 3;
  const int co0 = ((0 + half) ^ sw) << 4, co1 = ((2 + half) ^ sw) << 4;
  int offA[MT], offB[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) offA[mt] = (wm * 64 + 32 * mt + l31) * 64;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) offB[nt] = A_KB * 1024 + (wn * 64 + 32 * nt + l31) * 64;
  const int n_ktiles = (in_dim + 63) >> 6;
  stage(0, 0);
  for (int t = 0; t < n_ktiles; ++t) {
    const int cur = t & 1;
    RC_WAIT_VMEM();
    __syncthreads();
    if (t + 1 < n_ktiles) stage(cur ^ 1, t + 1);
    const unsigned char* st = lds + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ks == 0 ? co0 : co1;
      uint4 af[NL][MT], bf[NT];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[l][mt] = ld_u4(st + l * RC_PK_BLOCK + offA[mt] + co);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = ld_u4(st + offB[nt] + co);
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[l][mt][nt] = rc_mfma_i8(af[l][mt], bf[nt], acc[l][mt][nt]);
    }
  }
  // epilogue: z = scale_c * (acc0 + 2^8 acc1 + 2^16 acc2 + 2^24 acc3) + b1;  a1t[c][b] = lrelu(z)
  const int ncols = N * hid;
  const float* theta_s = theta + (long)s * N * ldp;
  const float* scale_s = scale + (long)s * wp_rt * 128;
  float* a1t_s = a1t + (long)s * ncols * ldb;
  __syncthreads();
  float* bias = reinterpret_cast<float*>(lds);
  float* scl = bias + 128;
  if (threadIdx.x < 128) {
    const int col = bm * 128 + threadIdx.x;
    const int ag = col / hid;
    bias[threadIdx.x] = col < ncols ? theta_s[(long)ag * ldp + in_dim * hid + (col - ag * hid)] : 0.f;
    scl[threadIdx.x] = col < ncols ? scale_s[col] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = bn * 128 + wn * 64 + 32 * nt + l31;
    if (n < B) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m0 = bm * 128 + wm * 64 + 32 * mt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2) + 4 * half;
          // smallest limbs first; every partial sum is an integer times a power of two
          float v = (float)acc[0][mt][nt][r];
          v = fmaf((float)acc[1][mt][nt][r], 256.f, v);
          v = fmaf((float)acc[2][mt][nt][r], 65536.f, v);
          v = fmaf((float)acc[3][mt][nt][r], 16777216.f, v);
          const float z = fmaf(v, scl[wm * 64 + 32 * mt + dr], bias[wm * 64 + 32 * mt + dr]);
          if (m0 + dr < ncols) RC_NT_STORE(a1t_s + (long)(m0 + dr) * ldb + n, fmaxf(z, RC_LEAK * z));
        }
      }
    }
  }
}

}  // namespace

RCMARL_EXPORT int rcmarl_lattice_encode_i8(const float* x, long x_seed_stride, const float* alpha, int S, int B, int in_dim,
                                           void* kp, int kp_rt, int kp_kt, int* flag, void* stream) {
  if (!x || !alpha || !kp || !flag || S <= 0 || B <= 0 || in_dim <= 0) return RCMARL_ERR_ARG;
  if (kp_rt * 128 < B || kp_kt * 64 < in_dim) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_encode_i8, dim3(kp_kt, kp_rt, S), dim3(256), 0, stream, x, x_seed_stride, alpha, B, in_dim, (unsigned char*)kp,
                kp_rt, kp_kt, flag);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_w1_split_i8(const float* theta, const float* alpha, void* wp, float* scale, int S, int N, int in_dim,
                                     int hid, int ldp, int wp_rt, int wp_kt, void* stream) {
  if (!theta || !alpha || !wp || !scale || S <= 0 || N <= 0 || in_dim <= 0 || hid <= 0 || (ldp & 63) || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  if ((long)wp_rt * 128 < (long)N * hid || wp_kt * 64 < in_dim) return RCMARL_ERR_ARG;
  RCMARL_LAUNCH(k_w1_split_i8, dim3(wp_rt, S), dim3(256), 0, stream, theta, alpha, (unsigned char*)wp, scale, N, in_dim, ldp, wp_rt,
                wp_kt, hid);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_layer1_forward_i8(const void* kp, int kp_rt, int kp_kt, const void* wp, int wp_rt, int wp_kt,
                                           const float* scale, const float* theta, float* a1t, int S, int N, int B, int in_dim,
                                           int hid, int ldp, int ldb, void* stream) {
  if (!kp || !wp || !scale || !theta || !a1t || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || hid <= 0 || (ldp & 63) || (ldb & 63) ||
      ldb < B || ldp < in_dim * hid + hid)
    return RCMARL_ERR_ARG;
  const int mtiles = rc_ceil_div(N * hid, 128), ntiles = rc_ceil_div(B, 128), ktiles = rc_ceil_div(in_dim, 64);
  if (wp_rt < mtiles || kp_rt < ntiles || wp_kt < ktiles || kp_kt < ktiles) return RCMARL_ERR_ARG;
  const size_t smem = (size_t)2 * 40 * 1024;
  static const bool ok = rc_want_lds(k_lat_forward_i8, smem);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH(k_lat_forward_i8, dim3((unsigned)(S * mtiles * ntiles)), dim3(256), smem, stream, (const unsigned char*)wp, wp_rt,
                wp_kt, (const unsigned char*)kp, kp_rt, kp_kt, scale, theta, a1t, S, N, B, in_dim, ldp, ldb, mtiles, ntiles, hid);
  return rcmarl_check_launch();
}
