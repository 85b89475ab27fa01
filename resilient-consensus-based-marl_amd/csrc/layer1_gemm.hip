// K4/K5 (layer 1) -- the two dense GEMMs of the batched-over-agents MLP.
//
// Every agent of a seed sees the SAME network input (global state / global
// state-action, reference training/train_agents.py:89-93), so the first Dense
// layer of all N agents is one genuine GEMM per seed
//     Z1[b, (n,j)] = sum_k X[b,k] * W1_n[k,j]          (forward,  Keras Dense: agents/...:66,114,181)
//     dW1_n[k,j]   = sum_b X[b,k] * dZ1[b,(n,j)]       (backward, inside critic.fit/TR.fit :118,136)
// with M x N x K = (N*hid) x B x in  and  in x (N*hid) x B.  They carry ~94 % of the
// FLOPs of a local fit at N=256.  Both run on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, bit-equal to an fmaf chain), LDS-tiled
// 128x128x32 per 256-thread workgroup, next k-tile prefetched into registers (4 wavefronts, each 64x64 = 2x2 MFMA tiles).
//
// Layouts: X[S][rows][in] row-major (replay buffer), theta[S][N][ldp] parameter
// rows, activations FEATURE-MAJOR a1t[S][N*hid][ldb] (contiguous over b).
#include "rcmarl_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA = BM + 4;   // A tile [BK][LDA]; +4 keeps 16-B row alignment for ds_write_b128
constexpr int LDB = BN + 1;   // B tile [BK][LDB]; 4*LDB mod 32 == 4 -> conflict-free transposing writes
constexpr int A_REGS = BK * BM / 256;   // 16 floats of the next A tile per thread
constexpr int B_REGS = BK * BN / 256;   // 16 floats of the next B tile per thread

__device__ __forceinline__ rc_f32x16 zero16() {
  rc_f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Every loader is split in two halves so the k-loop can software-pipeline:
//   fetch(m0|n0, k0, regs)  : global -> registers (issued before the MFMA phase of the previous tile)
//   commit(regs, tile)      : registers -> LDS     (after the barrier that retires the previous tile)

// B operand, shared by both GEMMs: source is [n][k] with k contiguous (X[b][k] for the
// forward, dz1t[col][b] for the backward); the tile lands transposed as Bs[k][n].
struct LoadB_KContig {
  const float* src; long ld; int n_lim, k_lim; bool vec_ok;
  __device__ __forceinline__ void fetch(int n0, int k0, float (&r)[B_REGS]) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q;     // 0..1023 : 128 n x 8 float4
      const int n = idx >> 3, kq = idx & 7;
      const int gn = n0 + n, gk = k0 + 4 * kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gn < n_lim) {
        const float* p = src + (long)gn * ld + gk;
        if (vec_ok && gk + 3 < k_lim) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk + 0 < k_lim) v.x = p[0];
          if (gk + 1 < k_lim) v.y = p[1];
          if (gk + 2 < k_lim) v.z = p[2];
          if (gk + 3 < k_lim) v.w = p[3];
        }
      }
      r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
  }
  __device__ __forceinline__ void commit(const float (&r)[B_REGS], float* Bs) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q;
      const int n = idx >> 3, kq = idx & 7;
      float* d = Bs + (4 * kq) * LDB + n;
      d[0] = r[4 * q]; d[LDB] = r[4 * q + 1]; d[2 * LDB] = r[4 * q + 2]; d[3 * LDB] = r[4 * q + 3];
    }
  }
};

// forward A operand: A(m = column (agent, j), k) = theta[agent][k*hid + j]
struct LoadA_W1 {
  const float* theta_s; int ldp, hid, ncols, in_dim;
  __device__ __forceinline__ void fetch(int m0, int k0, float (&r)[A_REGS]) const {
    const int m = threadIdx.x & 127;
    const int col = m0 + m;
    const bool ok = col < ncols;
    const int agent = ok ? col / hid : 0;
    const int j = col - agent * hid;
    const float* base = theta_s + (long)agent * ldp + j + (long)(k0 + (threadIdx.x >> 7)) * hid;
    const int kbase = k0 + (threadIdx.x >> 7);
#pragma unroll
    for (int q = 0; q < A_REGS; ++q) r[q] = (ok && kbase + 2 * q < in_dim) ? base[(long)(2 * q) * hid] : 0.f;
  }
  __device__ __forceinline__ void commit(const float (&r)[A_REGS], float* As) const {
    const int m = threadIdx.x & 127;
#pragma unroll
    for (int q = 0; q < A_REGS; ++q) As[((threadIdx.x >> 7) + 2 * q) * LDA + m] = r[q];
  }
};

// backward A operand: A(m = input feature, k = b) = X[b][m]  (m contiguous)
struct LoadA_XT {
  const float* x_s; int in_dim, B; bool vec_ok;
  __device__ __forceinline__ void fetch(int m0, int k0, float (&r)[A_REGS]) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q;   // 0..1023 : 32 rows x 32 float4
      const int k = idx >> 5, mq = idx & 31;
      const int gb = k0 + k, gm = m0 + 4 * mq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gb < B) {
        const float* p = x_s + (long)gb * in_dim + gm;
        if (vec_ok && gm + 3 < in_dim) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gm + 0 < in_dim) v.x = p[0];
          if (gm + 1 < in_dim) v.y = p[1];
          if (gm + 2 < in_dim) v.z = p[2];
          if (gm + 3 < in_dim) v.w = p[3];
        }
      }
      r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
  }
  __device__ __forceinline__ void commit(const float (&r)[A_REGS], float* As) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q;
      const int k = idx >> 5, mq = idx & 31;
      *reinterpret_cast<float4*>(As + k * LDA + 4 * mq) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
    }
  }
};

// 128x128 output tile, K swept in steps of 32 with register prefetch of the next tile.
// Calls epi(m, n, value) for every accumulator element (global indices; the functor bounds-checks).
template <class LA, class LB, class Epi>
__device__ __forceinline__ void gemm_tile(const LA& la, const LB& lb, const Epi& epi, int m0, int n0, int K) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  rc_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = zero16();
  float ra[A_REGS], rb[B_REGS];
  la.fetch(m0, 0, ra);
  lb.fetch(n0, 0, rb);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();                 // previous tile fully consumed
    la.commit(ra, As);
    lb.commit(rb, Bs);
    __syncthreads();
    if (k0 + BK < K) {               // next tile's HBM/L2 latency hides behind this tile's MFMAs
      la.fetch(m0, k0 + BK, ra);
      lb.fetch(n0, k0 + BK, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a0 = As[k * LDA + wm0 + (lane & 31)];
      const float a1 = As[k * LDA + wm0 + 32 + (lane & 31)];
      const float b0 = Bs[k * LDB + wn0 + (lane & 31)];
      const float b1 = Bs[k * LDB + wn0 + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      typename Epi::RowCtx ctx = epi.row(m);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) epi(ctx, m, n0 + wn0 + 32 * nt + (lane & 31), acc[mt][nt][r]);
    }
}

// ---- forward: a1t[col][b] = lrelu(sum_k W1[k][col] * X[b][k] + b1[col]) ---------------
struct EpiForward {
  const float* theta_s; float* a1t_s; int ldp, hid, ncols, B, ldb, o_b1;
  struct RowCtx { float bias; bool ok; };
  __device__ __forceinline__ RowCtx row(int m) const {      // m = column (agent, j): one bias per row
    RowCtx c;
    c.ok = m < ncols;
    const int agent = c.ok ? m / hid : 0;
    c.bias = c.ok ? theta_s[(long)agent * ldp + o_b1 + (m - agent * hid)] : 0.f;
    return c;
  }
  __device__ __forceinline__ void operator()(const RowCtx& c, int m, int n, float v) const {
    if (c.ok && n < B) a1t_s[(long)m * ldb + n] = rc_lrelu(v + c.bias);
  }
};

__global__ __launch_bounds__(256) void k_layer1_forward(const float* __restrict__ x, long x_seed_stride,
                                                        const float* __restrict__ theta, float* __restrict__ a1t,
                                                        int N, int B, int in_dim, int hid, int ldp, int ldb) {
  const int s = blockIdx.z;
  const int ncols = N * hid;
  const float* theta_s = theta + (long)s * N * ldp;
  const float* x_s = x + (long)s * x_seed_stride;
  float* a1t_s = a1t + (long)s * ncols * ldb;
  const bool vec_ok = (in_dim & 3) == 0 && ((x_seed_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  LoadA_W1 la{theta_s, ldp, hid, ncols, in_dim};
  LoadB_KContig lb{x_s, (long)in_dim, B, in_dim, vec_ok};
  EpiForward epi{theta_s, a1t_s, ldp, hid, ncols, B, ldb, in_dim * hid};
  gemm_tile(la, lb, epi, blockIdx.y * BM, blockIdx.x * BN, in_dim);
}

// ---- backward: W1[k][col] <- W1[k][col] - lr * sum_b X[b][k] * dz1t[col][b] ------------
struct EpiSgd {
  float* theta_s; const int* mask; int ldp, hid, ncols, in_dim; float lr;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row(int) const { return RowCtx(); }
  __device__ __forceinline__ void operator()(const RowCtx&, int m, int n, float g) const {
    if (m < in_dim && n < ncols) {
      const int agent = n / hid, j = n - agent * hid;
      if (mask == nullptr || mask[agent]) {
        float* w = theta_s + (long)agent * ldp + (long)m * hid + j;
        *w = *w - lr * g;
      }
    }
  }
};

// TF2 ResourceApplyAdam: m += (g-m)(1-b1); v += (g*g-v)(1-b2); w -= alpha*m/(sqrt(v)+eps)
struct EpiAdam {
  float* theta_s; float* m_s; float* v_s; const int* mask; int ldp, hid, ncols, in_dim;
  float alpha, one_m_b1, one_m_b2, eps;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row(int) const { return RowCtx(); }
  __device__ __forceinline__ void operator()(const RowCtx&, int m, int n, float g) const {
    if (m < in_dim && n < ncols) {
      const int agent = n / hid, j = n - agent * hid;
      if (mask == nullptr || mask[agent]) {
        const long o = (long)agent * ldp + (long)m * hid + j;
        float mm = m_s[o], vv = v_s[o];
        mm += (g - mm) * one_m_b1;
        vv += (g * g - vv) * one_m_b2;
        m_s[o] = mm; v_s[o] = vv;
        theta_s[o] = theta_s[o] - (mm * alpha) / (sqrtf(vv) + eps);
      }
    }
  }
};

__global__ __launch_bounds__(256) void k_layer1_backward_sgd(const float* __restrict__ x, long x_seed_stride,
                                                             const float* __restrict__ dz1t, float* __restrict__ theta,
                                                             const int* __restrict__ mask, int N, int B,
                                                             int in_dim, int hid, int ldp, int ldb, float lr) {
  const int s = blockIdx.z;
  const int ncols = N * hid;
  const float* x_s = x + (long)s * x_seed_stride;
  const bool vec_ok = (in_dim & 3) == 0 && ((x_seed_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  LoadA_XT la{x_s, in_dim, B, vec_ok};
  LoadB_KContig lb{dz1t + (long)s * ncols * ldb, (long)ldb, ncols, B, true};
  EpiSgd epi{theta + (long)s * N * ldp, mask, ldp, hid, ncols, in_dim, lr};
  gemm_tile(la, lb, epi, blockIdx.y * BM, blockIdx.x * BN, B);
}

__global__ __launch_bounds__(256) void k_layer1_backward_adam(const float* __restrict__ x, long x_seed_stride,
                                                              const float* __restrict__ dz1t,
                                                              float* __restrict__ theta, float* __restrict__ adam_m,
                                                              float* __restrict__ adam_v,
                                                              const int* __restrict__ mask, int N, int B,
                                                              int in_dim, int hid, int ldp, int ldb, float alpha,
                                                              float one_m_b1, float one_m_b2, float eps) {
  const int s = blockIdx.z;
  const int ncols = N * hid;
  const float* x_s = x + (long)s * x_seed_stride;
  const bool vec_ok = (in_dim & 3) == 0 && ((x_seed_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  LoadA_XT la{x_s, in_dim, B, vec_ok};
  LoadB_KContig lb{dz1t + (long)s * ncols * ldb, (long)ldb, ncols, B, true};
  const long so = (long)s * N * ldp;
  EpiAdam epi{theta + so, adam_m + so, adam_v + so, mask, ldp, hid, ncols, in_dim, alpha, one_m_b1, one_m_b2, eps};
  gemm_tile(la, lb, epi, blockIdx.y * BM, blockIdx.x * BN, B);
}

#include "layer1_fast.inc"
#include "layer1_small.inc"

// RCMARL_L1_SMALL=0 keeps networks with <= 32 inputs on the generic GEMM kernels (bisecting knob)
bool small_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RCMARL_L1_SMALL"); v = e ? atoi(e) : 1; }
  return v != 0;
}

// 0 = generic kernels only, 1 = fast path (default).  (A double-buffered LDS form of the fast path measured slower.)
// RCMARL_GEMM is a tuning/bisecting knob, read once.
int gemm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RCMARL_GEMM");
    v = e ? atoi(e) : 1;
    if (v < 0 || v > 1) v = 1;
  }
  return v;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class K>
bool want_lds(K kernel, size_t smem) {
  return rc_want_lds(kernel, smem, 64 * 1024);
}

bool bad_common(const void* a, const void* b, const void* c, int S, int N, int B, int in_dim, int hid, int ldp,
                int ldb) {
  return !a || !b || !c || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || hid <= 0 || (ldp & 63) || (ldb & 63) ||
         ldb < B || ldp < in_dim * hid + hid;
}

}  // namespace

RCMARL_EXPORT int rcmarl_layer1_forward(const float* x, long x_seed_stride, const float* theta, float* a1t, int S,
                                        int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream) {
  if (bad_common(x, theta, a1t, S, N, B, in_dim, hid, ldp, ldb)) return RCMARL_ERR_ARG;
  const int var = gemm_variant();
  if (hid == 20 && in_dim <= 32 && small_enabled()) {
    // chunks per workgroup: all of them once S*N workgroups fill the chip several times over, else one (short rows x few nets)
    const int nchunk = rc_ceil_div(B, small::SROWS);
    int cpw = (long)S * N >= 2048 ? nchunk : ((long)S * N >= 512 ? (nchunk + 3) / 4 : 1);
    if (cpw > nchunk) cpw = nchunk;
    const dim3 grid(rc_ceil_div(nchunk, cpw), N, S), block(256);
    if (in_dim <= 16) {
      RCMARL_LAUNCH((small::k_fwd<16>), grid, block, 0, stream, x, x_seed_stride, theta, a1t, N, B, in_dim, ldp, ldb, cpw);
    } else {
      RCMARL_LAUNCH((small::k_fwd<32>), grid, block, 0, stream, x, x_seed_stride, theta, a1t, N, B, in_dim, ldp, ldb, cpw);
    }
    return rcmarl_check_launch();
  }
  if (var > 0 && hid == 20 && (in_dim % fast::FBK) == 0 && aligned16(x) && (x_seed_stride & 3) == 0) {
    const dim3 grid(rc_ceil_div(B, fast::FBN), rc_ceil_div(N * 20, 160), S), block(256);
    const size_t smem = fast::smem_fwd(var);
    RCMARL_LAUNCH((fast::k_fwd<1>), grid, block, smem, stream, x, x_seed_stride, theta, a1t, N, B, in_dim, ldp, ldb);
    return rcmarl_check_launch();
  }
  const dim3 grid(rc_ceil_div(B, BN), rc_ceil_div(N * hid, BM), S), block(256);
  RCMARL_LAUNCH(k_layer1_forward, grid, block, 0, stream, x, x_seed_stride, theta, a1t, N, B, in_dim, hid, ldp, ldb);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_layer1_backward_sgd(const float* x, long x_seed_stride, const float* dz1t, float* theta,
                                             const int* mask, int S, int N, int B, int in_dim, int hid,
                                             int ldp, int ldb, float lr, void* stream) {
  if (bad_common(x, dz1t, theta, S, N, B, in_dim, hid, ldp, ldb)) return RCMARL_ERR_ARG;
  const int var = gemm_variant();
  if (hid == 20 && in_dim <= 32 && small_enabled()) {
    const dim3 grid(N, S), block(256);
    if (in_dim <= 16) {
      RCMARL_LAUNCH((small::k_bwd_sgd<16>), grid, block, 0, stream, x, x_seed_stride, dz1t, theta, mask, N, B, in_dim, ldp,
                    ldb, lr);
    } else {
      RCMARL_LAUNCH((small::k_bwd_sgd<32>), grid, block, 0, stream, x, x_seed_stride, dz1t, theta, mask, N, B, in_dim, ldp,
                    ldb, lr);
    }
    return rcmarl_check_launch();
  }
  if (var > 0 && hid == 20 && (in_dim & 3) == 0 && aligned16(x) && (x_seed_stride & 3) == 0) {
    const dim3 grid(rc_ceil_div(N * 20, fast::FBN), rc_ceil_div(in_dim, 128), S), block(256);
    const size_t smem = fast::smem_bwd(var);
    const fast::ApplySgd ap{lr};
    RCMARL_LAUNCH((fast::k_bwd<1, fast::ApplySgd>), grid, block, smem, stream, x, x_seed_stride, dz1t, theta, mask, N, B,
                    in_dim, ldp, ldb, ap);
    return rcmarl_check_launch();
  }
  const dim3 grid(rc_ceil_div(N * hid, BN), rc_ceil_div(in_dim, BM), S), block(256);
  RCMARL_LAUNCH(k_layer1_backward_sgd, grid, block, 0, stream, x, x_seed_stride, dz1t, theta, mask, N, B, in_dim, hid,
                ldp, ldb, lr);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_layer1_backward_adam(const float* x, long x_seed_stride, const float* dz1t, float* theta,
                                              float* adam_m, float* adam_v, const int* mask, int S, int N,
                                              int B, int in_dim, int hid, int ldp, int ldb, float alpha,
                                              float one_m_b1, float one_m_b2, float eps, void* stream) {
  if (bad_common(x, dz1t, theta, S, N, B, in_dim, hid, ldp, ldb) || !adam_m || !adam_v) return RCMARL_ERR_ARG;
  const int var = gemm_variant();
  if (var > 0 && hid == 20 && (in_dim & 3) == 0 && aligned16(x) && (x_seed_stride & 3) == 0) {
    const dim3 grid(rc_ceil_div(N * 20, fast::FBN), rc_ceil_div(in_dim, 128), S), block(256);
    const size_t smem = fast::smem_bwd(var);
    const fast::ApplyAdam ap{adam_m, adam_v, alpha, one_m_b1, one_m_b2, eps};
    RCMARL_LAUNCH((fast::k_bwd<1, fast::ApplyAdam>), grid, block, smem, stream, x, x_seed_stride, dz1t, theta, mask, N, B,
                    in_dim, ldp, ldb, ap);
    return rcmarl_check_launch();
  }
  const dim3 grid(rc_ceil_div(N * hid, BN), rc_ceil_div(in_dim, BM), S), block(256);
  RCMARL_LAUNCH(k_layer1_backward_adam, grid, block, 0, stream, x, x_seed_stride, dz1t, theta, adam_m, adam_v, mask, N,
                B, in_dim, hid, ldp, ldb, alpha, one_m_b1, one_m_b2, eps);
  return rcmarl_check_launch();
}
