// X1 -- the adversaries' mini-batch fits, one whole Keras `fit()` per launch.
//
// The non-cooperative agents of the reference train with *shuffled mini-batches*
//   Greedy_CAC_agent.critic_update_local / TR_update_local   fit(batch_size=32, epochs=10)
//       (agents/adversarial_CAC_agents.py:228-253)
//   Malicious_CAC_agent.critic_update_local / critic_update_compromised / TR_update_compromised
//       fit(batch_size=32, epochs=10)                         (:121-165)
//   {Faulty,Greedy,Malicious}_CAC_agent.actor_update         fit(batch_size=200, epochs=1), Adam
//       (:38-41, :111-117, :221-225)
// i.e. epochs*ceil(B/batch) *sequentially dependent* optimiser steps on one small MLP
// (940 steps at B=3000).  Launching the batched full-batch kernels once per step would cost
// ~4 launches x 940 steps x 3 nets x 10 consensus epochs per block; instead ONE workgroup owns one
// (seed, adversary) network for the whole fit: its parameters and gradient accumulators live in
// LDS, mini-batch rows are gathered from the replay tensor through the shuffle permutation, and
// the only HBM traffic is the 32-row input gather (L2-resident) and the final write-back.
//
// Shuffle: Keras shuffles with TensorFlow's RNG (irreproducible outside TF); the permutation is
// an INPUT here (perm[S][n_adv][epochs][B], drawn by the host from the stream the oracle defines,
// oracle/rpbcac_oracle.py::ShuffleStream).  perm == NULL keeps the natural row order.
#include "rcmarl_lattice.h"
#include <stdlib.h>

namespace {

constexpr int TR = 32;         // rows per tile (a mini-batch is processed in tiles of 32 rows)
constexpr int KC = 64;         // input columns staged per chunk
constexpr int XLD = KC + 1;    // LDS row stride of the staged chunk (conflict-free column walks)

struct MbArgs {
  const float* x; long x_seed_stride; float* theta; float* adam_m; float* adam_v; const int* agents;
  const float* y;       // MSE: targets [S][N][ldb].  CE: labels (as floats) [S][N][ldb]
  const float* w;       // CE: sample weights [S][N][ldb]
  const int* perm; float* loss_out;
  int N, B, in_dim, ldp, ldb, bs, epochs, n_adv, t0;
  float lr, one_m_b1, one_m_b2, eps;
  double lr_d, beta1, beta2;
};

template <int HID, int OUT, bool ADAM>
__global__ __launch_bounds__(256) void k_minibatch_train(MbArgs a) {
  RCMARL_DYN_SMEM(float, smem);
  const NetGeom g = make_geom(a.in_dim, HID, OUT);
  const int Ppad = (g.P + 3) & ~3;
  float* Ws = smem;                       // parameters, Keras order
  float* Gs = Ws + Ppad;                  // gradient accumulators, same layout
  float* xs = Gs + Ppad;                  // [TR][XLD] staged input chunk
  float* a1s = xs + TR * XLD;             // [TR][HID]
  float* a2s = a1s + TR * HID;
  float* dz2s = a2s + TR * HID;
  float* dz1s = dz2s + TR * HID;
  float* douts = dz1s + TR * HID;         // [TR][OUT]  dLoss/d(output or logits)
  float* lossr = douts + TR * OUT;        // [TR]
  int* idxs = reinterpret_cast<int*>(lossr + TR);   // [TR]
  const int t = threadIdx.x, r = t & (TR - 1), jg = t >> 5;
  const int s = blockIdx.y, adv = blockIdx.x;
  const int agent = a.agents[adv];
  const long row = (long)s * a.N + agent;
  float* th = a.theta + row * a.ldp;
  const float* xg = a.x + (long)s * a.x_seed_stride;
  const float* yv = a.y + row * a.ldb;
  const float* wv = a.w ? a.w + row * a.ldb : nullptr;
  const int* perm = a.perm ? a.perm + ((long)s * a.n_adv + adv) * a.epochs * a.B : nullptr;
  for (int e = t; e < g.P; e += 256) Ws[e] = th[e];
  float loss_part = 0.f;                  // threads < TR: sum over rows of the per-row loss, epoch 0
  int step = a.t0;                        // optimiser steps taken so far (Adam bias correction)
  __syncthreads();
  for (int ep = 0; ep < a.epochs; ++ep) {
    for (int lo = 0; lo < a.B; lo += a.bs) {
      const int nb = min(a.bs, a.B - lo);
      for (int e = t; e < g.P; e += 256) Gs[e] = 0.f;
      for (int t0r = 0; t0r < nb; t0r += TR) {
        const int nrt = min(TR, nb - t0r);
        __syncthreads();                  // previous tile fully consumed (idxs, activations)
        if (t < TR) {
          const int p = lo + t0r + (t < nrt ? t : 0);
          idxs[t] = perm ? perm[(long)ep * a.B + p] : p;
        }
        __syncthreads();
        // ---- layer 1 forward: thread (r, jg) owns units jg, jg+8, jg+16 of row r
        float acc[3] = {0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < a.in_dim; k0 += KC) {
          const int kc = min(KC, a.in_dim - k0);
          __syncthreads();
          for (int e = t; e < TR * KC; e += 256) {
            const int rr = e / KC, kk = e - rr * KC;
            xs[rr * XLD + kk] = (kk < kc) ? xg[(long)idxs[rr] * a.in_dim + k0 + kk] : 0.f;
          }
          __syncthreads();
          for (int kk = 0; kk < kc; ++kk) {
            const float xv = xs[r * XLD + kk];
            const float* wrow = Ws + (long)(k0 + kk) * HID;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int j = jg + 8 * u;
              if (j < HID) acc[u] = fmaf(xv, wrow[j], acc[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int j = jg + 8 * u;
          if (j < HID) a1s[r * HID + j] = rc_lrelu(acc[u] + Ws[g.o_b1 + j]);
        }
        __syncthreads();
        // ---- layer 2
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int j = jg + 8 * u;
          if (j < HID) {
            float z = 0.f;
#pragma unroll
            for (int i = 0; i < HID; ++i) z = fmaf(a1s[r * HID + i], Ws[g.o_W2 + i * HID + j], z);
            a2s[r * HID + j] = rc_lrelu(z + Ws[g.o_b2 + j]);
          }
        }
        __syncthreads();
        // ---- head + loss gradient (one thread per row)
        if (t < TR) {
          const bool valid = t < nrt;
          const int b = idxs[t];
          if (OUT == 1) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < HID; ++k) v = fmaf(a2s[t * HID + k], Ws[g.o_W3 + k], v);
            v += Ws[g.o_b3];
            const float diff = valid ? v - yv[b] : 0.f;
            douts[t] = (2.0f * diff) / (float)nb;               // Keras MSE, SUM_OVER_BATCH_SIZE
            if (ep == 0) loss_part += diff * diff;
          } else {
            float logit[OUT];
            float mx = -3.0e38f;
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
              float z = 0.f;
#pragma unroll
              for (int k = 0; k < HID; ++k) z = fmaf(a2s[t * HID + k], Ws[g.o_W3 + k * OUT + o], z);
              logit[o] = z + Ws[g.o_b3 + o];
              mx = fmaxf(mx, logit[o]);
            }
            float se = 0.f;
#pragma unroll
            for (int o = 0; o < OUT; ++o) se += expf(logit[o] - mx);
            const float lse = logf(se);
            const int label = valid ? (int)yv[b] : 0;
            const float wgt = valid ? wv[b] : 0.f;
            const float wB = wgt / (float)nb;
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
              const float logp = (logit[o] - mx) - lse;
              if (o == label && ep == 0) loss_part += (-logp) * wgt;
              douts[t * OUT + o] = (expf(logp) - (o == label ? 1.f : 0.f)) * wB;
            }
          }
        }
        __syncthreads();
        // ---- dz2 = (dout @ W3^T) * lrelu'(z2)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int k = jg + 8 * u;
          if (k < HID) {
            float da2 = 0.f;
#pragma unroll
            for (int o = 0; o < OUT; ++o) da2 = fmaf(douts[r * OUT + o], Ws[g.o_W3 + k * OUT + o], da2);
            dz2s[r * HID + k] = da2 * rc_lrelu_grad_from_act(a2s[r * HID + k]);
          }
        }
        __syncthreads();
        // ---- dz1 = (dz2 @ W2^T) * lrelu'(z1)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int j = jg + 8 * u;
          if (j < HID) {
            float da1 = 0.f;
#pragma unroll
            for (int k = 0; k < HID; ++k) da1 = fmaf(dz2s[r * HID + k], Ws[g.o_W2 + j * HID + k], da1);
            dz1s[r * HID + j] = da1 * rc_lrelu_grad_from_act(a1s[r * HID + j]);
          }
        }
        __syncthreads();
        // ---- gradient accumulation: every G element is owned by exactly one thread
        for (int e = t; e < HID * HID; e += 256) {              // gW2[j][k] += sum_r a1[r][j]*dz2[r][k]
          const int j = e / HID, k = e - j * HID;
          float sum = 0.f;
          for (int q = 0; q < TR; ++q) sum = fmaf(a1s[q * HID + j], dz2s[q * HID + k], sum);
          Gs[g.o_W2 + e] += sum;
        }
        for (int e = t; e < HID * OUT; e += 256) {              // gW3[k][o] += sum_r a2[r][k]*dout[r][o]
          const int k = e / OUT, o = e - k * OUT;
          float sum = 0.f;
          for (int q = 0; q < TR; ++q) sum = fmaf(a2s[q * HID + k], douts[q * OUT + o], sum);
          Gs[g.o_W3 + e] += sum;
        }
        if (t < HID) {                                          // gb1, gb2
          float s1 = 0.f, s2 = 0.f;
          for (int q = 0; q < TR; ++q) { s1 += dz1s[q * HID + t]; s2 += dz2s[q * HID + t]; }
          Gs[g.o_b1 + t] += s1;
          Gs[g.o_b2 + t] += s2;
        }
        if (t >= 64 && t < 64 + OUT) {                          // gb3
          float sum = 0.f;
          for (int q = 0; q < TR; ++q) sum += douts[q * OUT + (t - 64)];
          Gs[g.o_b3 + (t - 64)] += sum;
        }
        for (int k0 = 0; k0 < a.in_dim; k0 += KC) {             // gW1[k][j] += sum_r x[r][k]*dz1[r][j]
          const int kc = min(KC, a.in_dim - k0);
          if (a.in_dim > KC) {                                  // (single-chunk inputs are still staged)
            __syncthreads();
            for (int e = t; e < TR * KC; e += 256) {
              const int rr = e / KC, kk = e - rr * KC;
              xs[rr * XLD + kk] = (kk < kc) ? xg[(long)idxs[rr] * a.in_dim + k0 + kk] : 0.f;
            }
            __syncthreads();
          }
          for (int e = t; e < kc * HID; e += 256) {
            const int kk = e / HID, j = e - kk * HID;
            float sum = 0.f;
            for (int q = 0; q < TR; ++q) sum = fmaf(xs[q * XLD + kk], dz1s[q * HID + j], sum);
            Gs[(long)(k0 + kk) * HID + j] += sum;
          }
        }
      }
      __syncthreads();
      // ---- optimiser step
      ++step;
      if (ADAM) {
        // TF2 ResourceApplyAdam (see layer1_gemm.hip::EpiAdam); alpha from the step counter
        const float alpha = (float)(a.lr_d * sqrt(1.0 - pow(a.beta2, (double)step)) / (1.0 - pow(a.beta1, (double)step)));
        float* mrow = a.adam_m + row * a.ldp;
        float* vrow = a.adam_v + row * a.ldp;
        for (int e = t; e < g.P; e += 256) {
          const float gr = Gs[e];
          float mm = mrow[e], vv = vrow[e];
          mm += (gr - mm) * a.one_m_b1;
          vv += (gr * gr - vv) * a.one_m_b2;
          mrow[e] = mm; vrow[e] = vv;
          Ws[e] = Ws[e] - (mm * alpha) / (sqrtf(vv) + a.eps);
        }
      } else {
        for (int e = t; e < g.P; e += 256) Ws[e] = Ws[e] - a.lr * Gs[e];
      }
      __syncthreads();
    }
    if (ep == 0 && a.loss_out) {          // Keras History: first-epoch loss (agents/...:122)
      if (t < TR) lossr[t] = loss_part;
      __syncthreads();
      if (t == 0) {
        float sum = 0.f;
        for (int q = 0; q < TR; ++q) sum += lossr[q];
        a.loss_out[row] = sum / (float)a.B;
      }
      __syncthreads();
    }
  }
  for (int e = t; e < g.P; e += 256) th[e] = Ws[e];
}

// ---------------------------------------------------------------------------------------------
// The MSE fits again, ONE WAVEFRONT per (seed, adversary) network (in_dim <= 20, i.e. up to 10 agents for the
// critic / 6 for the team-reward net: the reference's scenarios).  k_minibatch_train spends most of a 32-row step
// in ~12 workgroup barriers and in single-thread reduction loops; a wavefront needs neither:
//   lane = (row r = lane&31 of the 32-row tile, half h = lane>>5); half h computes units 10h..10h+9 of both hidden
//   layers (v_pk_fma_f32, weights broadcast from LDS) and the halves swap their ten values with one cross-half
//   shuffle per value; every gradient is a matrix-core product over the 32 rows,
//     P1 = [a1 | 1 | 0 | a2[0:10]]^T [dz2 | dv | 0]   -> gW2, gb2, gW3[0:10], gb3
//     P2 = [x  | 1 | a2[10:20] | 0]^T [dz1 | dv | 0]  -> gW1, gb1, gW3[10:20]
//   (16 v_mfma_f32_32x32x2_f32 each; operands transposed through two LDS panels, ordered by the wavefront's own
//   program order -- no barrier).  Tiles of a mini-batch keep accumulating in the MFMA registers; the SGD step is
//   applied from the D layout straight into the LDS copy of the parameters (W2 also kept transposed).
// Same arithmetic as k_minibatch_train up to the order in which the 32 rows of a tile are summed.
constexpr int WP_W = 864, WP_W2T = 404, WP_LD = 33, WP_A = 32 * WP_LD, WP_B = 22 * WP_LD;
constexpr int WP_FLOATS = WP_W + WP_W2T + 2 * WP_A + WP_B;  // LDS floats per wavefront (16.4 KiB)

// (RC_WAVE_SYNC: rcmarl_common.h)
#ifdef RCMARL_EMU
__device__ __forceinline__ float rc_other_half(float v) { return __shfl_xor(v, 32, 64); }
#else
// value of the same row's lane in the other half (lane ^ 32): v_permlane32_swap_b32 exchanges lanes 32-63 of its
// first operand with lanes 0-31 of its second -- one VALU instruction + a select instead of a trip through the
// LDS crossbar (ds_bpermute)
__device__ __forceinline__ float rc_other_half(float v) {
  const unsigned u = __float_as_uint(v);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? sw[0] : sw[1]);
}
#endif

template <int INMAX>                                        // inputs padded to INMAX (16 or 20): branch-free loops
__device__ __forceinline__ void wave_fit_net(const MbArgs& a, int net, float* __restrict__ W, int* __restrict__ fix_flags) {
  constexpr int HID = 20, U = 10, XR = 11;                    // XR: first x row of the P2 panel
  const int lane = threadIdx.x & 63;
  // fix-up launch behind k_minibatch_mx: flagged networks only.  The wavefront that redoes a network is the flag's one reader: it
  // clears it (nothing host-side to go stale when the launch pair is replayed from a hipGraph)
  if (fix_flags != nullptr) {
    if (fix_flags[net] == 0) return;
    RC_WAVE_SYNC();
    if ((threadIdx.x & 63) == 0) fix_flags[net] = 0;
  }
  const int s = net / a.n_adv, adv = net - s * a.n_adv;
  float* W2T = W + WP_W;
  float* pA = W2T + WP_W2T;                                  // P1 A panel
  float* pA2 = pA + WP_A;                                    // P2 A panel
  float* pB = pA2 + WP_A;
  const int in = a.in_dim;
  const NetGeom g = make_geom(in, HID, 1);
  const int agent = a.agents[adv];
  const long row = (long)s * a.N + agent;
  float* th = a.theta + row * a.ldp;
  const float* xg = a.x + (long)s * a.x_seed_stride;
  const float* yv = a.y + row * a.ldb;
  const int* perm = a.perm ? a.perm + ((long)s * a.n_adv + adv) * a.epochs * a.B : nullptr;
  const int r = lane & 31, h = lane >> 5, u0 = U * h, l31 = r;
  for (int e = lane; e < g.P; e += 64) W[e] = th[e];
  RC_WAVE_SYNC();
  for (int e = lane; e < HID * HID; e += 64) { const int j = e / HID, k = e - j * HID; W2T[k * HID + j] = W[g.o_W2 + e]; }
  for (int e = lane; e < WP_B; e += 64) pB[e] = 0.f;       // incl. the zero row 21
  for (int e = lane; e < WP_A; e += 64) {                  // constant rows: P1 ones (20) / zeros (21), P2 ones (0) / zero tail
    pA[e] = (e / WP_LD) == 20 ? 1.f : 0.f;
    pA2[e] = (e / WP_LD) == 0 ? 1.f : 0.f;
  }
  RC_WAVE_SYNC();
  rc_f32x16 acc1, acc2;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc1[q] = 0.f; acc2[q] = 0.f; }
  // parameter owned by accumulator slot q of this lane in P1 / P2 (D[row = (q&3) + 8(q>>2) + 4h][col = lane&31]);
  // WP_W - 1 / WP_W2T - 1 are dummy words nobody reads
  int ix1[16], ixT[16], ix2[16];
  float wr1[16], wr2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rw = (q & 3) + 8 * (q >> 2) + 4 * h, col = l31;
    int i1 = WP_W - 1, iT = WP_W2T - 1, i2 = WP_W - 1;
    if (col < HID) {
      if (rw < HID) { i1 = g.o_W2 + rw * HID + col; iT = col * HID + rw; }
      else if (rw == HID) i1 = g.o_b2 + col;
      if (rw == 0) i2 = g.o_b1 + col;
      else if (rw >= XR && rw < XR + in) i2 = (rw - XR) * HID + col;
    } else if (col == HID) {
      if (rw >= 22) i1 = g.o_W3 + (rw - 22);
      else if (rw == HID) i1 = g.o_b3;
      if (rw >= 1 && rw <= U) i2 = g.o_W3 + U + (rw - 1);
    }
    ix1[q] = i1; ixT[q] = iT; ix2[q] = i2;
    wr1[q] = W[i1]; wr2[q] = W[i2];
  }
  const int ibp = (l31 < 21 ? l31 : 21) * WP_LD + h;       // B-panel row of this lane (cols >= 21 read zeros)
  const int iap = l31 * WP_LD + h;
  float loss_part = 0.f;
  // The walk over (epoch, mini-batch, 32-row tile) is flattened so that the rows of tile t+1 (shuffle index, then the
  // input row and the target: two dependent trips to L2) are requested while tile t is processed -- a single
  // wavefront has nothing else to hide that latency with.
  const int tpb = (a.bs + 31) / 32, nbatch = (a.B + a.bs - 1) / a.bs, tpe = tpb * nbatch, T = tpe * a.epochs;
  auto fetch = [&](int t, float (&xo)[INMAX], float& yo, bool& vo) {
    const int ep = t / tpe, w = t - ep * tpe, bi = w / tpb, lo = bi * a.bs;
    const int nb = min(a.bs, a.B - lo), nrt = nb - (w - bi * tpb) * 32;
    vo = r < nrt;
    const int p = lo + (w - bi * tpb) * 32 + (vo ? r : 0);
    const int b = vo ? (perm ? perm[(long)ep * a.B + p] : p) : 0;
#pragma unroll
    for (int k = 0; k < INMAX; ++k) xo[k] = (vo && k < in) ? xg[(long)b * in + k] : 0.f;
    yo = vo ? yv[b] : 0.f;
  };
  float xn[INMAX], ybn;
  bool validn;
  fetch(0, xn, ybn, validn);
  for (int t = 0; t < T; ++t) {
    {
      {
        const int ep = t / tpe, w = t - ep * tpe, bi = w / tpb;
        const int nb = min(a.bs, a.B - bi * a.bs);
        const bool last_tile = (w - bi * tpb) == tpb - 1, last_of_epoch = w == tpe - 1;
        float x[INMAX];
#pragma unroll
        for (int k = 0; k < INMAX; ++k) x[k] = xn[k];
        const float yb = ybn;
        const bool valid = validn;
        if (t + 1 < T) fetch(t + 1, xn, ybn, validn);
        // ---- layer 1, own ten units
        rc_f2 z1[U / 2];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) z1[q] = rc_f2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < INMAX; ++k) {                    // rows >= in_dim multiply x = 0 (whatever finite weight follows W1)
          const rc_f2 xk = rc_bcast2(x[k]);
#pragma unroll
          for (int q = 0; q < U / 2; ++q) {
            const float2 w = *reinterpret_cast<const float2*>(&W[k * HID + u0 + 2 * q]);
            z1[q] = rc_fma2(xk, rc_f2{w.x, w.y}, z1[q]);
          }
        }
        float a1[U], a1o[U];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) {
          a1[2 * q] = rc_lrelu(z1[q].x + W[g.o_b1 + u0 + 2 * q]);
          a1[2 * q + 1] = rc_lrelu(z1[q].y + W[g.o_b1 + u0 + 2 * q + 1]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a1o[u] = rc_other_half(a1[u]);
        // ---- layer 2, own ten units: z2[u] = sum_j a1[j] W2[j][u], j ascending
        rc_f2 z2[U / 2];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) z2[q] = rc_f2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < HID; ++j) {
          const float aj = j < U ? (h ? a1o[j] : a1[j]) : (h ? a1[j - U] : a1o[j - U]);
          const rc_f2 ajj = rc_bcast2(aj);
#pragma unroll
          for (int q = 0; q < U / 2; ++q) {
            const float2 w = *reinterpret_cast<const float2*>(&W[g.o_W2 + j * HID + u0 + 2 * q]);
            z2[q] = rc_fma2(ajj, rc_f2{w.x, w.y}, z2[q]);
          }
        }
        float a2[U], a2o[U];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) {
          a2[2 * q] = rc_lrelu(z2[q].x + W[g.o_b2 + u0 + 2 * q]);
          a2[2 * q + 1] = rc_lrelu(z2[q].y + W[g.o_b2 + u0 + 2 * q + 1]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a2o[u] = rc_other_half(a2[u]);
        // ---- head and loss gradient (both halves compute the same v)
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < HID; ++k) {
          const float ak = k < U ? (h ? a2o[k] : a2[k]) : (h ? a2[k - U] : a2o[k - U]);
          v = fmaf(ak, W[g.o_W3 + k], v);
        }
        v += W[g.o_b3];
        const float diff = valid ? v - yb : 0.f;
        const float dv = (2.0f * diff) / (float)nb;
        if (ep == 0 && h == 0) loss_part += diff * diff;
        float dz2[U], dz2o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dz2[u] = dv * W[g.o_W3 + u0 + u] * rc_lrelu_grad_from_act(a2[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) dz2o[u] = rc_other_half(dz2[u]);
        // ---- dz1, own ten units: da1[u] = sum_k dz2[k] W2[u][k], k ascending (W2 transposed in LDS)
        rc_f2 da[U / 2];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) da[q] = rc_f2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < HID; ++k) {
          const float dk = k < U ? (h ? dz2o[k] : dz2[k]) : (h ? dz2[k - U] : dz2o[k - U]);
          const rc_f2 dkk = rc_bcast2(dk);
#pragma unroll
          for (int q = 0; q < U / 2; ++q) {
            const float2 w = *reinterpret_cast<const float2*>(&W2T[k * HID + u0 + 2 * q]);
            da[q] = rc_fma2(dkk, rc_f2{w.x, w.y}, da[q]);
          }
        }
        float dz1[U];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) {
          dz1[2 * q] = da[q].x * rc_lrelu_grad_from_act(a1[2 * q]);
          dz1[2 * q + 1] = da[q].y * rc_lrelu_grad_from_act(a1[2 * q + 1]);
        }
        // ---- P1 = [a1 | 1 | 0 | a2[0:10]]^T [dz2 | dv | 0]
        RC_WAVE_SYNC();                                      // previous product's fragments consumed
#pragma unroll
        for (int u = 0; u < U; ++u) { pA[(u0 + u) * WP_LD + r] = a1[u]; pB[(u0 + u) * WP_LD + r] = dz2[u]; }
        if (h == 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) pA[(22 + u) * WP_LD + r] = a2[u];
          pB[20 * WP_LD + r] = dv;
        }
        RC_WAVE_SYNC();
#pragma unroll 4
        for (int m = 0; m < 16; ++m) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pA[iap + 2 * m], pB[ibp + 2 * m], acc1, 0, 0, 0);
        // ---- P2 = [1 | a2[10:20] | x | 0]^T [dz1 | dv | 0]   (own A panel: its constant rows are written once)
        RC_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < INMAX / 2; ++k) pA2[(XR + 2 * k + h) * WP_LD + r] = h ? x[2 * k + 1] : x[2 * k];
#pragma unroll
        for (int u = 0; u < U; ++u) pB[(u0 + u) * WP_LD + r] = dz1[u];
        if (h == 1) {
#pragma unroll
          for (int u = 0; u < U; ++u) pA2[(1 + u) * WP_LD + r] = a2[u];
        }
        RC_WAVE_SYNC();
#pragma unroll 4
        for (int m = 0; m < 16; ++m) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pA2[iap + 2 * m], pB[ibp + 2 * m], acc2, 0, 0, 0);
        if (last_tile) {
          // ---- SGD step in the D layout.  Each lane keeps the parameters its 2 x 16 accumulator slots own in
          // registers (master copy) and only WRITES the broadcast copies in LDS: no read-modify-write latency
          // chain, no branches (slots that own nothing write a dummy word).
          RC_WAVE_SYNC();
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            wr1[q] = wr1[q] - a.lr * acc1[q];
            wr2[q] = wr2[q] - a.lr * acc2[q];
            W[ix1[q]] = wr1[q];
            W2T[ixT[q]] = wr1[q];
            W[ix2[q]] = wr2[q];
            acc1[q] = 0.f; acc2[q] = 0.f;
          }
          RC_WAVE_SYNC();
        }
        if (ep == 0 && last_of_epoch && a.loss_out) {          // Keras History: first-epoch loss
          float tl = loss_part;
#pragma unroll
          for (int mk = 16; mk >= 1; mk >>= 1) tl += __shfl_xor(tl, mk, 64);
          if (lane == 0) a.loss_out[row] = tl / (float)a.B;
        }
      }
    }
  }
  RC_WAVE_SYNC();
  for (int e = lane; e < g.P; e += 64) th[e] = W[e];
}

template <int INMAX>
__global__ __launch_bounds__(256) void k_minibatch_wave(MbArgs a, int n_nets, int* __restrict__ fix_flags) {
  RCMARL_DYN_SMEM(float, smem);
  const int wave = threadIdx.x >> 6;
  const int net = blockIdx.x * (blockDim.x >> 6) + wave;
  if (net >= n_nets) return;                               // (wave-uniform; the kernel has no workgroup barrier)
  wave_fit_net<INMAX>(a, net, smem + wave * WP_FLOATS, fix_flags);
}

// Several INDEPENDENT fits (jobs) in one launch: the three chains a Malicious agent needs per consensus epoch (private critic,
// compromised team-reward net, compromised critic: agents/adversarial_CAC_agents.py:131-135,146-152,163-165) start together without
// side streams.  Block b belongs to job j with first[j] <= b < first[j + 1]; one wavefront per network as in the single-job kernels.
constexpr int MB_MAX_JOBS = 4;
struct MbMulti {
  MbArgs a[MB_MAX_JOBS];
  int* flags[MB_MAX_JOBS];
  int first[MB_MAX_JOBS + 1];
  int njobs;
};
__device__ __forceinline__ int mb_job_of(const MbMulti& m, int b) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < MB_MAX_JOBS; ++q) j += (q < m.njobs && b >= m.first[q]) ? 1 : 0;
  return j;
}
template <int INMAX>
__global__ __launch_bounds__(64) void k_minibatch_wave_multi(MbMulti m) {
  RCMARL_DYN_SMEM(float, smem);
  const int j = mb_job_of(m, (int)blockIdx.x);
  wave_fit_net<INMAX>(m.a[j], (int)blockIdx.x - m.first[j], smem, m.flags[j]);
}

// ---------------------------------------------------------------------------------------------
// The MSE fits on the f16 matrix core ("mx", round 3): one wavefront per (seed, adversary) network as above, every product of
// the step as v_mfma_f32_32x32x16_f16 on two-piece f16 operands (rcmarl_lattice.h; the data flow of k_mid_fit_v8).
// Why: k_minibatch_wave reads every weight of every layer product from LDS as a broadcast float2 -- 280 ds_read_b64 = 143 KB
// of LDS return traffic per 32-row step and wavefront; with the 512 x 3 networks of BASELINE configs[1] batched over seeds (six
// wavefronts per CU) that alone is 3 us per step, and the 2 x 16 f32-input MFMAs of its gradient products run on the vector
// ALUs.  Here the weights are the STATIONARY matrix operand -- twelve 16-byte fragment reads per step -- and the step is
//   z1 = x W1, z2 = a1 W2, da1 = dz2 W2^T      3 x (1..2 k-steps) x 4 piece products, lane (row j, half h) = ten units of row j
//   G1 = [a1 | 1 | 0 | a2_h0]^T [dz2 | dv]     -> gW2, gb2, gW3 (units of half 0), gb3
//   G2 = [x  | 1 | 0 | a2_h1]^T [dz1 | dv]     -> gW1, gb1, gW3 (units of half 1)
// with the operands of G1, G2 written row-major into f16 LDS planes and read back transposed (ds_read_b64_tr_b16).  The fp32
// master copy of every parameter lives in the register of the accumulator slot that owns its gradient (as in k_minibatch_wave);
// after the SGD step the owner writes the parameter's two f16 pieces straight into the weight fragments (ds_write_b16).
// Scales: W'' = 2^10 W, dz'' = 2^10 dz, dv'' = 2^10 dv; x, a1, a2 unscaled.  A network whose operands leave the f16 range is NOT
// written back: its flag is set and k_minibatch_wave redoes the whole fit in fp32 arithmetic (second launch, flagged networks only).
// Measured (512 seeds x 3 networks, BASELINE configs[1] batched): 2.9 us per step alone against 6.0 for k_minibatch_wave; inside a
// block, beside the cooperative agents' kernels, 5.2 us (clocks, LDS and L2 shared) -- the block goes 200 -> 163 ms, a single
// instance 65 -> 35 ms.  Capping a launch at one wavefront per CU (so that the other kernels find LDS) changes nothing: 162-164 ms.
constexpr int MX_PCA = 32, MX_PCB = 24;                          // plane row strides (f16 elements)
constexpr int MX_FRAG = 3 * 2 * 2 * 2 * 32 * 8;                  // [product][k-step][piece][k-group][row] x 8 f16
constexpr int MX_PA = 2 * 32 * MX_PCA, MX_PB = 2 * 32 * MX_PCB;  // one A / B plane pair (two pieces)
constexpr int MX_F16 = MX_FRAG + 2 * MX_PA + 2 * MX_PB + 32;     // + pad: a transposed read of B columns 24..31 of the last row
constexpr int MX_PARAMS = 64;                                    // fp32: b1[20] | b2[20] | W3[20] | b3 | two dummy slots
constexpr size_t MX_BYTES = (size_t)MX_F16 * 2 + MX_PARAMS * 4;  // 26 944 B per wavefront: six per CU
// COMPACT form (more networks than the GPU has wavefront slots: the cooperative agents' full-batch fits of 512 seeds): k-step 1 of
// the weight operands holds 4 (z1: features 16..19) or 2 (z2, da: local units 8, 9) slots per k-group and is stored as such, and
// the B planes of G2 overwrite those of G1 -> 19 264 B: EIGHT wavefronts per CU, at the price of G1 and G2 no longer interleaving.
constexpr int MX_CF0 = 3 * 2 * 2 * 32 * 8;                       // k-step 0: [product][piece][k-group][row] x 8 f16
constexpr int MX_CF1 = 2 * 32 * 4;                               // k-step 1 of z1: [piece][row] x 4 f16 (k-group 0 only)
constexpr int MX_CF2 = 2 * 2 * 2 * 32 * 2;                       // k-step 1 of z2, da: [product][piece][k-group][row] x 2 f16
constexpr int MX_CFRAG = MX_CF0 + MX_CF1 + MX_CF2;
constexpr int MX_CF16 = MX_CFRAG + 2 * MX_PA + MX_PB + 32;
constexpr size_t MX_CBYTES = (size_t)MX_CF16 * 2 + MX_PARAMS * 4;
#define MX_S 1024.f
#define MX_US 0.0009765625f
#define MX_RANGE 65000.f

__device__ __forceinline__ int mx_row_of_unit(int u) { return u < 16 ? 8 * ((u & 7) >> 2) + 4 * (u >> 3) + (u & 3) : 16 + 4 * ((u - 16) >> 1) + ((u - 16) & 1); }
__device__ __forceinline__ int mx_slot_of_unit(int u) { return u < 16 ? u : 16 + 8 * ((u - 16) >> 1) + ((u - 16) & 1); }
template <bool COMPACT>
__device__ __forceinline__ int mx_frag_elem(int prod, int ri, int k) {      // f16 index of piece 0; piece 1: mx_piece1() further
  const int kg = (k >> 3) & 1;
  if (!COMPACT) return ((((prod * 2 + (k >> 4)) * 2 + 0) * 2 + kg) * 32 + ri) * 8 + (k & 7);
  if (k < 16) return (((prod * 2 + 0) * 2 + kg) * 32 + ri) * 8 + (k & 7);
  if (prod == 0) return MX_CF0 + ri * 4 + (k & 3);                             // features 16..19
  return MX_CF0 + MX_CF1 + ((((prod - 1) * 2 + 0) * 2 + kg) * 32 + ri) * 2 + (k & 1);
}
template <bool COMPACT>
__device__ __forceinline__ int mx_piece1(int elem) { return !COMPACT ? 2 * 32 * 8 : (elem < MX_CF0 ? 2 * 32 * 8 : 128); }

#ifdef RCMARL_EMU
#define RC_MX_OCC
#else
#define RC_MX_OCC __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(2)))   // <= 256 registers
#endif
template <int KS1, bool COMPACT>                             // k-steps of layer 1: 1 (<= 16 inputs) or 2 (<= 20)
__device__ __forceinline__ void mx_fit_net(const MbArgs& a, int net, unsigned char* smem, int* __restrict__ ovf_flags) {
  constexpr int HID = 20, LU = 10, NX = 8 * KS1;
  const int lane = threadIdx.x & 63;
  unsigned short* frag = reinterpret_cast<unsigned short*>(smem);
  constexpr int NFRAG = COMPACT ? MX_CFRAG : MX_FRAG;
  unsigned short* pA1 = frag + NFRAG;                        // [piece][row][32]: a1 (0..19) | 1 (20) | 0 (21) | a2 of half 0: local 8,9 (22,23), 0..7 (24..31)
  unsigned short* pA2 = pA1 + MX_PA;                         // [piece][row][32]: x  (0..19) | 1 (20) | 0 (21) | a2 of half 1
  unsigned short* pB1 = pA2 + MX_PA;                         // [piece][row][24]: dz2'' (0..19) | dv'' (20)
  unsigned short* pB2 = COMPACT ? pB1 : pB1 + MX_PB;         // [piece][row][24]: dz1'' (0..19) | dv'' (20)   (COMPACT: over G1's)
  float* prm = reinterpret_cast<float*>(smem + (size_t)(COMPACT ? MX_CF16 : MX_F16) * 2);       // b1 | b2 | W3 | b3
  const int in = a.in_dim;
  const NetGeom g = make_geom(in, HID, 1);
  const int l31 = lane & 31, half = lane >> 5;
  rc_f16_saturate();
  const int fdummy = mx_frag_elem<COMPACT>(0, 31, 0);
  const uint4* wfA = reinterpret_cast<const uint4*>(frag) + half * 32 + l31;    // + ((prod*2 + ks)*2 + piece) * 64  (COMPACT: (prod*2 + piece) * 64)
  auto loadA = [&](int prod, int ks) {
    V8Pieces f;
    if (!COMPACT) {
      f.h = wfA[((prod * 2 + ks) * 2 + 0) * 64];
      f.l = wfA[((prod * 2 + ks) * 2 + 1) * 64];
    } else if (ks == 0) {
      f.h = wfA[(prod * 2 + 0) * 64];
      f.l = wfA[(prod * 2 + 1) * 64];
    } else if (prod == 0) {                                  // features 16..19: k-group 0 only
      const uint2* q = reinterpret_cast<const uint2*>(frag + MX_CF0) + l31;
      const uint2 vh = q[0], vl = q[32];
      f.h.x = half ? 0u : vh.x; f.h.y = half ? 0u : vh.y; f.h.z = 0u; f.h.w = 0u;
      f.l.x = half ? 0u : vl.x; f.l.y = half ? 0u : vl.y; f.l.z = 0u; f.l.w = 0u;
    } else {                                                 // local units 8, 9 of the k-group
      const unsigned* q = reinterpret_cast<const unsigned*>(frag + MX_CF0 + MX_CF1) + (((prod - 1) * 2 + 0) * 2 + half) * 32 + l31;
      f.h.x = q[0]; f.h.y = 0u; f.h.z = 0u; f.h.w = 0u;
      f.l.x = q[64]; f.l.y = 0u; f.l.z = 0u; f.l.w = 0u;
    }
    return f;
  };
  const int trA = (8 * half + ((lane & 15) >> 2)) * MX_PCA + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int trB = (8 * half + ((lane & 15) >> 2)) * MX_PCB + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto readT = [&](const unsigned short* plane, int pl_elems, int o, int pc) {
    V8Pieces r;
    uint2 t0 = rc_lds_read_tr16(plane + o), t1 = rc_lds_read_tr16(plane + o + 4 * pc);
    r.h.x = t0.x; r.h.y = t0.y; r.h.z = t1.x; r.h.w = t1.y;
    t0 = rc_lds_read_tr16(plane + pl_elems + o); t1 = rc_lds_read_tr16(plane + pl_elems + o + 4 * pc);
    r.l.x = t0.x; r.l.y = t0.y; r.l.z = t1.x; r.l.w = t1.y;
    return r;
  };
  uint4 z4;
  z4.x = z4.y = z4.z = z4.w = 0u;
  const int wr8A = l31 * MX_PCA + 8 * half, wr2A = l31 * MX_PCA + 16 + 2 * half;
  const int wr8B = l31 * MX_PCB + 8 * half, wr2B = l31 * MX_PCB + 16 + 2 * half;
  unsigned short* pAmine = half ? pA2 : pA1;                 // where this half's a2 values go (columns 22..31)
  const int tpb = (a.bs + 31) / 32, nbatch = (a.B + a.bs - 1) / a.bs, tpe = tpb * nbatch, T = tpe * a.epochs;
    const int s = net / a.n_adv, adv = net - s * a.n_adv;
    const int agent = a.agents[adv];
    const long row = (long)s * a.N + agent;
    float* th = a.theta + row * a.ldp;
    const float* xg = a.x + (long)s * a.x_seed_stride;
    const float* yv = a.y + row * a.ldb;
    const int* perm = a.perm ? a.perm + ((long)s * a.n_adv + adv) * a.epochs * a.B : nullptr;
    float amax = 0.f;                                        // largest |scaled operand| this lane formed
    // ---- planes: constants once per network (ones column 20 of both A pairs as (1.0, 0); everything else zero)
    RC_WAVE_SYNC();
    for (int e = lane; e < (COMPACT ? MX_CF16 : MX_F16); e += 64) frag[e] = 0;
    RC_WAVE_SYNC();
    if (lane < 32) { pA1[lane * MX_PCA + 20] = 0x3C00; pA2[lane * MX_PCA + 20] = 0x3C00; }
    // ---- ownership: accumulator slot q of this lane is G[i = (q&3) + 8(q>>2) + 4 half][jj = l31] of G1 / G2
    float wr1[16], wr2[16];                                  // fp32 masters of the owned parameters
    // where the owner publishes them (packed to keep the wavefront under 256 registers): fragment elements (piece 0) of W2 in the
    // z2 and da operands / of W1 in the z1 operand, fp32 slots in prm.  Slots that own nothing point at DUMMY targets: fragment row
    // 31 (a padding row of the weight operand: it feeds accumulator rows nobody reads) and prm[62], prm[63].
    // COMPACT keeps none of them: it recomputes them when it publishes (once per epoch in a full-batch fit) and has 32 registers more.
    auto targets = [&](int q, int& o1, int& o2, unsigned& k0, unsigned& k1) {
      const int i = (q & 3) + 8 * (q >> 2) + 4 * half, jj = l31;
      o1 = -1; o2 = -1;                                      // offsets into theta (Keras order)
      int f1a = fdummy, f1b = fdummy, f2a = fdummy, p1 = 62, p2 = 63;
      if (jj < HID) {
        if (i < HID) {                                       // W2[m = i][unit = jj]
          o1 = g.o_W2 + i * HID + jj;
          f1a = mx_frag_elem<COMPACT>(1, mx_row_of_unit(jj), mx_slot_of_unit(i));      // z2: A[row(unit)][slot(m)]
          f1b = mx_frag_elem<COMPACT>(2, mx_row_of_unit(i), mx_slot_of_unit(jj));      // da: A[row(m)][slot(unit)]
        } else if (i == HID) { o1 = g.o_b2 + jj; p1 = HID + jj; }
        if (i < in) { o2 = i * HID + jj; f2a = mx_frag_elem<COMPACT>(0, mx_row_of_unit(jj), i); }   // W1[k = i][unit = jj]: slot = feature
        else if (i == HID) { o2 = g.o_b1 + jj; p2 = jj; }
      } else if (jj == HID) {
        if (i == HID) { o1 = g.o_b3; p1 = 3 * HID; }
        else if (i >= 22) {
          const int lu = i < 24 ? 8 + (i - 22) : i - 24;
          o1 = g.o_W3 + v8_unit(0, lu); p1 = 2 * HID + v8_unit(0, lu);
          o2 = g.o_W3 + v8_unit(1, lu); p2 = 2 * HID + v8_unit(1, lu);
        }
      }
      k0 = (unsigned)f1a | ((unsigned)f1b << 16);            // f1a | f1b << 16
      k1 = (unsigned)f2a | ((unsigned)p1 << 16) | ((unsigned)p2 << 24);      // f2a | p1 << 16 | p2 << 24
    };
    unsigned pk0[COMPACT ? 1 : 16], pk1[COMPACT ? 1 : 16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int o1, o2;
      unsigned k0, k1;
      targets(q, o1, o2, k0, k1);
      if (!COMPACT) { pk0[q] = k0; pk1[q] = k1; }
      wr1[q] = o1 >= 0 ? th[o1] : 0.f;
      wr2[q] = o2 >= 0 ? th[o2] : 0.f;
    }
    // the owner's view -> fragments and fp32 parameters (also after every SGD step)
    auto publish = [&]() {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        unsigned h1, l1;
        rc_split2h_pair(wr1[q] * MX_S, wr2[q] * MX_S, h1, l1);  // low halves: wr1's pieces, high halves: wr2's
        unsigned k0, k1;
        if (COMPACT) { int o1, o2; targets(q, o1, o2, k0, k1); } else { k0 = pk0[q]; k1 = pk1[q]; }
        const int f1a = k0 & 0xffffu, f1b = k0 >> 16, f2a = k1 & 0xffffu;
        frag[f1a] = (unsigned short)h1; frag[f1a + mx_piece1<COMPACT>(f1a)] = (unsigned short)l1;
        frag[f1b] = (unsigned short)h1; frag[f1b + mx_piece1<COMPACT>(f1b)] = (unsigned short)l1;
        frag[f2a] = (unsigned short)(h1 >> 16); frag[f2a + mx_piece1<COMPACT>(f2a)] = (unsigned short)(l1 >> 16);
        prm[(k1 >> 16) & 0xffu] = wr1[q];
        prm[k1 >> 24] = wr2[q];
        amax = fmaxf(amax, fmaxf(fabsf(wr1[q]), fabsf(wr2[q])) * MX_S);   // (biases and W3 too: conservative, branch-free)
      }
    };
    RC_WAVE_SYNC();
    publish();
    RC_WAVE_SYNC();
    rc_f32x16 g1, g2;
#pragma unroll
    for (int q = 0; q < 16; ++q) { g1[q] = 0.f; g2[q] = 0.f; }
    float loss_part = 0.f;
    // Two dependent trips to L2 per tile -- its rows' shuffle indices, then the input rows and targets they name -- are taken one
    // tile apart: while tile t runs, the ROWS of tile t+1 (index known) and the INDEX of tile t+2 are in flight: a single wavefront
    // has nothing else to hide that latency with, and beside the cooperative agents' kernels it is several microseconds.
    auto fetch_idx = [&](int t, int& bo, bool& vo) {
      const int ep = t / tpe, w = t - ep * tpe, bi = w / tpb, lo = bi * a.bs;
      const int nb = min(a.bs, a.B - lo), nrt = nb - (w - bi * tpb) * 32;
      vo = l31 < nrt;
      const int p = lo + (w - bi * tpb) * 32 + (vo ? l31 : 0);
      bo = vo ? (perm ? perm[(long)ep * a.B + p] : p) : 0;
    };
    auto fetch_rows = [&](int b, bool vo, float (&xo)[NX], float& yo) {
#pragma unroll
      for (int e = 0; e < NX; ++e) {
        const int k = 16 * (e >> 3) + 8 * half + (e & 7);   // this lane's contraction slots: features 8 half + e, 16 + 8 half + e
        xo[e] = (vo && k < in) ? xg[(long)b * in + k] : 0.f;
      }
      yo = vo ? yv[b] : 0.f;
    };
    float xn[NX], ybn;
    bool validn, valid2 = false;
    int bn, b2 = 0;
    fetch_idx(0, bn, validn);
    fetch_rows(bn, validn, xn, ybn);
    if (T > 1) fetch_idx(1, b2, valid2);
    for (int t = 0; t < T; ++t) {
      const int ep = t / tpe, w = t - ep * tpe, bi = w / tpb;
      const int nb = min(a.bs, a.B - bi * a.bs);
      const bool last_tile = (w - bi * tpb) == tpb - 1, last_of_epoch = w == tpe - 1;
      float x[NX];
#pragma unroll
      for (int e = 0; e < NX; ++e) x[e] = xn[e];
      const float yb = ybn;
      const bool valid = validn;
      if (t + 1 < T) {
        validn = valid2;
        fetch_rows(b2, valid2, xn, ybn);
        if (t + 2 < T) fetch_idx(t + 2, b2, valid2);
      }
      // ---- layer 1: B = x pieces (also the main columns of the A2 planes)
      V8Pieces px0, px1;
      {
        const float x0[8] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
        px0 = v8_split8<false>(x0, 1.f);
        px1.h = z4; px1.l = z4;
        if (KS1 == 2) {
          rc_split2h_pair(x[NX - 8], x[NX - 7], px1.h.x, px1.l.x);
          rc_split2h_pair(x[NX - 6], x[NX - 5], px1.h.y, px1.l.y);     // features 16..19 (half 0; zeros in half 1 and beyond in_dim)
        }
      }
      RC_WAVE_SYNC();                                        // the previous tile's transposed reads are done
      *reinterpret_cast<uint4*>(pA2 + wr8A) = px0.h;
      *reinterpret_cast<uint4*>(pA2 + MX_PA / 2 + wr8A) = px0.l;
      if (KS1 == 2 && half == 0) {
        *reinterpret_cast<uint2*>(pA2 + l31 * MX_PCA + 16) = make_uint2(px1.h.x, px1.h.y);
        *reinterpret_cast<uint2*>(pA2 + MX_PA / 2 + l31 * MX_PCA + 16) = make_uint2(px1.l.x, px1.l.y);
      }
      rc_f32x16 zz;
#pragma unroll
      for (int q = 0; q < 16; ++q) zz[q] = 0.f;
      if (KS1 == 2) zz = v8_mfma4(loadA(0, 1), px1, zz);
      zz = v8_mfma4(loadA(0, 0), px0, zz);
      float a1[LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) a1[u] = rc_lrelu(fmaf(zz[u], MX_US, prm[v8_unit(half, u)]));
      // ---- layer 2
      V8Pieces pa0, pa1;
      {
        const float x0[8] = {a1[0], a1[1], a1[2], a1[3], a1[4], a1[5], a1[6], a1[7]};
        pa0 = v8_split8<false>(x0, 1.f);
        pa1.h = z4; pa1.l = z4;
        rc_split2h_pair(a1[8], a1[9], pa1.h.x, pa1.l.x);
      }
      *reinterpret_cast<uint4*>(pA1 + wr8A) = pa0.h;
      *reinterpret_cast<uint4*>(pA1 + MX_PA / 2 + wr8A) = pa0.l;
      *reinterpret_cast<unsigned*>(pA1 + wr2A) = pa1.h.x;
      *reinterpret_cast<unsigned*>(pA1 + MX_PA / 2 + wr2A) = pa1.l.x;
#pragma unroll
      for (int q = 0; q < 16; ++q) zz[q] = 0.f;
      zz = v8_mfma4(loadA(1, 1), pa1, zz);
      zz = v8_mfma4(loadA(1, 0), pa0, zz);
      float a2[LU], vp = 0.f;
#pragma unroll
      for (int u = 0; u < LU; ++u) a2[u] = rc_lrelu(fmaf(zz[u], MX_US, prm[HID + v8_unit(half, u)]));
#pragma unroll
      for (int u = 0; u < LU; ++u) vp = fmaf(a2[u], prm[2 * HID + v8_unit(half, u)], vp);
      float va = vp, vb = vp;
      rc_swap32(va, vb);                                     // va: lanes 32-63 hold the low half's partial; vb: lanes 0-31 the high half's
      const float v = (vp + (half ? va : vb)) + prm[3 * HID];
      const float diff = valid ? v - yb : 0.f;
      const float dvs = ((2.0f * diff) / (float)nb) * MX_S;   // dv'' = 2^10 dv
      if (ep == 0 && half == 0) loss_part = fmaf(diff, diff, loss_part);
      // a2 of this half -> columns 22..31 of its own A planes (local 8, 9 at 22, 23; local 0..7 at 24..31)
      {
        V8Pieces q0, q1;
        const float x0[8] = {a2[0], a2[1], a2[2], a2[3], a2[4], a2[5], a2[6], a2[7]};
        q0 = v8_split8<false>(x0, 1.f);
        rc_split2h_pair(a2[8], a2[9], q1.h.x, q1.l.x);
        *reinterpret_cast<uint4*>(pAmine + l31 * MX_PCA + 24) = q0.h;
        *reinterpret_cast<uint4*>(pAmine + MX_PA / 2 + l31 * MX_PCA + 24) = q0.l;
        *reinterpret_cast<unsigned*>(pAmine + l31 * MX_PCA + 22) = q1.h.x;
        *reinterpret_cast<unsigned*>(pAmine + MX_PA / 2 + l31 * MX_PCA + 22) = q1.l.x;
      }
      float dz2[LU];                                           // 2^10 dz2
#pragma unroll
      for (int u = 0; u < LU; ++u) dz2[u] = dvs * prm[2 * HID + v8_unit(half, u)] * rc_lrelu_grad_from_act(a2[u]);
      V8Pieces pd0, pd1;
      {
        const float x0[8] = {dz2[0], dz2[1], dz2[2], dz2[3], dz2[4], dz2[5], dz2[6], dz2[7]};
        pd0 = v8_split8<false>(x0, 1.f);
        pd1.h = z4; pd1.l = z4;
        rc_split2h_pair(dz2[8], dz2[9], pd1.h.x, pd1.l.x);
      }
      unsigned dvh, dvl;
      rc_split2h_pair(dvs, 0.f, dvh, dvl);
      *reinterpret_cast<uint4*>(pB1 + wr8B) = pd0.h;
      *reinterpret_cast<uint4*>(pB1 + MX_PB / 2 + wr8B) = pd0.l;
      *reinterpret_cast<unsigned*>(pB1 + wr2B) = pd1.h.x;
      *reinterpret_cast<unsigned*>(pB1 + MX_PB / 2 + wr2B) = pd1.l.x;
      if (half == 0) {
        pB1[l31 * MX_PCB + 20] = (unsigned short)dvh; pB1[MX_PB / 2 + l31 * MX_PCB + 20] = (unsigned short)dvl;
        pB2[l31 * MX_PCB + 20] = (unsigned short)dvh; pB2[MX_PB / 2 + l31 * MX_PCB + 20] = (unsigned short)dvl;
      }
      // ---- layer 2 backward: dd = sum_j dz2''[j] W2''[m][j] = 2^20 da1
      rc_f32x16 dd;
#pragma unroll
      for (int q = 0; q < 16; ++q) dd[q] = 0.f;
      dd = v8_mfma4(loadA(2, 1), pd1, dd);
      dd = v8_mfma4(loadA(2, 0), pd0, dd);
      if (COMPACT) {                                           // G1 now: its B planes are about to be overwritten by G2's
        RC_WAVE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          g1 = v8_mfma4(readT(pA1, MX_PA / 2, trA + 16 * MX_PCA * ks, MX_PCA), readT(pB1, MX_PB / 2, trB + 16 * MX_PCB * ks, MX_PCB), g1);
      }
      float dz1[LU];                                           // 2^10 dz1
#pragma unroll
      for (int u = 0; u < LU; ++u) dz1[u] = (dd[u] * MX_US) * rc_lrelu_grad_from_act(a1[u]);
      V8Pieces qd0, qd1;
      {
        const float x0[8] = {dz1[0], dz1[1], dz1[2], dz1[3], dz1[4], dz1[5], dz1[6], dz1[7]};
        qd0 = v8_split8<false>(x0, 1.f);
        rc_split2h_pair(dz1[8], dz1[9], qd1.h.x, qd1.l.x);
      }
      auto write_dz1 = [&]() {
        *reinterpret_cast<uint4*>(pB2 + wr8B) = qd0.h;
        *reinterpret_cast<uint4*>(pB2 + MX_PB / 2 + wr8B) = qd0.l;
        *reinterpret_cast<unsigned*>(pB2 + wr2B) = qd1.h.x;
        *reinterpret_cast<unsigned*>(pB2 + MX_PB / 2 + wr2B) = qd1.l.x;
      };
      if (!COMPACT) write_dz1();
#pragma unroll
      for (int u = 0; u < LU; u += 2) {
        amax = fmaxf(amax, fmaxf(fabsf(a1[u]), fabsf(a1[u + 1])));
        amax = fmaxf(amax, fmaxf(fabsf(a2[u]), fabsf(a2[u + 1])));
        amax = fmaxf(amax, fmaxf(fabsf(dz2[u]), fabsf(dz2[u + 1])));
        amax = fmaxf(amax, fmaxf(fabsf(dz1[u]), fabsf(dz1[u + 1])));
      }
      amax = fmaxf(amax, fabsf(dvs));
      // ---- the two gradient products over the tile's 32 rows (operands read back transposed)
      RC_WAVE_SYNC();
      if (!COMPACT) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const V8Pieces ra1 = readT(pA1, MX_PA / 2, trA + 16 * MX_PCA * ks, MX_PCA), rb1 = readT(pB1, MX_PB / 2, trB + 16 * MX_PCB * ks, MX_PCB);
          const V8Pieces ra2 = readT(pA2, MX_PA / 2, trA + 16 * MX_PCA * ks, MX_PCA), rb2 = readT(pB2, MX_PB / 2, trB + 16 * MX_PCB * ks, MX_PCB);
          g1 = rc_mfma_f16(ra1.l, rb1.l, g1); g2 = rc_mfma_f16(ra2.l, rb2.l, g2);
          g1 = rc_mfma_f16(ra1.l, rb1.h, g1); g2 = rc_mfma_f16(ra2.l, rb2.h, g2);
          g1 = rc_mfma_f16(ra1.h, rb1.l, g1); g2 = rc_mfma_f16(ra2.h, rb2.l, g2);
          g1 = rc_mfma_f16(ra1.h, rb1.h, g1); g2 = rc_mfma_f16(ra2.h, rb2.h, g2);
        }
      } else {
        write_dz1();                                         // (G1's transposed reads were issued above: the LDS executes a wavefront's instructions in order)
        RC_WAVE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          g2 = v8_mfma4(readT(pA2, MX_PA / 2, trA + 16 * MX_PCA * ks, MX_PCA), readT(pB2, MX_PB / 2, trB + 16 * MX_PCB * ks, MX_PCB), g2);
      }
      if (last_tile) {
        // ---- SGD step: the owner of a gradient slot holds the parameter's fp32 master and refreshes its broadcast copies
        const float lrs = a.lr * MX_US;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          wr1[q] = wr1[q] - lrs * g1[q];
          wr2[q] = wr2[q] - lrs * g2[q];
          g1[q] = 0.f; g2[q] = 0.f;
        }
        RC_WAVE_SYNC();                                      // (the fragment reads of this tile are done)
        publish();
        RC_WAVE_SYNC();
      }
      if (ep == 0 && last_of_epoch && a.loss_out) {            // Keras History: first-epoch loss
        float tl = loss_part;
#pragma unroll
        for (int mk = 16; mk >= 1; mk >>= 1) tl += __shfl_xor(tl, mk, 64);
        if (lane == 0) a.loss_out[row] = tl / (float)a.B;
      }
    }
    // ---- write back -- unless an operand left the f16 range anywhere in the wavefront: then the fp32 kernel redoes this network
    float am = amax;
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) am = fmaxf(am, __shfl_xor(am, mk, 64));
    if (am > MX_RANGE) {
      if (lane == 0) ovf_flags[net] = 1;
      return;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = (q & 3) + 8 * (q >> 2) + 4 * half, jj = l31;
      int o1 = -1, o2 = -1;
      if (jj < HID) {
        if (i < HID) o1 = g.o_W2 + i * HID + jj; else if (i == HID) o1 = g.o_b2 + jj;
        if (i < in) o2 = i * HID + jj; else if (i == HID) o2 = g.o_b1 + jj;
      } else if (jj == HID) {
        if (i == HID) o1 = g.o_b3;
        else if (i >= 22) { const int lu = i < 24 ? 8 + (i - 22) : i - 24; o1 = g.o_W3 + v8_unit(0, lu); o2 = g.o_W3 + v8_unit(1, lu); }
      }
      if (o1 >= 0) th[o1] = wr1[q];
      if (o2 >= 0) th[o2] = wr2[q];
    }
}

template <int KS1, bool COMPACT>
__global__ RC_MX_OCC void k_minibatch_mx(MbArgs a, int net0, int* __restrict__ ovf_flags) {
  RCMARL_DYN_SMEM(unsigned char, smem);
  mx_fit_net<KS1, COMPACT>(a, net0 + (int)blockIdx.x, smem, ovf_flags);
}

template <int KS1, bool COMPACT>
__global__ RC_MX_OCC void k_minibatch_mx_multi(MbMulti m) {
  RCMARL_DYN_SMEM(unsigned char, smem);
  const int j = mb_job_of(m, (int)blockIdx.x);
  mx_fit_net<KS1, COMPACT>(m.a[j], (int)blockIdx.x - m.first[j], smem, m.flags[j]);
}

// wavefronts of the 27-KB form the GPU holds at once: FIVE per CU, not the six its LDS arithmetic promises (measured, round 5,
// tools/kbench.py multi, profiles/r05k_*: 1026 networks 4.1 ms, 1536 networks 7.2 ms -- two rounds -- against 5.6 ms in the compact
// form); beyond that the compact form (eight per CU) takes over
int mx_slots() {
  static const int n = [] {
#ifdef RCMARL_EMU
    return 1280;
#else
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return 5 * cus;
#endif
  }();
  return n;
}

size_t mb_smem_bytes(int in_dim, int hid, int out) {
  const NetGeom g = make_geom(in_dim, hid, out);
  const int Ppad = (g.P + 3) & ~3;
  return sizeof(float) * ((size_t)2 * Ppad + TR * XLD + 4 * TR * hid + TR * out + TR) + sizeof(int) * TR;
}

template <class K>
int mb_launch(K kernel, const MbArgs& a, int S, size_t smem, void* stream) {
  if (smem > 160 * 1024) return RCMARL_ERR_UNSUPPORTED;
  if (!rc_want_lds(kernel, smem, 48 * 1024)) return RCMARL_ERR_LAUNCH;
  const dim3 grid(a.n_adv, S), block(256);
  RCMARL_LAUNCH(kernel, grid, block, smem, stream, a);
  return rcmarl_check_launch();
}

}  // namespace

RCMARL_EXPORT int rcmarl_minibatch_fit(const float* x, long x_seed_stride, float* theta, const int* agents, int n_adv,
                                       const float* y, const int* perm, int S, int N, int B, int in_dim, int hid,
                                       int ldp, int ldb, int batch_size, int epochs, float lr, float* loss_out,
                                       int* ovf_flags, void* stream) {
  if (!x || !theta || !agents || !y || n_adv <= 0 || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || batch_size <= 0 ||
      epochs <= 0 || (ldp & 63) || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid != 20) return RCMARL_ERR_UNSUPPORTED;
  MbArgs a{};
  a.x = x; a.x_seed_stride = x_seed_stride; a.theta = theta; a.agents = agents; a.y = y; a.perm = perm;
  a.loss_out = loss_out; a.N = N; a.B = B; a.in_dim = in_dim; a.ldp = ldp; a.ldb = ldb;
  a.bs = batch_size < B ? batch_size : B; a.epochs = epochs; a.n_adv = n_adv; a.lr = lr;
  // <= 20 inputs (the reference's own 5-agent scenarios): one wavefront per network -- on the f16 matrix core (k_minibatch_mx) with
  // the fp32 wavefront kernel as the fix-up for networks that leave the f16 range, or (RCMARL_MB_MX=0) the fp32 kernel alone;
  // wider inputs: one workgroup per network
  if (in_dim <= 20) {
    const int n_nets = n_adv * S;
    const size_t smem = (size_t)4 * WP_FLOATS * sizeof(float);
    static const bool attr_ok = rc_want_lds(k_minibatch_wave<16>, smem) && rc_want_lds(k_minibatch_wave<20>, smem);
    if (!attr_ok) return RCMARL_ERR_LAUNCH;
    int* flags = nullptr;
    const char* e = getenv("RCMARL_MB_MX");
    if (ovf_flags != nullptr && !(e && atoi(e) == 0)) {       // (no flag buffer: the fp32 kernel alone)
      int* fl = ovf_flags;
      flags = fl;
      // more networks than wavefront slots at six per CU: the compact form (eight per CU)
      const char* ce = getenv("RCMARL_MB_MX_COMPACT");
      const bool compact = ce ? atoi(ce) != 0 : n_nets > mx_slots();
      if (compact) {
        if (in_dim <= 16) {
          RCMARL_LAUNCH((k_minibatch_mx<1, true>), dim3(n_nets), dim3(64), MX_CBYTES, stream, a, 0, fl);
        } else {
          RCMARL_LAUNCH((k_minibatch_mx<2, true>), dim3(n_nets), dim3(64), MX_CBYTES, stream, a, 0, fl);
        }
      } else if (in_dim <= 16) {
        RCMARL_LAUNCH((k_minibatch_mx<1, false>), dim3(n_nets), dim3(64), MX_BYTES, stream, a, 0, fl);
      } else {
        RCMARL_LAUNCH((k_minibatch_mx<2, false>), dim3(n_nets), dim3(64), MX_BYTES, stream, a, 0, fl);
      }
    }
    // alone: four networks per workgroup; as the fix-up: one (64 threads, 16 KB of LDS: it finds room beside anything and returns at once)
    const int wpb = flags ? 1 : 4;
    const size_t smem_w = (size_t)wpb * WP_FLOATS * sizeof(float);
    if (in_dim <= 16) {
      RCMARL_LAUNCH((k_minibatch_wave<16>), dim3((n_nets + wpb - 1) / wpb), dim3(64 * wpb), smem_w, stream, a, n_nets, flags);
    } else {
      RCMARL_LAUNCH((k_minibatch_wave<20>), dim3((n_nets + wpb - 1) / wpb), dim3(64 * wpb), smem_w, stream, a, n_nets, flags);
    }
    return rcmarl_check_launch();
  }
  return mb_launch(k_minibatch_train<20, 1, false>, a, S, mb_smem_bytes(in_dim, 20, 1), stream);
}

RCMARL_EXPORT int rcmarl_minibatch_fit_multi(const rcmarl_mb_job* jobs, int njobs, int S, int N, int B, int hid, int ldb,
                                             int batch_size, int epochs, float lr, void* stream) {
  if (!jobs || njobs <= 0 || S <= 0 || N <= 0 || B <= 0 || batch_size <= 0 || epochs <= 0 || ldb < B) return RCMARL_ERR_ARG;
  if (njobs > MB_MAX_JOBS || hid != 20) return RCMARL_ERR_UNSUPPORTED;
  MbMulti m{};
  m.njobs = njobs;
  int total = 0, ks1 = 0;
  for (int j = 0; j < njobs; ++j) {
    const rcmarl_mb_job& q = jobs[j];
    if (!q.x || !q.theta || !q.agents || !q.y || !q.ovf_flags || q.n_adv <= 0 || q.in_dim <= 0 || (q.ldp & 63)) return RCMARL_ERR_ARG;
    if (q.in_dim > 20) return RCMARL_ERR_UNSUPPORTED;            // (wider inputs: one workgroup per network, rcmarl_minibatch_fit)
    const int k = q.in_dim <= 16 ? 1 : 2;
    if (ks1 && k != ks1) return RCMARL_ERR_UNSUPPORTED;           // one kernel form per launch: all jobs <= 16 inputs, or all 17..20
    ks1 = k;
    MbArgs& a = m.a[j];
    a.x = q.x; a.x_seed_stride = q.x_seed_stride; a.theta = q.theta; a.agents = q.agents; a.y = q.y; a.perm = q.perm;
    a.loss_out = q.loss_out; a.N = N; a.B = B; a.in_dim = q.in_dim; a.ldp = q.ldp; a.ldb = ldb;
    a.bs = batch_size < B ? batch_size : B; a.epochs = epochs; a.n_adv = q.n_adv; a.lr = lr;
    m.flags[j] = q.ovf_flags;
    m.first[j] = total;
    total += q.n_adv * S;
  }
  for (int j = njobs; j <= MB_MAX_JOBS; ++j) m.first[j] = total;
  const size_t smem_w = (size_t)WP_FLOATS * sizeof(float);
  static const bool attr_ok = rc_want_lds(k_minibatch_wave_multi<16>, smem_w) && rc_want_lds(k_minibatch_wave_multi<20>, smem_w);
  if (!attr_ok) return RCMARL_ERR_LAUNCH;
  const char* e = getenv("RCMARL_MB_MX");
  const bool mx = !(e && atoi(e) == 0);
  if (mx) {
    const char* ce = getenv("RCMARL_MB_MX_COMPACT");
    const bool compact = ce ? atoi(ce) != 0 : total > mx_slots();
    if (compact) {
      if (ks1 == 1) { RCMARL_LAUNCH((k_minibatch_mx_multi<1, true>), dim3(total), dim3(64), MX_CBYTES, stream, m); }
      else { RCMARL_LAUNCH((k_minibatch_mx_multi<2, true>), dim3(total), dim3(64), MX_CBYTES, stream, m); }
    } else if (ks1 == 1) {
      RCMARL_LAUNCH((k_minibatch_mx_multi<1, false>), dim3(total), dim3(64), MX_BYTES, stream, m);
    } else {
      RCMARL_LAUNCH((k_minibatch_mx_multi<2, false>), dim3(total), dim3(64), MX_BYTES, stream, m);
    }
  } else {
    for (int j = 0; j < njobs; ++j) m.flags[j] = nullptr;          // the fp32 kernel alone fits every network
  }
  // the fp32 wavefront kernel: the fix-up of flagged networks (returns at once otherwise), or alone with RCMARL_MB_MX=0
  if (ks1 == 1) { RCMARL_LAUNCH((k_minibatch_wave_multi<16>), dim3(total), dim3(64), smem_w, stream, m); }
  else { RCMARL_LAUNCH((k_minibatch_wave_multi<20>), dim3(total), dim3(64), smem_w, stream, m); }
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_minibatch_actor(const float* x, long x_seed_stride, float* theta, float* adam_m,
                                         float* adam_v, const int* agents, int n_adv, const float* act_t,
                                         const float* delta, const int* perm, int S, int N, int B, int in_dim,
                                         int hid, int n_actions, int ldp, int ldb, int batch_size, int epochs,
                                         double lr, double beta1, double beta2, double eps, int t0,
                                         float* loss_out, void* stream) {
  if (!x || !theta || !adam_m || !adam_v || !agents || !act_t || !delta || n_adv <= 0 || S <= 0 || N <= 0 || B <= 0 ||
      in_dim <= 0 || batch_size <= 0 || epochs <= 0 || t0 < 0 || (ldp & 63) || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid != 20 || n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  MbArgs a{};
  a.x = x; a.x_seed_stride = x_seed_stride; a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v; a.agents = agents;
  a.y = act_t; a.w = delta; a.perm = perm; a.loss_out = loss_out; a.N = N; a.B = B; a.in_dim = in_dim; a.ldp = ldp;
  a.ldb = ldb; a.bs = batch_size < B ? batch_size : B; a.epochs = epochs; a.n_adv = n_adv; a.t0 = t0; a.lr_d = lr;
  a.beta1 = beta1; a.beta2 = beta2;      // alpha and (1-beta) are formed in double, then rounded (Keras/TF2 order)
  a.one_m_b1 = (float)(1.0 - beta1); a.one_m_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
  return mb_launch(k_minibatch_train<20, 5, true>, a, S, mb_smem_bytes(in_dim, 20, 5), stream);
}
