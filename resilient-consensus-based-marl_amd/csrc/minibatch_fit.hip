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
#include "rcmarl_common.h"

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

size_t mb_smem_bytes(int in_dim, int hid, int out) {
  const NetGeom g = make_geom(in_dim, hid, out);
  const int Ppad = (g.P + 3) & ~3;
  return sizeof(float) * ((size_t)2 * Ppad + TR * XLD + 4 * TR * hid + TR * out + TR) + sizeof(int) * TR;
}

template <class K>
int mb_launch(K kernel, const MbArgs& a, int S, size_t smem, void* stream) {
  if (smem > 160 * 1024) return RCMARL_ERR_UNSUPPORTED;
#ifndef RCMARL_EMU
  if (smem > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return RCMARL_ERR_LAUNCH;
#endif
  const dim3 grid(a.n_adv, S), block(256);
  RCMARL_LAUNCH(kernel, grid, block, smem, stream, a);
  return rcmarl_check_launch();
}

}  // namespace

RCMARL_EXPORT int rcmarl_minibatch_fit(const float* x, long x_seed_stride, float* theta, const int* agents, int n_adv,
                                       const float* y, const int* perm, int S, int N, int B, int in_dim, int hid,
                                       int ldp, int ldb, int batch_size, int epochs, float lr, float* loss_out,
                                       void* stream) {
  if (!x || !theta || !agents || !y || n_adv <= 0 || S <= 0 || N <= 0 || B <= 0 || in_dim <= 0 || batch_size <= 0 ||
      epochs <= 0 || (ldp & 63) || ldb < B)
    return RCMARL_ERR_ARG;
  if (hid != 20) return RCMARL_ERR_UNSUPPORTED;
  MbArgs a{};
  a.x = x; a.x_seed_stride = x_seed_stride; a.theta = theta; a.agents = agents; a.y = y; a.perm = perm;
  a.loss_out = loss_out; a.N = N; a.B = B; a.in_dim = in_dim; a.ldp = ldp; a.ldb = ldb;
  a.bs = batch_size < B ? batch_size : B; a.epochs = epochs; a.n_adv = n_adv; a.lr = lr;
  return mb_launch(k_minibatch_train<20, 1, false>, a, S, mb_smem_bytes(in_dim, 20, 1), stream);
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
