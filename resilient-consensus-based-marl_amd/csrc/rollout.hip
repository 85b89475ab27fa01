// K8 + the single-row forwards of the rollout.
//
// Replaces, per environment step, the reference's N Keras `predict` calls on a
// batch of one (RPBCAC_agent.get_action, agents/resilient_CAC_agents.py:208-219;
// the reference's dominant wall-time cost) and the per-agent Python loop of
// Grid_World.step (environments/grid_world.py:47-64), and per episode the
// critic evaluations at the start state (training/train_agents.py:60-62).
//
// One wavefront owns one (seed, agent): lanes stride over the input features of
// layer 1 (W1 rows are 80-B contiguous -> fully coalesced), a butterfly sums the
// 20 partial pre-activations, the tiny layers 2-3 run redundantly in every lane
// with wave-uniform (scalar) weights.  Lane 0 then samples the action (Philox
// stream, rcmarl_rng.h) and applies the grid-world transition, writing this
// agent's slice of the replay row.  One launch per environment step for ALL
// seeds and agents; no host round trip.
#include "rcmarl_common.h"
#include "rcmarl_rng.h"

namespace {

struct EnvCfg {
  int nrow, ncol;
  double mean_x, mean_y, std_x, std_y;   // get_data() scaling (environments/grid_world.py:29-33,70)
};

template <int HID>
__device__ __forceinline__ void wave_hidden(const float* __restrict__ th, const NetGeom& g,
                                            const float* __restrict__ x /*[in_dim] state vector*/,
                                            float (&a2)[HID]) {
  const int lane = threadIdx.x & 63;
  float acc[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) acc[j] = 0.f;
  for (int k = lane; k < g.in_dim; k += 64) {
    const float xv = x[k];
    const float* w = th + (long)k * HID;
#pragma unroll
    for (int j = 0; j < HID; ++j) acc[j] = fmaf(xv, w[j], acc[j]);
  }
  float a1[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) a1[j] = rc_lrelu(rc_wave_sum(acc[j]) + th[g.o_b1 + j]);
#pragma unroll
  for (int k = 0; k < HID; ++k) a2[k] = 0.f;
#pragma unroll
  for (int j = 0; j < HID; ++j)
#pragma unroll
    for (int k = 0; k < HID; ++k) a2[k] = fmaf(a1[j], th[g.o_W2 + j * HID + k], a2[k]);
#pragma unroll
  for (int k = 0; k < HID; ++k) a2[k] = rc_lrelu(a2[k] + th[g.o_b2 + k]);
}

template <int HID, int A>
__device__ __forceinline__ void wave_policy(const float* __restrict__ th, const NetGeom& g,
                                            const float* __restrict__ x, float (&p)[A]) {
  float a2[HID];
  wave_hidden<HID>(th, g, x, a2);
  float mx = -3.0e38f;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < HID; ++k) acc = fmaf(a2[k], th[g.o_W3 + k * A + a], acc);
    p[a] = acc + th[g.o_b3 + a];
    mx = fmaxf(mx, p[a]);
  }
  float se = 0.f;
#pragma unroll
  for (int a = 0; a < A; ++a) { p[a] = expf(p[a] - mx); se += p[a]; }
#pragma unroll
  for (int a = 0; a < A; ++a) p[a] = p[a] / se;
}

// one grid-world transition of one agent (environments/grid_world.py:52-64 with the
// dist_to_agents branch being dead: an agent always "collides" with itself)
__device__ __forceinline__ void env_transition(const EnvCfg& c, int a, int px, int py, int gx, int gy, int& nx,
                                               int& ny, int& reward_int) {
  const int dist = abs(px - gx) + abs(py - gy);            // distance BEFORE the move
  const int mx = (a == 1) ? -1 : (a == 2 ? 1 : 0);
  const int my = (a == 3) ? -1 : (a == 4 ? 1 : 0);
  nx = min(max(px + mx, 0), c.nrow - 1);
  ny = min(max(py + my, 0), c.nrow - 1);                   // nrow-1 on BOTH axes (:55)
  reward_int = (dist == 0 && a == 0) ? 0 : -dist - 1;
}

struct Replay {
  float* s; float* ns; float* sa; float* a; float* r;      // [S][cap][2N|2N|3N|N|N]
  long cap;
};

__device__ __forceinline__ void record_step(const EnvCfg& c, const Replay& rp, int S_idx, int N, int i, long row,
                                            int px, int py, int nx, int ny, int a, int reward_int,
                                            int* __restrict__ pos_next, float* __restrict__ xs_next,
                                            double* __restrict__ ret, double gpow) {
  const float sx = (float)(((double)px - c.mean_x) / c.std_x), sy = (float)(((double)py - c.mean_y) / c.std_y);
  const float tx = (float)(((double)nx - c.mean_x) / c.std_x), ty = (float)(((double)ny - c.mean_y) / c.std_y);
  const double rew = (double)reward_int / 5.0;             // get_data(): reward / 5 (:71)
  const long base = (long)S_idx * rp.cap + row;
  rp.s[base * 2 * N + 2 * i] = sx;  rp.s[base * 2 * N + 2 * i + 1] = sy;
  rp.ns[base * 2 * N + 2 * i] = tx; rp.ns[base * 2 * N + 2 * i + 1] = ty;
  rp.sa[base * 3 * N + 3 * i] = sx; rp.sa[base * 3 * N + 3 * i + 1] = sy; rp.sa[base * 3 * N + 3 * i + 2] = (float)a;
  rp.a[base * N + i] = (float)a;
  rp.r[base * N + i] = (float)rew;
  pos_next[((long)S_idx * N + i) * 2] = nx; pos_next[((long)S_idx * N + i) * 2 + 1] = ny;
  xs_next[(long)S_idx * 2 * N + 2 * i] = tx; xs_next[(long)S_idx * 2 * N + 2 * i + 1] = ty;
  ret[(long)S_idx * N + i] += rew * gpow;                  // ep_returns += reward*gamma**j (train_agents.py:71)
}

// probs[s][i][:] = actor_i(xs[s])
template <int HID, int A>
__global__ __launch_bounds__(256) void k_policy_probs(const float* __restrict__ xs, const float* __restrict__ theta,
                                                      float* __restrict__ probs, int N, int in_dim, int ldp) {
  const int s = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const NetGeom g = make_geom(in_dim, HID, A);
  float p[A];
  wave_policy<HID, A>(theta + ((long)s * N + i) * ldp, g, xs + (long)s * in_dim, p);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < A; ++a) probs[((long)s * N + i) * A + a] = p[a];
  }
}

// out[s][i] = value_i(xs[s])  (linear head, out_dim 1)
template <int HID>
__global__ __launch_bounds__(256) void k_value_rows(const float* __restrict__ xs, const float* __restrict__ theta,
                                                    float* __restrict__ out, int N, int in_dim, int ldp) {
  const int s = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a2[HID];
  wave_hidden<HID>(th, g, xs + (long)s * in_dim, a2);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) v = fmaf(a2[k], th[g.o_W3 + k], v);
  if ((threadIdx.x & 63) == 0) out[(long)s * N + i] = v + th[g.o_b3];
}

// fused step, device RNG: actor forward -> sample -> transition -> replay/returns
template <int HID, int A>
__global__ __launch_bounds__(256) void k_rollout_step(const float* __restrict__ xs, const int* __restrict__ pos,
                                                      const int* __restrict__ goal, const float* __restrict__ theta,
                                                      const unsigned long long* __restrict__ seeds, EnvCfg cfg,
                                                      Replay rp, long row, int* __restrict__ pos_next,
                                                      float* __restrict__ xs_next, double* __restrict__ ret,
                                                      double gpow, int episode, int step, float mu, int N, int in_dim,
                                                      int ldp, int* __restrict__ act_out) {
  const int s = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const NetGeom g = make_geom(in_dim, HID, A);
  float p[A];
  wave_policy<HID, A>(theta + ((long)s * N + i) * ldp, g, xs + (long)s * in_dim, p);
  if ((threadIdx.x & 63) != 0) return;
  const unsigned long long key = seeds[s];
  const RcPhilox rn = rc_philox4x32_10((uint32_t)i, (uint32_t)step, (uint32_t)episode, 0u, (uint32_t)key,
                                       (uint32_t)(key >> 32));
  const int a_rand = rc_mulhi_range(rn.r0, A);
  const float u1 = rc_u01(rn.r1), u2 = rc_u01(rn.r2);
  float c = 0.f;
  int a_pol = 0;
#pragma unroll
  for (int a = 0; a < A - 1; ++a) { c += p[a]; a_pol += (u1 >= c) ? 1 : 0; }
  const int act = (u2 < 1.0f - mu) ? a_pol : a_rand;
  const long pi = ((long)s * N + i) * 2;
  int nx, ny, rew;
  env_transition(cfg, act, pos[pi], pos[pi + 1], goal[pi], goal[pi + 1], nx, ny, rew);
  record_step(cfg, rp, s, N, i, row, pos[pi], pos[pi + 1], nx, ny, act, rew, pos_next, xs_next, ret, gpow);
  if (act_out) act_out[(long)s * N + i] = act;
}

// host-sampled actions (rng_mode='numpy'): transition + replay/returns only
__global__ __launch_bounds__(256) void k_env_apply(const int* __restrict__ pos, const int* __restrict__ goal,
                                                   const int* __restrict__ actions, EnvCfg cfg, Replay rp, long row,
                                                   int* __restrict__ pos_next, float* __restrict__ xs_next,
                                                   double* __restrict__ ret, double gpow, int S, int N) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)S * N) return;
  const int s = (int)(t / N), i = (int)(t - (long)s * N);
  int nx, ny, rew;
  env_transition(cfg, actions[t], pos[t * 2], pos[t * 2 + 1], goal[t * 2], goal[t * 2 + 1], nx, ny, rew);
  record_step(cfg, rp, s, N, i, row, pos[t * 2], pos[t * 2 + 1], nx, ny, actions[t], rew, pos_next, xs_next, ret, gpow);
}

// episode start: positions (device RNG or copied from `pos_in`), scaled state vector, returns = 0
__global__ __launch_bounds__(256) void k_env_reset(const int* __restrict__ pos_in,
                                                   const unsigned long long* __restrict__ seeds, EnvCfg cfg,
                                                   int episode, int* __restrict__ pos, float* __restrict__ xs,
                                                   double* __restrict__ ret, int S, int N) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)S * N) return;
  const int s = (int)(t / N), i = (int)(t - (long)s * N);
  int px, py;
  if (pos_in) {
    px = pos_in[t * 2]; py = pos_in[t * 2 + 1];
  } else {
    const unsigned long long key = seeds[s];
    const RcPhilox rn = rc_philox4x32_10((uint32_t)i, 0u, (uint32_t)episode, 1u, (uint32_t)key, (uint32_t)(key >> 32));
    px = rc_mulhi_range(rn.r0, cfg.nrow); py = rc_mulhi_range(rn.r1, cfg.ncol);
  }
  pos[t * 2] = px; pos[t * 2 + 1] = py;
  xs[t * 2] = (float)(((double)px - cfg.mean_x) / cfg.std_x);
  xs[t * 2 + 1] = (float)(((double)py - cfg.mean_y) / cfg.std_y);
  ret[t] = 0.0;
}


// ---------------------------------------------------------------------------------------------
// Episode-parallel rollout (rng_mode 'device').
//
// Between two update blocks every network is frozen (training/train_agents.py:86: updates happen
// only after n_ep_fixed episodes) and, with the counter-based Philox stream, every draw is a pure
// function of (seed, episode, step, agent).  The n_ep_fixed episodes of a block are therefore
// independent and are stepped TOGETHER: one lane = one episode, one wavefront = one (seed, agent),
// so the agent's actor weights are read ONCE per environment step for all E episodes (wave-uniform
// scalar loads) instead of once per episode, and a block needs max_ep_len launches instead of
// n_ep_fixed*max_ep_len.  State is kept episode-minor so lanes are coalesced:
//   xsT[S][2N][EP]   scaled global state, feature-major      posT[S][N][2][EP]   integer cells
//   retT[S][N][EP]   discounted returns (double)             EP = E rounded up to 64
template <int HID>
__device__ __forceinline__ void lane_hidden(const float* __restrict__ th, const NetGeom& g,
                                            const float* __restrict__ xT /*[in_dim][EP]*/, int EP, int e,
                                            float (&a2)[HID]) {
  float acc[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int k = 0; k < g.in_dim; ++k) {
    const float xv = xT[(long)k * EP + e];
    const float* w = th + (long)k * HID;
#pragma unroll
    for (int j = 0; j < HID; ++j) acc[j] = fmaf(xv, w[j], acc[j]);
  }
  float a1[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) a1[j] = rc_lrelu(acc[j] + th[g.o_b1 + j]);
#pragma unroll
  for (int k = 0; k < HID; ++k) a2[k] = 0.f;
#pragma unroll
  for (int j = 0; j < HID; ++j)
#pragma unroll
    for (int k = 0; k < HID; ++k) a2[k] = fmaf(a1[j], th[g.o_W2 + j * HID + k], a2[k]);
#pragma unroll
  for (int k = 0; k < HID; ++k) a2[k] = rc_lrelu(a2[k] + th[g.o_b2 + k]);
}

template <int HID, int A>
__global__ __launch_bounds__(256) void k_rollout_step_ep(const float* __restrict__ xsT, const int* __restrict__ posT,
                                                         const int* __restrict__ goal,
                                                         const float* __restrict__ theta,
                                                         const unsigned long long* __restrict__ seeds, EnvCfg cfg,
                                                         Replay rp, long row0, int ep_len, int* __restrict__ posT_next,
                                                         float* __restrict__ xsT_next, double* __restrict__ retT,
                                                         double gpow, int episode0, int step, float mu, int N, int E,
                                                         int EP, int ldp) {
  const int s = blockIdx.y;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (i >= N) return;
  const int e = blockIdx.z * 64 + (threadIdx.x & 63);
  const int in_dim = 2 * N;
  const NetGeom g = make_geom(in_dim, HID, A);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a2[HID];
  lane_hidden<HID>(th, g, xsT + (long)s * in_dim * EP, EP, e, a2);
  float p[A];
  float mx = -3.0e38f;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < HID; ++k) acc = fmaf(a2[k], th[g.o_W3 + k * A + a], acc);
    p[a] = acc + th[g.o_b3 + a];
    mx = fmaxf(mx, p[a]);
  }
  float se = 0.f;
#pragma unroll
  for (int a = 0; a < A; ++a) { p[a] = expf(p[a] - mx); se += p[a]; }
#pragma unroll
  for (int a = 0; a < A; ++a) p[a] = p[a] / se;
  if (e >= E) return;
  const unsigned long long key = seeds[s];
  const RcPhilox rn = rc_philox4x32_10((uint32_t)i, (uint32_t)step, (uint32_t)(episode0 + e), 0u, (uint32_t)key,
                                       (uint32_t)(key >> 32));
  const int a_rand = rc_mulhi_range(rn.r0, A);
  const float u1 = rc_u01(rn.r1), u2 = rc_u01(rn.r2);
  float c = 0.f;
  int a_pol = 0;
#pragma unroll
  for (int a = 0; a < A - 1; ++a) { c += p[a]; a_pol += (u1 >= c) ? 1 : 0; }
  const int act = (u2 < 1.0f - mu) ? a_pol : a_rand;
  const long pi = (((long)s * N + i) * 2) * EP + e;
  const int px = posT[pi], py = posT[pi + EP];
  const long gi = ((long)s * N + i) * 2;
  int nx, ny, rew;
  env_transition(cfg, act, px, py, goal[gi], goal[gi + 1], nx, ny, rew);
  const float sx = (float)(((double)px - cfg.mean_x) / cfg.std_x), sy = (float)(((double)py - cfg.mean_y) / cfg.std_y);
  const float tx = (float)(((double)nx - cfg.mean_x) / cfg.std_x), ty = (float)(((double)ny - cfg.mean_y) / cfg.std_y);
  const double rw = (double)rew / 5.0;
  const long base = (long)s * rp.cap + row0 + (long)e * ep_len + step;     // replay row of (episode e, step)
  rp.s[base * 2 * N + 2 * i] = sx;  rp.s[base * 2 * N + 2 * i + 1] = sy;
  rp.ns[base * 2 * N + 2 * i] = tx; rp.ns[base * 2 * N + 2 * i + 1] = ty;
  rp.sa[base * 3 * N + 3 * i] = sx; rp.sa[base * 3 * N + 3 * i + 1] = sy; rp.sa[base * 3 * N + 3 * i + 2] = (float)act;
  rp.a[base * N + i] = (float)act;
  rp.r[base * N + i] = (float)rw;
  posT_next[pi] = nx; posT_next[pi + EP] = ny;
  xsT_next[((long)s * in_dim + 2 * i) * EP + e] = tx;
  xsT_next[((long)s * in_dim + 2 * i + 1) * EP + e] = ty;
  retT[((long)s * N + i) * EP + e] += rw * gpow;
}

// est[(e*S + s)*N + i] = critic_i(start state of episode e)      (training/train_agents.py:60-62)
template <int HID>
__global__ __launch_bounds__(256) void k_value_rows_ep(const float* __restrict__ xsT, const float* __restrict__ theta,
                                                       float* __restrict__ est, int S, int N, int E, int EP, int ldp) {
  const int s = blockIdx.y;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (i >= N) return;
  const int e = blockIdx.z * 64 + (threadIdx.x & 63);
  const int in_dim = 2 * N;
  const NetGeom g = make_geom(in_dim, HID, 1);
  const float* th = theta + ((long)s * N + i) * ldp;
  float a2[HID];
  lane_hidden<HID>(th, g, xsT + (long)s * in_dim * EP, EP, e, a2);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < HID; ++k) v = fmaf(a2[k], th[g.o_W3 + k], v);
  if (e < E) est[((long)e * S + s) * N + i] = v + th[g.o_b3];
}

__global__ __launch_bounds__(256) void k_env_reset_ep(const int* __restrict__ pos_in,
                                                      const unsigned long long* __restrict__ seeds, EnvCfg cfg,
                                                      int episode0, int* __restrict__ posT, float* __restrict__ xsT,
                                                      double* __restrict__ retT, int S, int N, int E, int EP) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)S * N * EP) return;
  const int e = (int)(t % EP);
  const long si = t / EP;
  const int s = (int)(si / N), i = (int)(si - (long)s * N);
  int px = 0, py = 0;
  if (e < E) {
    if (pos_in) {
      px = pos_in[si * 2]; py = pos_in[si * 2 + 1];
    } else {
      const unsigned long long key = seeds[s];
      const RcPhilox rn = rc_philox4x32_10((uint32_t)i, 0u, (uint32_t)(episode0 + e), 1u, (uint32_t)key, (uint32_t)(key >> 32));
      px = rc_mulhi_range(rn.r0, cfg.nrow); py = rc_mulhi_range(rn.r1, cfg.ncol);
    }
  }
  posT[(si * 2) * EP + e] = px; posT[(si * 2 + 1) * EP + e] = py;
  xsT[((long)s * 2 * N + 2 * i) * EP + e] = (float)(((double)px - cfg.mean_x) / cfg.std_x);
  xsT[((long)s * 2 * N + 2 * i + 1) * EP + e] = (float)(((double)py - cfg.mean_y) / cfg.std_y);
  retT[si * EP + e] = 0.0;
}

}  // namespace

#define RC_HID_SWITCH(hid, STMT)               \
  switch (hid) {                               \
    case 20: { constexpr int HID_ = 20; STMT; } break; \
    default: return RCMARL_ERR_UNSUPPORTED;    \
  }

RCMARL_EXPORT int rcmarl_policy_probs(const float* xs, const float* theta, float* probs, int S, int N, int in_dim,
                                      int hid, int n_actions, int ldp, void* stream) {
  if (!xs || !theta || !probs || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63)) return RCMARL_ERR_ARG;
  if (n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  const dim3 grid(rc_ceil_div(N, 4), S), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_policy_probs<HID_, 5>), grid, block, 0, stream, xs, theta, probs, N, in_dim, ldp));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_value_rows(const float* xs, const float* theta, float* out, int S, int N, int in_dim, int hid,
                                    int ldp, void* stream) {
  if (!xs || !theta || !out || S <= 0 || N <= 0 || in_dim <= 0 || (ldp & 63)) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(N, 4), S), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_value_rows<HID_>), grid, block, 0, stream, xs, theta, out, N, in_dim, ldp));
  return rcmarl_check_launch();
}

// env_cfg = {nrow, ncol}; scale = {mean_x, mean_y, std_x, std_y}
RCMARL_EXPORT int rcmarl_rollout_step(const float* xs, const int* pos, const int* goal, const float* theta,
                                      const unsigned long long* seeds, int nrow, int ncol, const double* scale,
                                      float* rp_s, float* rp_ns, float* rp_sa, float* rp_a, float* rp_r, long cap,
                                      long row, int* pos_next, float* xs_next, double* ret, double gpow, int episode,
                                      int step, float mu, int S, int N, int hid, int n_actions, int ldp,
                                      int* act_out, void* stream) {
  if (!xs || !pos || !goal || !theta || !seeds || !scale || !rp_s || !rp_ns || !rp_sa || !rp_a || !rp_r ||
      !pos_next || !xs_next || !ret || row < 0 || row >= cap || S <= 0 || N <= 0 || (ldp & 63))
    return RCMARL_ERR_ARG;
  if (n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  const EnvCfg cfg{nrow, ncol, scale[0], scale[1], scale[2], scale[3]};
  const Replay rp{rp_s, rp_ns, rp_sa, rp_a, rp_r, cap};
  const dim3 grid(rc_ceil_div(N, 4), S), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_rollout_step<HID_, 5>), grid, block, 0, stream, xs, pos, goal, theta, seeds, cfg,
                                   rp, row, pos_next, xs_next, ret, gpow, episode, step, mu, N, 2 * N, ldp, act_out));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_env_apply(const int* pos, const int* goal, const int* actions, int nrow, int ncol,
                                   const double* scale, float* rp_s, float* rp_ns, float* rp_sa, float* rp_a,
                                   float* rp_r, long cap, long row, int* pos_next, float* xs_next, double* ret,
                                   double gpow, int S, int N, void* stream) {
  if (!pos || !goal || !actions || !scale || !rp_s || !rp_ns || !rp_sa || !rp_a || !rp_r || !pos_next || !xs_next ||
      !ret || row < 0 || row >= cap || S <= 0 || N <= 0)
    return RCMARL_ERR_ARG;
  const EnvCfg cfg{nrow, ncol, scale[0], scale[1], scale[2], scale[3]};
  const Replay rp{rp_s, rp_ns, rp_sa, rp_a, rp_r, cap};
  const dim3 grid(rc_ceil_div(S * N, 256)), block(256);
  RCMARL_LAUNCH(k_env_apply, grid, block, 0, stream, pos, goal, actions, cfg, rp, row, pos_next, xs_next, ret, gpow, S, N);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_env_reset(const int* pos_in, const unsigned long long* seeds, int nrow, int ncol,
                                   const double* scale, int episode, int* pos, float* xs, double* ret, int S, int N,
                                   void* stream) {
  if ((!pos_in && !seeds) || !scale || !pos || !xs || !ret || S <= 0 || N <= 0) return RCMARL_ERR_ARG;
  const EnvCfg cfg{nrow, ncol, scale[0], scale[1], scale[2], scale[3]};
  const dim3 grid(rc_ceil_div(S * N, 256)), block(256);
  RCMARL_LAUNCH(k_env_reset, grid, block, 0, stream, pos_in, seeds, cfg, episode, pos, xs, ret, S, N);
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_rollout_step_episodes(const float* xsT, const int* posT, const int* goal, const float* theta,
                                               const unsigned long long* seeds, int nrow, int ncol, const double* scale,
                                               float* rp_s, float* rp_ns, float* rp_sa, float* rp_a, float* rp_r,
                                               long cap, long row0, int ep_len, int* posT_next, float* xsT_next,
                                               double* retT, double gpow, int episode0, int step, float mu, int S, int N,
                                               int E, int EP, int hid, int n_actions, int ldp, void* stream) {
  if (!xsT || !posT || !goal || !theta || !seeds || !scale || !rp_s || !rp_ns || !rp_sa || !rp_a || !rp_r ||
      !posT_next || !xsT_next || !retT || row0 < 0 || E <= 0 || EP < E || (EP & 63) || ep_len <= 0 || step < 0 ||
      step >= ep_len || row0 + (long)E * ep_len > cap || S <= 0 || N <= 0 || (ldp & 63))
    return RCMARL_ERR_ARG;
  if (n_actions != 5) return RCMARL_ERR_UNSUPPORTED;
  const EnvCfg cfg{nrow, ncol, scale[0], scale[1], scale[2], scale[3]};
  const Replay rp{rp_s, rp_ns, rp_sa, rp_a, rp_r, cap};
  const dim3 grid(rc_ceil_div(N, 4), S, EP / 64), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_rollout_step_ep<HID_, 5>), grid, block, 0, stream, xsT, posT, goal, theta, seeds,
                                   cfg, rp, row0, ep_len, posT_next, xsT_next, retT, gpow, episode0, step, mu, N, E, EP,
                                   ldp));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_value_rows_episodes(const float* xsT, const float* theta, float* est, int S, int N, int E, int EP,
                                             int hid, int ldp, void* stream) {
  if (!xsT || !theta || !est || S <= 0 || N <= 0 || E <= 0 || EP < E || (EP & 63) || (ldp & 63)) return RCMARL_ERR_ARG;
  const dim3 grid(rc_ceil_div(N, 4), S, EP / 64), block(256);
  RC_HID_SWITCH(hid, RCMARL_LAUNCH((k_value_rows_ep<HID_>), grid, block, 0, stream, xsT, theta, est, S, N, E, EP, ldp));
  return rcmarl_check_launch();
}

RCMARL_EXPORT int rcmarl_env_reset_episodes(const int* pos_in, const unsigned long long* seeds, int nrow, int ncol,
                                            const double* scale, int episode0, int* posT, float* xsT, double* retT,
                                            int S, int N, int E, int EP, void* stream) {
  if ((!pos_in && !seeds) || !scale || !posT || !xsT || !retT || S <= 0 || N <= 0 || E <= 0 || EP < E || (EP & 63))
    return RCMARL_ERR_ARG;
  const EnvCfg cfg{nrow, ncol, scale[0], scale[1], scale[2], scale[3]};
  const long total = (long)S * N * EP;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  RCMARL_LAUNCH(k_env_reset_ep, grid, block, 0, stream, pos_in, seeds, cfg, episode0, posT, xsT, retT, S, N, E, EP);
  return rcmarl_check_launch();
}
