/* rcmarl.h -- C-ABI of the MI355X-native RPBCAC hot path (librcmarl_hip.so).
 *
 * The reference (mfigura/Resilient-consensus-based-MARL) has NO native/FFI
 * boundary: its hot path is Python calling TensorFlow/Keras.  This header is
 * the boundary a maintainer would bind instead (ctypes stub in
 * INTEGRATION.md; rcmarl_amd/capi.py is that binding).  Each entry point
 * names the reference code it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain pointers + sizes, no framework types.  All pointers are DEVICE
 *     pointers unless stated; `stream` is a hipStream_t (NULL = default).
 *   - every function only enqueues work on `stream`: no allocation, no
 *     synchronisation.  Returns 0 or an RCMARL_ERR_* code; never throws.
 *   - parameter matrices theta[S][N][ldp] (fp32): one row per (seed, agent),
 *     row = [W1(in x hid) | b1 | W2(hid x hid) | b2 | W3(hid x out) | b3 | pad]
 *     in Keras order (main.py:59-82), ldp % 64 == 0.
 *   - replay tensors X[S][rows][width] fp32 row-major; `x` may point at a row
 *     window, `x_seed_stride` (elements) is the distance between seeds.
 *   - feature-major activations a1t[S][N*hid][ldb], ldb % 64 == 0, ldb >= B.
 *   - agent-major per-sample vectors y[S][N][ldb].
 *   - S seeds, N agents, B replay rows, d in-neighbourhood size incl. self,
 *     H trim parameter, hid hidden width (compiled: 20), out = 1 or n_actions.
 */
#ifndef RCMARL_H
#define RCMARL_H

#ifdef __cplusplus
extern "C" {
#endif

#define RCMARL_OK 0
#define RCMARL_ERR_ARG 1          /* bad argument (null pointer, misaligned ld, d < 2H+1, ...) */
#define RCMARL_ERR_LAUNCH 2       /* hipGetLastError() != hipSuccess after the launch */
#define RCMARL_ERR_UNSUPPORTED 3  /* shape outside the compiled kernels (hid != 20, n_actions != 5, ...) */

int rcmarl_abi_version(void);                          /* 4 (round 6): ABI 3 (= ABI 2 minus the fused local-fit prototypes, plus
                                                        * rcmarl_minibatch_fit_multi and rcmarl_lattice_forget) plus the packed-operand
                                                        * dense layers of wide networks, rcmarl_pk_* */
int rcmarl_mb_job_layout(int what);                    /* sizeof(rcmarl_mb_job) (what = 0) / offset of its field number `what` (1 = x_seed_stride
                                                        * ... 10 = ovf_flags), -1 otherwise: a binding checks its own declaration against it */
int rcmarl_fit_partial_size(int hid);                  /* floats per partial record of rcmarl_mid_fit */
int rcmarl_actor_partial_size(int hid, int n_actions); /* floats per partial record of rcmarl_mid_actor */
int rcmarl_rows_per_chunk(void);                       /* replay rows per workgroup (256); nchunk = ceil(B/256) */

/* K1: resilient consensus over hidden-layer parameters.
 * Replaces RPBCAC_agent.resilient_consensus_critic_hidden/_TR_hidden
 * (agents/resilient_CAC_agents.py:142-166) + _resilient_aggregation (:42-58) for ALL
 * cooperative agents of all seeds (loop at training/train_agents.py:125-135).
 * theta[s][i][p] <- mean_k clamp(msg[s][nbr[i][k]][p], lower, upper), p < P_hid, coop[i] != 0.
 * nbr: int[N][d] with nbr[i][0] == i; coop: int32[N] (0/1 flags).  lo_dbg/hi_dbg (both or neither,
 * [S][N][ldp]) receive the clip window for bit-exact tests. */
int rcmarl_consensus_params(const float* msg, float* theta, const int* nbr, const int* coop, int S, int N,
                            int ldp, int P_hid, int d, int H, float* lo_dbg, float* hi_dbg, void* stream);
/* The same aggregation for a CIRCULANT in-graph, in_nodes[i] = [i, i+1, .., i+d-1 mod N] (the reference's own
 * pattern, main.py:28; the caller guarantees it) with d == 2H+2: G consecutive agents share d-G+1 of their inputs,
 * so one selection network over the shared values serves G agents (csrc/consensus_params.hip).  Bit-identical to
 * rcmarl_consensus_params.  ..._supported() tells whether a kernel is compiled for (N, d, H); otherwise the call
 * returns RCMARL_ERR_UNSUPPORTED and the general entry point must be used. */
int rcmarl_consensus_params_circulant_supported(int N, int d, int H);
int rcmarl_consensus_params_circulant(const float* msg, float* theta, const int* coop, int S, int N, int ldp, int P_hid,
                                      int d, int H, float* lo_dbg, float* hi_dbg, void* stream);

/* K4 layer 1 forward (shared input => one GEMM per seed), Keras Dense + LeakyReLU(0.1):
 * a1t[s][n*hid+j][b] = lrelu(sum_k x[s][b][k]*W1[s][n][k][j] + b1[s][n][j]).
 * Replaces the first Dense of every model call at agents/resilient_CAC_agents.py:66,79,95-97,114,181,201. */
int rcmarl_layer1_forward(const float* x, long x_seed_stride, const float* theta, float* a1t, int S, int N, int B,
                          int in_dim, int hid, int ldp, int ldb, void* stream);

/* K5 layer 1 backward + optimizer: W1[s][n][k][j] -= lr * sum_b x[s][b][k]*dz1t[s][n*hid+j][b]
 * (inside critic.fit / TR.fit, agents/resilient_CAC_agents.py:118,136); mask: int32[N] flags or NULL. */
int rcmarl_layer1_backward_sgd(const float* x, long x_seed_stride, const float* dz1t, float* theta,
                               const int* mask, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                               float lr, void* stream);
/* same with the Adam rule of actor.train_on_batch (agents/resilient_CAC_agents.py:38,99):
 * m += (g-m)(1-b1); v += (g*g-v)(1-b2); w -= alpha*m/(sqrt(v)+eps), alpha = lr*sqrt(1-b2^t)/(1-b1^t). */
int rcmarl_layer1_backward_adam(const float* x, long x_seed_stride, const float* dz1t, float* theta, float* adam_m,
                                float* adam_v, const int* mask, int S, int N, int B, int in_dim, int hid,
                                int ldp, int ldb, float alpha, float one_m_b1, float one_m_b2, float eps, void* stream);

/* K4/K5 layers 2-3 of one full-batch SGD step of critic.fit / TR.fit with MSE loss
 * (agents/resilient_CAC_agents.py:116-118,134-136): forward, loss vs y[s][n][b], backward.
 * a1t is overwritten IN PLACE by dz1 (feature-major); partial gradient records go to
 * partials[S][N][nrec][rcmarl_fit_partial_size(hid)] = [gW2|gb2|gW3|gb3|gb1|sum (v-y)^2], one record per workgroup of the
 * launch (nrec <= nchunk = ceil(B / rcmarl_rows_per_chunk()): size the buffer for nchunk records); rcmarl_small_sgd with the
 * same (S, N, B) reads exactly the records this call wrote. */
int rcmarl_mid_fit(float* a1t, const float* theta, const float* y, float* partials, int S, int N, int B, int in_dim,
                   int hid, int ldp, int ldb, void* stream);
/* ---- lattice (exact bf16x3) form of the layer-1 GEMMs: csrc/lattice_gemm.hip, csrc/rcmarl_lattice.h ----
 * The grid-world state/action columns are alpha_c * (small integer) (environments/grid_world.py:66-72,
 * training/train_agents.py:89-93), so X*W1 and X^T*dZ1 can run on the bf16 matrix core with every
 * product exact: integers K (bf16) times the three bf16 pieces of a fp32 value, fp32 accumulation.
 * Same results as the f32 entry points above up to fp32 summation order (tests: 1e-6 relative).
 * Packed bf16 operands ("PK", rcmarl_lattice.h): 8-KiB blocks [rows/128][kt][pieces][128][32];
 * every *_rt / *_kt argument is the ALLOCATED number of 128-row tiles / 32-deep k-tiles of that buffer
 * (bytes per seed = rt*kt*pieces*8192).
 *
 * rcmarl_lattice_encode: K = x/alpha (verified integer, |K| <= 256, else *flag = 1) packed as
 *   kp  (rows = replay row, reduction = feature; forward operand)  and/or
 *   ktp (rows = feature,    reduction = replay row; backward operand); either may be NULL.
 *   Rows/features up to the next multiple of 256 replay rows / of the buffer extents are zero-filled.
 *
 * Operand form (rcmarl_lattice_f16_mode(), env RCMARL_LAT_F16, default 3): bit 0 -- the forward operand (wp, and kp) --
 *   and bit 1 -- the backward operand (dzp, and ktp) -- are carried as TWO f16 pieces of the value scaled by a fixed
 *   power of two (2^10 alpha W1; 2^8 dz1) instead of three bf16 pieces: the value up to one unit in the last place of
 *   its fp32 significand (exact for 3 in 4), two matrix passes and 4 bytes per value instead of three and 6.  The
 *   epilogues multiply the scale out; finite while |alpha W1| < 64 and |dz1| < 256.  0 = three exact bf16 pieces
 *   everywhere.  Buffers are sized for three pieces in either form; producer and consumer calls must see the same mode. */
int rcmarl_lattice_f16_mode(void);
/* The form is read from RCMARL_LAT_F16 ONCE, at the first call that needs it; a process that switches forms (tests, a benchmark
 * that reports both) says so here: mode 0..3, or -1 = read the environment again.  Host-side state only.  A packed buffer remembers
 * the form it was last written in (by base pointer): handing a consumer (rcmarl_layer1_forward_lattice,
 * rcmarl_layer1_backward_sgd_lattice) a buffer written in the other form returns RCMARL_ERR_ARG. */
int rcmarl_lattice_set_f16_mode(int mode);
/* Drop the recorded form of a packed buffer the caller is about to free or re-purpose (a recycled device address must not inherit
 * it).  Host-side state only; unknown pointers are fine. */
int rcmarl_lattice_forget(const void* buf);
int rcmarl_lattice_encode(const float* x, long x_seed_stride, const float* alpha, int S, int B, int in_dim, void* kp,
                          int kp_rt, int kp_kt, void* ktp, int ktp_rt, int ktp_kt, int* flag, void* stream);
/* wp (rows = (agent,unit) column, reduction = feature) <- the pieces of alpha[k]*W1[s][n][k][j] in the current operand form
 * (two f16 pieces of 2^10 times it, or three exact bf16 pieces) */
int rcmarl_w1_split(const float* theta, const float* alpha, void* wp, int S, int N, int in_dim, int hid, int ldp,
                    int wp_rt, int wp_kt, void* stream);
/* dzp (rows = (agent,unit) column, reduction = replay row, zero beyond B) <- the pieces (current operand form) of the fp32
 * feature-major dz[S][N*hid][ldb]: the backward operand when dz1 was produced by rcmarl_dense_backward_data */
int rcmarl_lattice_pack_dz(const float* dz, void* dzp, int S, int N, int B, int hid, int ldb, int dzp_rt, int dzp_kt,
                           void* stream);
/* rcmarl_lattice_pack_dz and, in the same pass over dz, its row sums (= rcmarl_wide_bias_grad: gb1 of a wide net):
 * sums[(s*N + n)*sums_ld + sums_off + j] = sum_b dz[s][n*hid + j][b].  One workgroup walks a row block's whole reduction length:
 * a fixed summation order, no atomics. */
int rcmarl_lattice_pack_dz_rowsum(const float* dz, void* dzp, float* sums, int sums_ld, int sums_off, int S, int N, int B, int hid,
                                  int ldb, int dzp_rt, int dzp_kt, void* stream);
/* = rcmarl_layer1_forward on (kp, wp); theta supplies b1 */
int rcmarl_layer1_forward_lattice(const void* kp, int kp_rt, int kp_kt, const void* wp, int wp_rt, int wp_kt,
                                  const float* theta, float* a1t, int S, int N, int B, int in_dim, int hid, int ldp,
                                  int ldb, void* stream);
/* = rcmarl_layer1_backward_sgd on (ktp, dzp): W1[k][col] -= lr * alpha[k] * sum_b K[b][k]*dz1[col][b].
 * wp_out (or NULL): additionally receives what rcmarl_w1_split would produce from the UPDATED theta (masked
 * agents: from their unchanged rows), so the next forward of the same local fit needs no split pass. */
int rcmarl_layer1_backward_sgd_lattice(const void* ktp, int ktp_rt, int ktp_kt, const void* dzp, int dzp_rt, int dzp_kt,
                                       const float* alpha, float* theta, const int* mask, int S, int N, int B,
                                       int in_dim, int hid, int ldp, float lr, void* wp_out, int wp_rt, int wp_kt,
                                       void* stream);
/* = rcmarl_mid_fit, but a1t is left intact and dz1 is written as dzp (its 16-bit pieces in the current operand form --
 * two f16 pieces of 2^8 dz1, or three exact bf16 pieces --, rows = (agent,unit) column, reduction = replay row, zero
 * beyond B) for rcmarl_layer1_backward_sgd_lattice.  With f16 backward operands the step itself runs on the f16 matrix
 * core (two-piece operands, four exact products per fp32 product; RCMARL_MIDFIT=5: the fp32 kernel of rcmarl_mid_fit);
 * an agent whose activations / gradients leave the f16 range is redone in fp32 arithmetic by a second launch.
 * ovf_flags: int32[S*N + 1] owned by the caller, ZERO before its first use and not touched by anybody else while a call is in
 * flight (one buffer per stream / call site): [0, S*N) the call generation in which an agent was last flagged, [S*N] the
 * generation counter (bumped on the device: a hipGraph replay draws a fresh one).  NULL: the fp32-arithmetic kernel alone. */
int rcmarl_mid_fit_lattice(const float* a1t, const float* theta, const float* y, float* partials, void* dzp, int dzp_rt,
                           int dzp_kt, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, int* ovf_flags, void* stream);

/* (ABI 3: the fused local-fit prototypes -- rcmarl_fit_fused, rcmarl_forward_mid and their six helpers, ABI 2 -- left the library:
 * exact, measured not faster than the three launches above; tools/prototypes/fused_fit.hip keeps the source.) */

/* Shuffle permutations of the adversaries' mini-batch fits (Keras fit(shuffle=True) inside
 * agents/adversarial_CAC_agents.py:38-41,131-135,163-165,237-253).  TensorFlow's shuffle RNG is not reproducible
 * outside TensorFlow; the shuffle is DEFINED here (csrc/shuffle.hip; same statement in the oracle):
 * perm[s][q][e][.] = row order that sorts Philox4x32-10(counter = (row, e, calls[q], 2), key = seeds[s]).word0.
 * seeds: uint64[S]; calls: int[n] (index of each fit call in the run); perm: int[S][n][epochs][B], B <= 8192. */
int rcmarl_shuffle_perms(const void* seeds, const int* calls, int n, int epochs, int B, int* perm, int S, void* stream);

/* reduce the partials over chunks and apply SGD to b1,W2,b2,W3,b3; loss_out[S][N] (or NULL) = MSE. */
int rcmarl_small_sgd(const float* partials, float* theta, const int* mask, float* loss_out, int S, int N,
                     int B, int in_dim, int hid, int ldp, float lr, void* stream);
/* K6: out[s][n][b] = head(a1t) (r_applied NULL), or the TD target r_applied + gamma*V
 * (local_TD_target, agents/resilient_CAC_agents.py:114-115).  Also serves r_team, V, nV of :95-97. */
int rcmarl_mid_value(const float* a1t, const float* theta, const float* r_applied, float gamma, float* out, int S,
                     int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream);
/* The same on the vector ALUs in fp32 whatever the operand form (rcmarl_mid_value runs layer 2 of 20-unit nets on the f16 matrix core
 * with two-piece operands: last bits differ).  For callers whose results feed a long sequential chain that amplifies last-bit
 * differences -- the adversaries' targets ahead of their 32-row mini-batch fits (agents/adversarial_CAC_agents.py:111-117,131-135). */
int rcmarl_mid_value_f32(const float* a1t, const float* theta, const float* r_applied, float gamma, float* out, int S,
                         int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream);

/* K2+K3: consensus over estimates + projection residual.
 * Replaces resilient_consensus_critic/_TR (agents/resilient_CAC_agents.py:168-206) and the
 * gradient of critic_update_team/TR_update_team (:60-84) for all cooperative agents:
 *   phi = features_{theta_i}(x); V_k = phi . W3(msg[nbr[i][k]]) + b3(msg[...]); agg = aggregate_H(V)
 *   e = (agg - V_{theta_i}) / (|phi|^2 + 1); partials[S][N][nchunk][hid+1] = [sum_b e*phi | sum_b e].
 * a1t must hold layer-1 activations of theta (the freshly aggregated hidden layers).
 * agg_out[S][N][ldb] optional (tests). */
int rcmarl_consensus_head(const float* a1t, const float* theta, const float* msg, const int* nbr,
                          const int* coop, float* partials, float* agg_out, int S, int N, int B, int in_dim,
                          int hid, int ldp, int ldb, int d, int H, void* stream);
/* K3 alone: the same projection residual toward a caller-supplied aggregate agg[S][N][ldb]
 * (critic_update_team(s, agg) / TR_update_team(sa, agg) called on their own, :60-84). */
int rcmarl_projection_residual(const float* a1t, const float* theta, const float* agg, const int* coop,
                               float* partials, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                               void* stream);
/* W3 += sum/B, b3 += sum/B: the normalised projection step (fast_lr cancels, :67-71). */
int rcmarl_head_apply(const float* partials, float* theta, const int* coop, int S, int N, int B, int in_dim,
                      int hid, int ldp, void* stream);

/* K7 actor: softmax + sample-weighted sparse CE forward/backward through layers 3-2
 * (actor.train_on_batch(s, a_local, sample_weight=TD), agents/resilient_CAC_agents.py:99).
 * act_t, delta: [S][N][ldb] labels (as floats) and sample weights; a1t overwritten by dz1. */
int rcmarl_mid_actor(float* a1t, const float* theta, const float* act_t, const float* delta, float* partials, int S,
                     int N, int B, int in_dim, int hid, int n_actions, int ldp, int ldb, void* stream);
int rcmarl_small_adam(const float* partials, float* theta, float* adam_m, float* adam_v, const int* mask,
                      float* loss_out, int S, int N, int B, int in_dim, int hid, int n_actions, int ldp, float alpha,
                      float one_m_b1, float one_m_b2, float eps, void* stream);

/* Several independent rcmarl_minibatch_fit jobs in ONE launch (no side streams: the three chains a Malicious agent needs per consensus
 * epoch -- private critic, compromised team-reward net, compromised critic, agents/adversarial_CAC_agents.py:131-135,146-152,163-165 --
 * start together and the epoch stays capturable in a hipGraph).  Same arithmetic and results as njobs calls of rcmarl_minibatch_fit.
 * Networks of at most 20 inputs, all jobs in the same input class (<= 16, or 17..20), at most 4 jobs; otherwise
 * RCMARL_ERR_UNSUPPORTED (call rcmarl_minibatch_fit per job).  `jobs` is HOST memory, read during the call only. */
typedef struct rcmarl_mb_job {
  const float* x; long x_seed_stride;      /* replay tensor of this network's input family and its seed stride (floats) */
  float* theta;                            /* [S][N][ldp], rows `agents` fitted in place */
  const int* agents; int n_adv;            /* the fitted agents of every seed */
  int in_dim, ldp, reserved_;
  const float* y;                          /* targets [S][N][ldb] */
  const int* perm;                         /* int32 [S][n_adv][epochs][B], or NULL (natural order) */
  float* loss_out;                         /* [S][N] first-epoch loss, or NULL */
  int* ovf_flags;                          /* int32[S * n_adv], zero before first use, one buffer per job and call site */
} rcmarl_mb_job;
int rcmarl_minibatch_fit_multi(const rcmarl_mb_job* jobs, int njobs, int S, int N, int B, int hid, int ldb, int batch_size,
                               int epochs, float lr, void* stream);

/* X1: one whole Keras fit() of the adversaries' networks per launch (one workgroup per (seed, adversary)).
 * agents: int[n_adv] agent indices; the rows theta[s][agents[k]] are trained IN PLACE.
 * perm: int[S][n_adv][epochs][B] row permutation per epoch (Keras shuffle; NULL = natural order).
 * rcmarl_minibatch_fit: SGD + MSE against y[S][N][ldb]  -- Greedy/Malicious critic & TR fits,
 *   fit(batch_size=32, epochs=10), agents/adversarial_CAC_agents.py:121-165, 228-253.  With <= 20 inputs every
 *   product of a step runs on the f16 matrix core (two-piece f16 operands, fp32 masters and accumulation); a network
 *   whose operands leave the f16 range is redone in fp32 arithmetic by a second launch (RCMARL_MB_MX=0: fp32 only).
 *   ovf_flags: int32[S * n_adv] owned by the caller, ZERO before its first use (the fix-up clears what it consumes, so the
 *   buffer is zero again when the call has run; safe under hipGraph replay); one buffer per call in flight.  NULL: the fp32
 *   kernel alone.
 * rcmarl_minibatch_actor: Adam + sample-weighted sparse CE -- the adversaries' actor_update,
 *   fit(batch_size=200, epochs=1), :38-41, :111-117, :221-225.  t0 = Adam steps taken before this call.
 * loss_out[S][N] (or NULL): first-epoch loss. */
int rcmarl_minibatch_fit(const float* x, long x_seed_stride, float* theta, const int* agents, int n_adv,
                         const float* y, const int* perm, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                         int batch_size, int epochs, float lr, float* loss_out, int* ovf_flags, void* stream);
int rcmarl_minibatch_actor(const float* x, long x_seed_stride, float* theta, float* adam_m, float* adam_v,
                           const int* agents, int n_adv, const float* act_t, const float* delta, const int* perm, int S,
                           int N, int B, int in_dim, int hid, int n_actions, int ldp, int ldb, int batch_size,
                           int epochs, double lr, double beta1, double beta2, double eps, int t0, float* loss_out,
                           void* stream);

/* r_coop[s][b] = sum_{coop n, index order} r[s][b][n]/n_coop   (training/train_agents.py:96-98) */
int rcmarl_team_reward(const float* r, long seed_stride, const int* coop, int n_coop, float* rcoop, int S,
                       int N, int B, int ldb, void* stream);
/* out[s][n][b] = src[s][b][n] (mode NULL or mode[n]==0), rcoop[s][b] (1), -rcoop[s][b] (2):
 * the r_applied selection of training/train_agents.py:106-116 and the a[:,node] slices of :151-153 */
int rcmarl_gather_agent_major(const float* src, long seed_stride, const float* rcoop, const int* mode, float* out,
                              int S, int N, int B, int ldb, void* stream);
/* The TD target's own forward pass (agents/resilient_CAC_agents.py:114-115) only concerns the LAST next-state row of every episode
 * (inside an episode ns[b] = s[b+1], training/train_agents.py:66-80): rcmarl_gather_rows collects every `step`-th row of a replay
 * tensor, dst[s][k][:] = src[s][first + k*step][:]; rcmarl_scatter_values puts their values back into an agent-major vector,
 * out[s][n][first + k*step] = v[s][n][k]  (r_applied == NULL)  or  r_applied[s][n][first + k*step] + gamma * v[s][n][k]. */
int rcmarl_gather_rows(const float* src, long seed_stride, int first, int step, int n_rows, int width, float* dst, int S,
                       void* stream);
int rcmarl_scatter_values(const float* v, const float* r_applied, float gamma, float* out, int first, int step, int n_rows,
                          int S, int N, int ldb, void* stream);

/* delta = r_team + gamma*nV - V   (global_TD_error, agents/resilient_CAC_agents.py:98) */
int rcmarl_td_error(const float* r_team, const float* v_next, const float* v_cur, float gamma, float* delta,
                    long n_total, void* stream);

/* Rollout.  xs[S][2N]: scaled global state vector; pos/goal: int[S][N][2]; scale: DEVICE double[4] =
 * {mean_x, mean_y, std_x, std_y} (environments/grid_world.py:29-33); seeds: DEVICE uint64[S].
 * probs[s][n][:] = actor_n(xs[s])                    (actor.predict, agents/resilient_CAC_agents.py:215) */
int rcmarl_policy_probs(const float* xs, const float* theta, float* probs, int S, int N, int in_dim, int hid,
                        int n_actions, int ldp, void* stream);
/* out[s][n] = critic_n(xs[s])                        (training/train_agents.py:60-62) */
int rcmarl_value_rows(const float* xs, const float* theta, float* out, int S, int N, int in_dim, int hid, int ldp,
                      void* stream);
/* fused environment step with the on-device Philox stream (rng_mode 'device'): actor forward,
 * get_action's three draws (agents/resilient_CAC_agents.py:208-219), Grid_World.step
 * (environments/grid_world.py:47-64), replay append and discounted-return accumulation
 * (training/train_agents.py:69-80).  Writes row `row` of the five replay tensors [S][cap][.]. */
int rcmarl_rollout_step(const float* xs, const int* pos, const int* goal, const float* theta,
                        const unsigned long long* seeds, int nrow, int ncol, const double* scale, float* rp_s,
                        float* rp_ns, float* rp_sa, float* rp_a, float* rp_r, long cap, long row, int* pos_next,
                        float* xs_next, double* ret, double gpow, int episode, int step, float mu, int S, int N,
                        int hid, int n_actions, int ldp, int* act_out, void* stream);
/* same transition for host-sampled actions int[S][N] (rng_mode 'numpy': NumPy legacy stream parity) */
int rcmarl_env_apply(const int* pos, const int* goal, const int* actions, int nrow, int ncol, const double* scale,
                     float* rp_s, float* rp_ns, float* rp_sa, float* rp_a, float* rp_r, long cap, long row,
                     int* pos_next, float* xs_next, double* ret, double gpow, int S, int N, void* stream);
/* Grid_World.reset (environments/grid_world.py:37-45): positions from pos_in (int[S][N][2]) or, if NULL,
 * from the Philox stream; also zeroes the per-episode return accumulator ret[S][N] (double). */
int rcmarl_env_reset(const int* pos_in, const unsigned long long* seeds, int nrow, int ncol, const double* scale,
                     int episode, int* pos, float* xs, double* ret, int S, int N, void* stream);

/* Episode-parallel rollout (rng_mode 'device'): networks are frozen between update blocks
 * (training/train_agents.py:86) and the Philox draws depend only on (seed, episode, step, agent), so the
 * E episodes of a block are stepped together -- max_ep_len launches per block, actor weights read once per
 * step for all episodes.  Episode-minor state (EP = E rounded up to 64, lanes = episodes):
 * xsT[S][2N][EP], posT[S][N][2][EP], retT[S][N][EP] (double).  Replay row of (episode e, step j) =
 * row0 + e*ep_len + j, i.e. the order the sequential loop (:46-80) would have produced.
 * est[E][S][N] = start-state critic values (:60-62). */
int rcmarl_rollout_step_episodes(const float* xsT, const int* posT, const int* goal, const float* theta,
                                 const unsigned long long* seeds, int nrow, int ncol, const double* scale, float* rp_s,
                                 float* rp_ns, float* rp_sa, float* rp_a, float* rp_r, long cap, long row0, int ep_len,
                                 int* posT_next, float* xsT_next, double* retT, double gpow, int episode0, int step,
                                 float mu, int S, int N, int E, int EP, int hid, int n_actions, int ldp, void* stream);
int rcmarl_value_rows_episodes(const float* xsT, const float* theta, float* est, int S, int N, int E, int EP, int hid,
                               int ldp, void* stream);
int rcmarl_env_reset_episodes(const int* pos_in, const unsigned long long* seeds, int nrow, int ncol,
                              const double* scale, int episode0, int* posT, float* xsT, double* retT, int S, int N,
                              int E, int EP, void* stream);

/* ---- wide networks (any hidden width: BASELINE configs[4], the 512-unit critic) -- csrc/wide_kernels.hip ----
 * The reference builds its networks in main.py:59-82 with 20 hidden units; a wider model object handed to the same
 * agent API makes every layer a true dense GEMM per agent (f32 MFMA), the 1-unit head a column/row pass over the
 * feature-major activations act[S][N*hid][ldb].  Same reference functions as the hid = 20 entry points above.
 *
 * out[s][n][j][b] = lrelu(sum_k W[k][j] in(k,b) + bias[j]); W = theta[s][n] + w_off (K x J, Keras kernel order),
 * bias = theta[s][n] + b_off.  in_row_major != 0: in[s][b][k] (replay rows; ld_in floats per row; agent stride 0 for
 * the shared global state), else feature-major in[s][n][k][b] with ld_in = its row length.   (model(x), agents/
 * resilient_CAC_agents.py:95-97,114) */
int rcmarl_dense_forward(const float* in, long in_seed_stride, long in_agent_stride, int in_row_major, int ld_in,
                         const float* theta, int w_off, int b_off, float* out, int S, int N, int B, int K, int J,
                         int ldp, int ldb, void* stream);
/* dz_in[s][n][k][b] = (sum_j W[k][j] dz_out[j][b]) * lrelu'(act_in[k][b])      (backward of fit(), :118,:136) */
int rcmarl_dense_backward_data(const float* dz_out, const float* theta, int w_off, const float* act_in, float* dz_in,
                               int S, int N, int B, int K, int J, int ldp, int ldb, void* stream);
/* W[k][j] -= lr * sum_b in(k,b) dz[j][b] for agents with mask[n] != 0 (mask NULL: all)   (plain SGD step of fit()) */
int rcmarl_dense_backward_sgd(const float* in, long in_seed_stride, long in_agent_stride, int in_row_major, int ld_in,
                              const float* dz, float* theta, int w_off, const int* mask, int S, int N, int B, int K,
                              int J, int ldp, int ldb, float lr, void* stream);
/* The dense layers below run on the 16-bit matrix core (default; RCMARL_WIDE_F16=0: the fp32-input MFMA kernel): both fp32
 * operands travel as two f16 pieces of the value times a fixed power of two (weights 2^10, dz 2^8, activations 2^6) -- to 2^-22
 * relative while the scaled value is at least 2^-3 (|w| >= 1.2e-4, |dz| >= 4.9e-4, |a| >= 2.0e-3), to an ABSOLUTE 2^-25 of the scaled
 * unit below that (2.9e-11 / 1.2e-10 / 4.7e-10: the low piece is a subnormal f16 there) -- and a product is three matrix passes
 * (l*h + h*l + h*h; the l*l term, 2^-22 of the product, is dropped), fp32 accumulate; a workgroup whose operands leave the f16 range
 * (|w| > 63, |dz| > 254, |a| > 1015) recomputes its tile in fp32 inside the same launch.  Read from the environment once;
 * rcmarl_wide_set_f16_mode(0 / 1, or -1 = read the environment again) switches it.  Inputs given as replay rows (in_row_major)
 * always take the fp32 kernel. */
int rcmarl_wide_f16_mode(void);
int rcmarl_wide_set_f16_mode(int mode);
int rcmarl_wide_grad_size(int hid);      /* floats per (seed, agent) of `grads`: [gW3 (hid) | gb3 | gb2 (hid) | gb1 (hid)] */
int rcmarl_wide_rows_per_chunk(void);    /* `losspart` holds ceil(B / this) floats per (seed, agent) */
/* out[s][n][b] = a2[:,b] . W3 + b3, or r_applied + gamma * that (TD target, :114-115) */
int rcmarl_wide_head_value(const float* a2, const float* theta, const float* r_applied, float gamma, float* out, int S,
                           int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream);
/* MSE head of fit() (:118): dz3 = 2 (V - y) / B; a2 is OVERWRITTEN by dz2 = W3 dz3 lrelu'(a2); grads gets gW3, gb3, gb2;
 * losspart the chunk sums of (V - y)^2 */
int rcmarl_wide_head_fit(float* a2, const float* theta, const float* y, float* dz3, float* grads, float* losspart, int S,
                         int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream);
/* grads.gb1[j] = sum_b dz1[j][b] */
int rcmarl_wide_bias_grad(const float* dz1, float* grads, int S, int N, int B, int hid, int ldb, void* stream);
/* b1, b2, W3, b3 -= lr * grads (masked agents); loss_out[s][n] = sum(losspart)/B if not NULL */
int rcmarl_wide_small_sgd(const float* grads, const float* losspart, float* theta, const int* mask, float* loss_out,
                          int S, int N, int B, int in_dim, int hid, int ldp, float lr, void* stream);
/* K2+K3 (resilient_consensus_critic + critic_update_team, :168-206 + :60-71) for a wide head.  phi = layer-2
 * activations of the live net.  Scratch: hmat [S][N][d+1][hid], hb [S][N][d+1], est [S][N][d+1][ldb],
 * ebuf [S][N][ldb].  grads[s][n][0..hid] = [sum_b e phi | sum_b e] for cooperative agents, e = (agg - V_live) /
 * (|phi|^2 + 1); agg_out optional.  agg_in != NULL: projection toward that aggregate only (msg/nbr/hmat/hb/est unused) */
int rcmarl_wide_consensus_head(const float* phi, const float* theta, const float* msg, const int* nbr, const int* coop,
                               const float* agg_in, float* hmat, float* hb, float* est, float* ebuf, float* grads,
                               float* agg_out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, int d, int H,
                               void* stream);
/* rcmarl_wide_consensus_head (estimate consensus, no agg_in) with |phi|^2 per replay row given as n_parts parts
 * nparts[s][n][n_parts][ldb] (rcmarl_pk_forward2's npart): the selection pass reads d + 1 estimates per row instead of d + 1 + hid values */
int rcmarl_wide_consensus_head_nrm(const float* phi, const float* nparts, int n_parts, const float* theta, const float* msg,
                                   const int* nbr, const int* coop, float* hmat, float* hb, float* est, float* ebuf, float* grads,
                                   float* agg_out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, int d, int H, void* stream);
/* W3 += grads[0..hid)/B, b3 += grads[hid]/B (cooperative agents) */
int rcmarl_wide_head_apply(const float* grads, float* theta, const int* coop, int S, int N, int B, int in_dim, int hid,
                           int ldp, void* stream);


/* ---- wide networks on PRE-SPLIT packed operands (ABI 4, round 6) -- csrc/dense_pk.hip ------------------------------------------
 * The same reference functions as the block above (one full-batch SGD step of fit(), agents/resilient_CAC_agents.py:103-122, and
 * model(x), :95-97,114) for a hidden width that is a multiple of 128 in the two-piece f16 operand form: every GEMM operand reaches
 * its kernel as packed f16 pieces written by the epilogue of the kernel that produced it ("PK" blocks of 8 KiB:
 * [rows/128][k/32][pieces][128 rows][32 k], 16-byte chunks XOR-swizzled; csrc/rcmarl_lattice.h), and the GEMMs run the LDS-DMA main
 * loop of the layer-1 lattice kernels.  Neither a2 nor dz2 is ever stored: LeakyReLU' takes two values, so the layer-2 masks travel as
 * ONE 16-bit piece and dz2 = W3 dz3 lrelu'(z2) is re-formed inside the consumers (csrc/dense_pk.hip has the algebra).
 * Z = S*N below; Bp = B rounded up to 256; JT = hid/128; JK = hid/32.
 *   a1_bk [Z][bk_rt >= Bp/128][JK][2][8 KiB]   rows = replay row, reduction = unit: two f16 pieces of 2^6 a1
 *   a1_kb [Z][JT][kb_kt >= Bp/32][2][8 KiB]    rows = unit, reduction = replay row
 *   s1    [Z*hid][s1_ld >= Bp/32] uint32       sign bits of a1 (bit b & 31 of word b >> 5)
 *   w2t, w2w3 [Z][JT][JK][2][8 KiB]            2^10 W2 (rows = j, reduction = k) / 2^10 W2[k][j] W3[j] (rows = k, reduction = j)
 *   mask_bj [Z][mbj_rt >= Bp/128][JK][8 KiB]   [z2 > 0] as f16 1.0 / 0;  mask_jb [Z][JT][mjb_kt >= Bp/32][8 KiB] as 0xffff / 0
 *   vpart [Z][JT][ldb], dzv [Z][4][Bp] uint16, gw3part [Z][JT][hid], q [Z][hid], rs [Z][hid], gb1part [Z][ceil(B/128)][hid]
 * Every buffer must be readable over its whole extent (allocate zero-filled). */
int rcmarl_pk_supported(int hid);        /* 1: hid % 128 == 0 and the lattice path is in its two-piece f16 form (RCMARL_LAT_F16 = 3) */
/* the layer-1 lattice forward GEMM (rcmarl_layer1_forward_lattice) whose epilogue writes a1_bk / a1_kb / s1 (each optional)
 * INSTEAD of the fp32 activations */
int rcmarl_layer1_forward_lattice_pk(const void* kp, int kp_rt, int kp_kt, const void* wp, int wp_rt, int wp_kt, const float* theta,
                                     void* a1_bk, int bk_rt, void* a1_kb, int kb_kt, unsigned* s1, int s1_ld, int* ovf_flag, int S, int N,
                                     int B, int in_dim, int hid, int ldp, void* stream);
/* RANGE.  The pieces saturate instead of overflowing (|a1| > 1015, |W2| or |W2 W3| > 63, |dz1| > 254 are carried clipped, finite).  The
 * three producers of large operands take an optional `ovf_flag` (one int in device memory, never cleared by the library): set to 1 when
 * a value left the range -- a caller polls it between blocks and switches to the rcmarl_dense_* entry points (which recompute such
 * tiles in fp32) if it matters to it.
 * W2, W3 of theta[s][n] -> w2t, w2w3, rs[k] = sum_j W2[k][j] W3[j] */
int rcmarl_pk_pack_w2(const float* theta, void* w2t, void* w2w3, float* rs, int* ovf_flag, int S, int N, int in_dim, int hid, int ldp,
                      void* stream);
/* layer 2 forward; outputs, each optional: a2 (fp32 feature-major [S][N*hid][ldb]: phi of rcmarl_wide_consensus_head), mask_bj,
 * mask_jb, vpart (per tile of units: sum_j a2[j][b] W3[j]), npart (same shape: sum_j a2[j][b]^2, |phi|^2 of the projection step) */
int rcmarl_pk_forward2(const void* w2t, const void* a1_bk, int bk_rt, const float* theta, float* a2, void* mask_bj, int mbj_rt,
                       void* mask_jb, int mjb_kt, float* vpart, float* npart, int S, int N, int B, int in_dim, int hid, int ldp, int ldb,
                       void* stream);
int rcmarl_pk_parts(int hid);            /* tiles of units per agent = parts per row in vpart / npart / gw3part (hid / 256, or hid / 128) */
/* V = sum_t vpart[t] + b3.  mode 0: out = V; 1: out = aux + gamma V (TD target, :114-115); 2: MSE head of fit() with aux = y:
 * out = dz3 = 2 (V - y) / B, dzv = f16 pieces of 2^8 dz3 and 2^8 (0.1 dz3), losspart [Z][ceil(B/256)] */
int rcmarl_pk_head(const float* vpart, const float* theta, const float* aux, float gamma, int mode, float* out, void* dzv,
                   float* losspart, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, void* stream);
/* dz1 = lrelu'(a1) W2 dz2 straight into the lattice backward GEMM's operand dzp ([S][dzp_rt][dzp_kt][2][8 KiB], rows = n*hid + unit)
 * + gb1part (sums of dz1 over each tile of 128 replay rows) */
int rcmarl_pk_backward_data(const void* mask_bj, int mbj_rt, const void* w2w3, const float* rs, const unsigned* s1, int s1_ld,
                            const float* dz3, void* dzp, int dzp_rt, int dzp_kt, float* gb1part, int* ovf_flag, int S, int N, int B,
                            int hid, int ldb, void* stream);
/* W2 -= lr a1^T dz2 (agents with mask[n] != 0; NULL: all); gw3part and q carry the head's and b2's gradients to rcmarl_pk_small_sgd */
int rcmarl_pk_backward_w2(const void* a1_kb, int kb_kt, const void* mask_jb, int mjb_kt, const void* dzv, float* theta, const int* mask,
                          float* gw3part, float* q, int S, int N, int B, int in_dim, int hid, int ldp, float lr, void* stream);
/* b1, b2, W3, b3 -= lr grad (masked agents); loss_out [S][N] = sum(losspart) / B if not NULL */
int rcmarl_pk_small_sgd(const float* gw3part, const float* q, const float* gb1part, const float* dz3, const float* losspart, float* theta,
                        const int* mask, float* loss_out, int S, int N, int B, int in_dim, int hid, int ldp, int ldb, float lr,
                        void* stream);

/* C2 (one instance over several GPUs): the pack / unpack pass of the two all-to-all transposes of the message
 * matrix, which replace the reference's in-process gather `[critic_weights[i] for i in in_nodes[node]]`
 * (training/train_agents.py:129-130).  dst[b][r][c] = src[b][r][c] for b < batches, r < rows, c < cols; strides in
 * floats; rows with row_mask[r] == 0 are left untouched (row_mask may be NULL).  One pass, 16-byte accesses when the
 * strides allow. */
int rcmarl_copy3d(const float* src, long src_batch, long ld_src, float* dst, long dst_batch, long ld_dst, int batches,
                  int rows, int cols, const int* row_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCMARL_H */
