"""ctypes binding of the C-ABI declared in include/rcmarl.h.

The product loads ``lib/librcmarl_hip.so`` (built by rcmarl_amd.build for
gfx950) and FAILS LOUDLY if it is missing -- there is no CPU fallback.  The
same signature table is used by the tests to bind the hipemu build of the very
same kernel sources (tests/hipemu), which is test infrastructure only.
"""
import ctypes as C
import os

c_f32p = C.c_void_p      # float*   (device pointer, or host pointer for the hipemu build)
c_i32p = C.c_void_p      # int*
c_u8p = C.c_void_p       # unsigned char*


class MbJob(C.Structure):
    """rcmarl_mb_job (include/rcmarl.h): one network family of rcmarl_minibatch_fit_multi"""
    _fields_ = [("x", C.c_void_p), ("x_seed_stride", C.c_long), ("theta", C.c_void_p), ("agents", C.c_void_p), ("n_adv", C.c_int),
                ("in_dim", C.c_int), ("ldp", C.c_int), ("reserved_", C.c_int), ("y", C.c_void_p), ("perm", C.c_void_p),
                ("loss_out", C.c_void_p), ("ovf_flags", C.c_void_p)]
c_f64p = C.c_void_p      # double*
c_stream = C.c_void_p    # hipStream_t

# name -> argtypes (every function returns int: 0 ok, see RCMARL_ERR_* in include/rcmarl.h)
c_long = C.c_long
c_int = C.c_int
c_float = C.c_float

# name -> argtypes (every function returns int: 0 ok, see RCMARL_ERR_* in include/rcmarl.h)
SIGNATURES = {
    "rcmarl_abi_version": [],
    "rcmarl_mb_job_layout": [c_int],
    # jobs (host array of MbJob), njobs, S, N, B, hid, ldb, batch_size, epochs, lr, stream
    "rcmarl_minibatch_fit_multi": [C.c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_stream],
    "rcmarl_lattice_forget": [c_u8p],
    "rcmarl_fit_partial_size": [c_int],
    "rcmarl_actor_partial_size": [c_int, c_int],
    "rcmarl_rows_per_chunk": [],
    "rcmarl_lattice_f16_mode": [],
    "rcmarl_lattice_set_f16_mode": [c_int],
    # msg, theta, nbr, coop, S, N, ldp, P_hid, d, H, lo_dbg, hi_dbg, stream
    "rcmarl_consensus_params": [c_f32p, c_f32p, c_i32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_f32p, c_f32p, c_stream],
    "rcmarl_consensus_params_circulant_supported": [c_int, c_int, c_int],
    # msg, theta, coop, S, N, ldp, P_hid, d, H, lo_dbg, hi_dbg, stream
    "rcmarl_consensus_params_circulant": [c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p,
                                          c_stream],
    # x, x_seed_stride, theta, a1t, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_layer1_forward": [c_f32p, c_long, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_stream],
    # x, x_seed_stride, dz1t, theta, mask, S, N, B, in_dim, hid, ldp, ldb, lr, stream
    "rcmarl_layer1_backward_sgd": [c_f32p, c_long, c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_float, c_stream],
    # x, x_seed_stride, dz1t, theta, m, v, mask, S, N, B, in_dim, hid, ldp, ldb, alpha, 1-b1, 1-b2, eps, stream
    "rcmarl_layer1_backward_adam": [c_f32p, c_long, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_stream],
    # a1t, theta, y, partials, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_mid_fit": [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # lattice (exact bf16x3) layer-1 path, csrc/lattice_gemm.hip ------------------------------------------
    # x, x_seed_stride, alpha, S, B, in_dim, kp, kp_rt, kp_kt, ktp, ktp_rt, ktp_kt, flag, stream
    "rcmarl_lattice_encode": [c_f32p, c_long, c_f32p, c_int, c_int, c_int, c_u8p, c_int, c_int, c_u8p, c_int, c_int,
                              c_i32p, c_stream],
    # theta, alpha, wp, S, N, in_dim, hid, ldp, wp_rt, wp_kt, stream
    "rcmarl_w1_split": [c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # dz, dzp, S, N, B, hid, ldb, dzp_rt, dzp_kt, stream
    "rcmarl_lattice_pack_dz": [c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # dz, dzp, sums, sums_ld, sums_off, S, N, B, hid, ldb, dzp_rt, dzp_kt, stream
    "rcmarl_lattice_pack_dz_rowsum": [c_f32p, c_u8p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta, a1t, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_layer1_forward_lattice": [c_u8p, c_int, c_int, c_u8p, c_int, c_int, c_f32p, c_f32p, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_stream],
    # ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, hid, ldp, lr, wp_out, wp_rt, wp_kt,
    # stream
    "rcmarl_layer1_backward_sgd_lattice": [c_u8p, c_int, c_int, c_u8p, c_int, c_int, c_f32p, c_f32p, c_u8p, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_float, c_u8p, c_int, c_int, c_stream],
    # a1t, theta, y, partials, dzp, dzp_rt, dzp_kt, S, N, B, in_dim, hid, ldp, ldb, ovf_flags, stream
    "rcmarl_mid_fit_lattice": [c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_i32p, c_stream],
    # seeds(u64[S]), calls(int[n]), n, epochs, B, perm(int[S][n][epochs][B]), S, stream
    "rcmarl_shuffle_perms": [C.c_void_p, c_i32p, c_int, c_int, c_int, c_i32p, c_int, c_stream],
    # partials, theta, mask, loss_out, S, N, B, in_dim, hid, ldp, lr, stream
    "rcmarl_small_sgd": [c_f32p, c_f32p, c_u8p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_stream],
    # a1t, theta, r_applied, gamma, out, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_mid_value": [c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_stream],
    "rcmarl_mid_value_f32": [c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_stream],
    # a1t, theta, msg, nbr, coop, partials, agg_out, S, N, B, in_dim, hid, ldp, ldb, d, H, stream
    "rcmarl_consensus_head": [c_f32p, c_f32p, c_f32p, c_i32p, c_u8p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_int, c_int, c_stream],
    # partials, theta, coop, S, N, B, in_dim, hid, ldp, stream
    "rcmarl_head_apply": [c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # a1t, theta, act_t, delta, partials, S, N, B, in_dim, hid, n_actions, ldp, ldb, stream
    "rcmarl_mid_actor": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_stream],
    # partials, theta, m, v, mask, loss_out, S, N, B, in_dim, hid, n_actions, ldp, alpha, 1-b1, 1-b2, eps, stream
    "rcmarl_small_adam": [c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_float, c_float, c_float, c_float, c_stream],
    # r, seed_stride, coop, n_coop, rcoop, S, N, B, ldb, stream
    "rcmarl_team_reward": [c_f32p, c_long, c_u8p, c_int, c_f32p, c_int, c_int, c_int, c_int, c_stream],
    # a1t, theta, agg, coop, partials, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_projection_residual": [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_stream],
    # x, x_seed_stride, theta, agents, n_adv, y, perm, S, N, B, in_dim, hid, ldp, ldb, batch_size, epochs, lr,
    # loss_out, ovf_flags, stream
    "rcmarl_minibatch_fit": [c_f32p, c_long, c_f32p, c_i32p, c_int, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_float, c_f32p, c_i32p, c_stream],
    # x, x_seed_stride, theta, adam_m, adam_v, agents, n_adv, act_t, delta, perm, S, N, B, in_dim, hid, n_actions,
    # ldp, ldb, batch_size, epochs, lr, beta1, beta2, eps, t0, loss_out, stream
    "rcmarl_minibatch_actor": [c_f32p, c_long, c_f32p, c_f32p, c_f32p, c_i32p, c_int, c_f32p, c_f32p, c_i32p, c_int,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, C.c_double, C.c_double,
                               C.c_double, C.c_double, c_int, c_f32p, c_stream],
    # src, seed_stride, rcoop, mode, out, S, N, B, ldb, stream
    "rcmarl_gather_agent_major": [c_f32p, c_long, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_stream],
    # src, seed_stride, first, step, n_rows, width, dst, S, stream
    "rcmarl_gather_rows": [c_f32p, c_long, c_int, c_int, c_int, c_int, c_f32p, c_int, c_stream],
    # v, r_applied, gamma, out, first, step, n_rows, S, N, ldb, stream
    "rcmarl_scatter_values": [c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # r_team, v_next, v_cur, gamma, delta, n_total, stream
    "rcmarl_td_error": [c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_long, c_stream],
    # xs, theta, probs, S, N, in_dim, hid, n_actions, ldp, stream
    "rcmarl_policy_probs": [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # xs, theta, out, S, N, in_dim, hid, ldp, stream
    "rcmarl_value_rows": [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_stream],
    # xs, pos, goal, theta, seeds, nrow, ncol, scale, rp_s, rp_ns, rp_sa, rp_a, rp_r, cap, row, pos_next, xs_next,
    # ret, gpow, episode, step, mu, S, N, hid, n_actions, ldp, act_out, stream
    "rcmarl_rollout_step": [c_f32p, c_i32p, c_i32p, c_f32p, C.c_void_p, c_int, c_int, c_f64p, c_f32p, c_f32p, c_f32p,
                            c_f32p, c_f32p, c_long, c_long, c_i32p, c_f32p, c_f64p, C.c_double, c_int, c_int, c_float,
                            c_int, c_int, c_int, c_int, c_int, c_i32p, c_stream],
    # pos, goal, actions, nrow, ncol, scale, rp_s, rp_ns, rp_sa, rp_a, rp_r, cap, row, pos_next, xs_next, ret, gpow,
    # S, N, stream
    "rcmarl_env_apply": [c_i32p, c_i32p, c_i32p, c_int, c_int, c_f64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_long,
                         c_long, c_i32p, c_f32p, c_f64p, C.c_double, c_int, c_int, c_stream],
    # pos_in, seeds, nrow, ncol, scale, episode, pos, xs, ret, S, N, stream
    "rcmarl_env_reset": [c_i32p, C.c_void_p, c_int, c_int, c_f64p, c_int, c_i32p, c_f32p, c_f64p, c_int, c_int,
                         c_stream],
    # xsT, posT, goal, theta, seeds, nrow, ncol, scale, rp_s, rp_ns, rp_sa, rp_a, rp_r, cap, row0, ep_len, posT_next,
    # xsT_next, retT, gpow, episode0, step, mu, S, N, E, EP, hid, n_actions, ldp, stream
    "rcmarl_rollout_step_episodes": [c_f32p, c_i32p, c_i32p, c_f32p, C.c_void_p, c_int, c_int, c_f64p, c_f32p, c_f32p,
                                     c_f32p, c_f32p, c_f32p, c_long, c_long, c_int, c_i32p, c_f32p, c_f64p, C.c_double,
                                     c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # xsT, theta, est, S, N, E, EP, hid, ldp, stream
    "rcmarl_value_rows_episodes": [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # pos_in, seeds, nrow, ncol, scale, episode0, posT, xsT, retT, S, N, E, EP, stream
    "rcmarl_env_reset_episodes": [c_i32p, C.c_void_p, c_int, c_int, c_f64p, c_int, c_i32p, c_f32p, c_f64p, c_int, c_int,
                                  c_int, c_int, c_stream],
    # ---- wide networks (csrc/wide_kernels.hip) -----------------------------------------------------------
    # in, in_seed_stride, in_agent_stride, in_row_major, ld_in, theta, w_off, b_off, out, S, N, B, K, J, ldp, ldb, stream
    "rcmarl_dense_forward": [c_f32p, c_long, c_long, c_int, c_int, c_f32p, c_int, c_int, c_f32p, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_stream],
    # dz_out, theta, w_off, act_in, dz_in, S, N, B, K, J, ldp, ldb, stream
    "rcmarl_dense_backward_data": [c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_stream],
    # in, in_seed_stride, in_agent_stride, in_row_major, ld_in, dz, theta, w_off, mask, S, N, B, K, J, ldp, ldb, lr, stream
    "rcmarl_dense_backward_sgd": [c_f32p, c_long, c_long, c_int, c_int, c_f32p, c_f32p, c_int, c_i32p, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_float, c_stream],
    "rcmarl_wide_grad_size": [c_int],
    "rcmarl_wide_f16_mode": [],
    "rcmarl_wide_set_f16_mode": [c_int],
    "rcmarl_wide_rows_per_chunk": [],
    # a2, theta, r_applied, gamma, out, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_wide_head_value": [c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_stream],
    # a2, theta, y, dz3, grads, losspart, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_wide_head_fit": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_stream],
    # dz1, grads, S, N, B, hid, ldb, stream
    "rcmarl_wide_bias_grad": [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_stream],
    # grads, losspart, theta, mask, loss_out, S, N, B, in_dim, hid, ldp, lr, stream
    "rcmarl_wide_small_sgd": [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                              c_stream],
    # phi, theta, msg, nbr, coop, agg_in, hmat, hb, est, ebuf, grads, agg_out, S, N, B, in_dim, hid, ldp, ldb, d, H, stream
    "rcmarl_wide_consensus_head": [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                   c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # grads, theta, coop, S, N, B, in_dim, hid, ldp, stream
    "rcmarl_wide_head_apply": [c_f32p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # ---- wide networks on pre-split packed operands (csrc/dense_pk.hip) -----------------------------------
    "rcmarl_pk_supported": [c_int],
    # kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta, a1_bk, bk_rt, a1_kb, kb_kt, s1, s1_ld, ovf_flag, S, N, B, in_dim, hid, ldp, stream
    "rcmarl_layer1_forward_lattice_pk": [c_u8p, c_int, c_int, c_u8p, c_int, c_int, c_f32p, c_u8p, c_int, c_u8p, c_int, c_i32p, c_int,
                                         c_i32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # theta, w2t, w2w3, rs, ovf_flag, S, N, in_dim, hid, ldp, stream
    "rcmarl_pk_pack_w2": [c_f32p, c_u8p, c_u8p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_int, c_stream],
    # w2t, a1_bk, bk_rt, theta, a2, mask_bj, mbj_rt, mask_jb, mjb_kt, vpart, npart, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_pk_forward2": [c_u8p, c_u8p, c_int, c_f32p, c_f32p, c_u8p, c_int, c_u8p, c_int, c_f32p, c_f32p, c_int, c_int, c_int, c_int,
                           c_int, c_int, c_int, c_stream],
    "rcmarl_pk_parts": [c_int],
    # phi, nparts, n_parts, theta, msg, nbr, coop, hmat, hb, est, ebuf, grads, agg_out, S, N, B, in_dim, hid, ldp, ldb, d, H, stream
    "rcmarl_wide_consensus_head_nrm": [c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                       c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_stream],
    # vpart, theta, aux, gamma, mode, out, dzv, losspart, S, N, B, in_dim, hid, ldp, ldb, stream
    "rcmarl_pk_head": [c_f32p, c_f32p, c_f32p, c_float, c_int, c_f32p, c_u8p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                       c_stream],
    # mask_bj, mbj_rt, w2w3, rs, s1, s1_ld, dz3, dzp, dzp_rt, dzp_kt, gb1part, ovf_flag, S, N, B, hid, ldb, stream
    "rcmarl_pk_backward_data": [c_u8p, c_int, c_u8p, c_f32p, c_i32p, c_int, c_f32p, c_u8p, c_int, c_int, c_f32p, c_i32p, c_int, c_int,
                                c_int, c_int, c_int, c_stream],
    # a1_kb, kb_kt, mask_jb, mjb_kt, dzv, theta, mask, gw3part, q, S, N, B, in_dim, hid, ldp, lr, stream
    "rcmarl_pk_backward_w2": [c_u8p, c_int, c_u8p, c_int, c_u8p, c_f32p, c_i32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_float, c_stream],
    # gw3part, q, gb1part, dz3, losspart, theta, mask, loss_out, S, N, B, in_dim, hid, ldp, ldb, lr, stream
    "rcmarl_pk_small_sgd": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_float, c_stream],
    # ---- sharded instance (csrc/shard_pack.hip) ----------------------------------------------------------
    # src, src_batch, ld_src, dst, dst_batch, ld_dst, batches, rows, cols, row_mask, stream
    "rcmarl_copy3d": [c_f32p, c_long, c_long, c_f32p, c_long, c_long, c_int, c_int, c_int, c_i32p, c_stream],
}
UNCHECKED = {"rcmarl_abi_version", "rcmarl_mb_job_layout", "rcmarl_lattice_forget", "rcmarl_fit_partial_size", "rcmarl_lattice_set_f16_mode",  "rcmarl_actor_partial_size", "rcmarl_rows_per_chunk", "rcmarl_lattice_f16_mode",
             "rcmarl_wide_grad_size", "rcmarl_wide_rows_per_chunk", "rcmarl_wide_f16_mode", "rcmarl_wide_set_f16_mode",
             "rcmarl_consensus_params_circulant_supported", "rcmarl_pk_supported", "rcmarl_pk_parts"}

ERRORS = {1: "RCMARL_ERR_ARG (bad argument)", 2: "RCMARL_ERR_LAUNCH (HIP launch failed)",
          3: "RCMARL_ERR_UNSUPPORTED (shape outside the compiled kernels)"}


class RcmarlError(RuntimeError):
    pass


def _preload_hip_runtime():
    """librcmarl_hip.so is linked WITHOUT its own HIP runtime (-no-hip-rt): it must run on the very
    runtime instance PyTorch uses, otherwise streams/pointers handed over from torch belong to a
    different libamdhip64 and every launch fails.  PyTorch wheels bundle their runtime and load it
    RTLD_LOCAL; re-open it RTLD_GLOBAL so our undefined hip* symbols bind to it."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    try:
        C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so.7", mode=C.RTLD_GLOBAL)
    except OSError as e:                       # pragma: no cover
        raise RcmarlError("cannot load the HIP runtime (libamdhip64): %s" % e)


class CLib:
    """Thin checked wrapper: ``lib.rcmarl_xxx(...)`` raises on a non-zero status."""

    def __init__(self, path, needs_hip=True):
        if not os.path.exists(path):
            raise RcmarlError(
                "rcmarl HIP library not found at %s -- build it with `python -m rcmarl_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        self.path = path
        if needs_hip:
            _preload_hip_runtime()
        self._dll = C.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self._dll, name)      # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = C.c_int
            setattr(self, name, fn if name in UNCHECKED else self._checked(name, fn))

    @staticmethod
    def _checked(name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise RcmarlError("%s failed: %s" % (name, ERRORS.get(rc, rc)))
            return 0
        call.__name__ = name
        return call


_lib = None


def load():
    """The product library (HIP, gfx950).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        from . import build
        _lib = CLib(build.lib_path())
    return _lib
