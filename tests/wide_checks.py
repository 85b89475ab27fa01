"""Wide-network (hid != 20) kernel checks shared by the hipemu and GPU tests: the dense-GEMM path of
csrc/wide_kernels.hip against the same oracle functions as the 20-unit kernels (kernel_checks.py)."""
import numpy as np

from oracle import mlp_np as M
from oracle import rpbcac_oracle as O
from kernel_checks import pad64, pack_rows, unpack_row, circulant, random_regular, rel_close


def wgeom(in_dim, hid):
    o_b1 = in_dim * hid
    o_W2 = o_b1 + hid
    o_b2 = o_W2 + hid * hid
    o_W3 = o_b2 + hid
    o_b3 = o_W3 + hid
    return dict(o_b1=o_b1, o_W2=o_W2, o_b2=o_b2, o_W3=o_W3, o_b3=o_b3, P=o_b3 + 1, P_hid=o_W3)


def wide_params(rng, S, N, in_dim, hid, bias_scale=0.1):
    out = []
    for s in range(S):
        row = []
        for n in range(N):
            p = M.init_mlp(rng, in_dim, hid, 1)
            for k in (1, 3, 5):
                p[k] += (bias_scale * rng.normal(size=p[k].shape)).astype(np.float32)
            row.append(p)
        out.append(row)
    return out


class WideBuffers:
    """Device scratch of the wide path for S seeds x N agents, B rows."""

    def __init__(self, bk, S, N, B, hid, d=1):
        ldb = pad64(B)
        z = lambda *sh: bk.dev(np.zeros(sh, np.float32))
        self.a1, self.a2, self.dz1 = z(S, N * hid, ldb), z(S, N * hid, ldb), z(S, N * hid, ldb)
        self.dz3 = z(S, N, ldb)
        self.grads = z(S, N, bk.lib.rcmarl_wide_grad_size(hid))
        self.losspart = z(S, N, (B + bk.lib.rcmarl_wide_rows_per_chunk() - 1) // bk.lib.rcmarl_wide_rows_per_chunk())
        self.hmat, self.hb, self.est, self.ebuf = z(S, N, d + 1, hid), z(S, N, d + 1), z(S, N, d + 1, ldb), z(S, N, ldb)


def forward2(bk, wb, d_x, x_stride, d_theta, S, N, B, in_dim, hid, ldp, ldb):
    """layers 1 and 2 of every agent: replay rows -> wb.a1 -> wb.a2"""
    g, L = wgeom(in_dim, hid), bk.lib
    L.rcmarl_dense_forward(bk.ptr(d_x), x_stride, 0, 1, in_dim, bk.ptr(d_theta), 0, g["o_b1"], bk.ptr(wb.a1), S, N, B, in_dim,
                           hid, ldp, ldb, bk.stream)
    L.rcmarl_dense_forward(bk.ptr(wb.a1), N * hid * ldb, hid * ldb, 0, ldb, bk.ptr(d_theta), g["o_W2"], g["o_b2"],
                           bk.ptr(wb.a2), S, N, B, hid, hid, ldp, ldb, bk.stream)


def fit_step(bk, wb, d_x, x_stride, d_msg, d_y, d_mask, d_loss, S, N, B, in_dim, hid, ldp, ldb, lr):
    """one full-batch SGD step of fit() on all three layers (the sequence engine._local_fit_wide runs)"""
    g, L = wgeom(in_dim, hid), bk.lib
    forward2(bk, wb, d_x, x_stride, d_msg, S, N, B, in_dim, hid, ldp, ldb)
    L.rcmarl_wide_head_fit(bk.ptr(wb.a2), bk.ptr(d_msg), bk.ptr(d_y), bk.ptr(wb.dz3), bk.ptr(wb.grads), bk.ptr(wb.losspart),
                           S, N, B, in_dim, hid, ldp, ldb, bk.stream)
    L.rcmarl_dense_backward_data(bk.ptr(wb.a2), bk.ptr(d_msg), g["o_W2"], bk.ptr(wb.a1), bk.ptr(wb.dz1), S, N, B, hid, hid,
                                 ldp, ldb, bk.stream)
    L.rcmarl_wide_bias_grad(bk.ptr(wb.dz1), bk.ptr(wb.grads), S, N, B, hid, ldb, bk.stream)
    L.rcmarl_dense_backward_sgd(bk.ptr(wb.a1), N * hid * ldb, hid * ldb, 0, ldb, bk.ptr(wb.a2), bk.ptr(d_msg), g["o_W2"],
                                bk.ptr(d_mask), S, N, B, hid, hid, ldp, ldb, lr, bk.stream)
    L.rcmarl_dense_backward_sgd(bk.ptr(d_x), x_stride, 0, 1, in_dim, bk.ptr(wb.dz1), bk.ptr(d_msg), 0, bk.ptr(d_mask), S, N,
                                B, in_dim, hid, ldp, ldb, lr, bk.stream)
    L.rcmarl_wide_small_sgd(bk.ptr(wb.grads), bk.ptr(wb.losspart), bk.ptr(d_msg), bk.ptr(d_mask), bk.ptr(d_loss), S, N, B,
                            in_dim, hid, ldp, lr, bk.stream)


def check_wide_forward(bk, S, N, B, in_dim, hid):
    rng = np.random.default_rng(S * 1000 + N * 100 + B + in_dim + hid)
    g = wgeom(in_dim, hid)
    ldp, ldb = pad64(g["P"]), pad64(B)
    params = wide_params(rng, S, N, in_dim, hid)
    theta = pack_rows(params, ldp)
    x = rng.normal(size=(S, B, in_dim)).astype(np.float32)
    r = rng.normal(size=(S, N, ldb)).astype(np.float32)
    d_x, d_th, d_r = bk.dev(x), bk.dev(theta), bk.dev(r)
    wb = WideBuffers(bk, S, N, B, hid)
    d_v = bk.dev(np.zeros((S, N, ldb), np.float32))
    d_y = bk.dev(np.zeros((S, N, ldb), np.float32))
    forward2(bk, wb, d_x, B * in_dim, d_th, S, N, B, in_dim, hid, ldp, ldb)
    bk.lib.rcmarl_wide_head_value(bk.ptr(wb.a2), bk.ptr(d_th), None, 0.0, bk.ptr(d_v), S, N, B, in_dim, hid, ldp, ldb, bk.stream)
    bk.lib.rcmarl_wide_head_value(bk.ptr(wb.a2), bk.ptr(d_th), bk.ptr(d_r), 0.9, bk.ptr(d_y), S, N, B, in_dim, hid, ldp, ldb,
                                  bk.stream)
    a1, a2, v, y = bk.host(wb.a1), bk.host(wb.a2), bk.host(d_v), bk.host(d_y)
    for s in range(S):
        for n in range(N):
            p = params[s][n]
            z1 = x[s] @ p[0] + p[1]
            w1 = np.where(z1 > 0, z1, np.float32(0.1) * z1)
            z2 = w1 @ p[2] + p[3]
            w2 = np.where(z2 > 0, z2, np.float32(0.1) * z2)
            rel_close(a1[s, n * hid:(n + 1) * hid, :B].T, w1, 2e-6, "layer-1 activations")
            rel_close(a2[s, n * hid:(n + 1) * hid, :B].T, w2, 4e-6, "layer-2 activations")
            want = M.forward(p, x[s])[:, 0]
            rel_close(v[s, n, :B], want, 5e-6, "value")
            rel_close(y[s, n, :B], r[s, n, :B] + np.float32(0.9) * want, 5e-6, "td target")


def check_wide_out_of_range(bk, S, N, B, in_dim, hid):
    """Operands beyond the f16 range of the 16-bit form (weights: |2^10 W| > 65000) in ONE agent: its workgroups recompute their tiles
    with the fp32 loop inside the same launch (k_wgemm16), everybody else stays on the 16-bit matrix core; both match the reference."""
    rng = np.random.default_rng(S * 1000 + N * 100 + B + in_dim + hid + 7)
    g = wgeom(in_dim, hid)
    ldp, ldb = pad64(g["P"]), pad64(B)
    params = wide_params(rng, S, N, in_dim, hid)
    for s in range(S):
        params[s][0][2] *= np.float32(2000.0)                   # W2 of agent 0: entries up to ~2000 * 0.3
        assert np.abs(params[s][0][2]).max() * 1024 > 65000
    theta = pack_rows(params, ldp)
    x = rng.normal(size=(S, B, in_dim)).astype(np.float32)
    d_x, d_th = bk.dev(x), bk.dev(theta)
    wb = WideBuffers(bk, S, N, B, hid)
    forward2(bk, wb, d_x, B * in_dim, d_th, S, N, B, in_dim, hid, ldp, ldb)
    a2 = bk.host(wb.a2)
    for s in range(S):
        for n in range(N):
            p = params[s][n]
            z1 = x[s] @ p[0] + p[1]
            w1 = np.where(z1 > 0, z1, np.float32(0.1) * z1)
            z2 = (w1.astype(np.float64) @ p[2].astype(np.float64) + p[3]).astype(np.float32)
            w2 = np.where(z2 > 0, z2, np.float32(0.1) * z2)
            assert np.isfinite(a2[s, n * hid:(n + 1) * hid, :B]).all()
            rel_close(a2[s, n * hid:(n + 1) * hid, :B].T / np.abs(w2).max(), w2 / np.abs(w2).max(), 4e-6, "layer-2 activations, agent %d" % n)


def check_wide_fit(bk, S, N, B, in_dim, hid, steps=2, lr=0.01, masked_agent=None):
    """`steps` full-batch SGD steps through the dense-GEMM path vs the oracle's fit (M.fit_mse)."""
    rng = np.random.default_rng(S * 1000 + N * 100 + B + in_dim + hid + 1)
    g = wgeom(in_dim, hid)
    ldp, ldb = pad64(g["P"]), pad64(B)
    params = wide_params(rng, S, N, in_dim, hid)
    theta = pack_rows(params, ldp)
    x = rng.normal(size=(S, B, in_dim)).astype(np.float32)
    yv = rng.normal(size=(S, N, ldb)).astype(np.float32)
    mask = np.ones(N, np.int32)
    if masked_agent is not None:
        mask[masked_agent] = 0
    d_x, d_y, d_mask, d_msg = bk.dev(x), bk.dev(yv), bk.dev(mask), bk.dev(theta.copy())
    d_loss = bk.dev(np.zeros((S, N), np.float32))
    wb = WideBuffers(bk, S, N, B, hid)
    for st in range(steps):
        fit_step(bk, wb, d_x, B * in_dim, d_msg, d_y, d_mask, d_loss if st == 0 else None, S, N, B, in_dim, hid, ldp, ldb, lr)
    msg, loss = bk.host(d_msg), bk.host(d_loss)
    for s in range(S):
        for n in range(N):
            if not mask[n]:
                np.testing.assert_array_equal(msg[s, n], theta[s, n])
                continue
            pw = M.copy_params(params[s][n])
            hist = M.fit_mse(pw, x[s], yv[s, n, :B, None], lr, epochs=steps)
            got = unpack_row(msg[s, n], in_dim, 1, hid)
            for k in range(6):
                rel_close(got[k], pw[k], 1e-5, "wide fit param %d" % k)
            assert abs(loss[s, n] - hist[0]) <= 1e-5 * max(1.0, abs(hist[0])), (loss[s, n], hist[0])


def check_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph="circ"):
    """K2+K3 for a wide head vs the oracle agent (consensus_estimates_critic + projection_step_critic)."""
    rng = np.random.default_rng(S + N * 10 + B + d * 7 + H + hid)
    g = wgeom(in_dim, hid)
    P, P_hid = g["P"], g["P_hid"]
    ldp, ldb = pad64(P), pad64(B)
    live = wide_params(rng, S, N, in_dim, hid)
    msgp = wide_params(rng, S, N, in_dim, hid)
    theta, msg = pack_rows(live, ldp), pack_rows(msgp, ldp)
    x = rng.normal(size=(S, B, in_dim)).astype(np.float32)
    nbr = circulant(N, d) if graph == "circ" else random_regular(N, d, rng)
    coop = np.ones(N, np.int32)
    coop[0] = 0
    for s in range(S):                              # an outlier head among the messages
        msg[s, 1, P_hid:P] *= 50.0
        msgp[s][1][4] = msgp[s][1][4] * np.float32(50.0)
        msgp[s][1][5] = msgp[s][1][5] * np.float32(50.0)
    d_x, d_th, d_msg, d_nbr, d_coop = bk.dev(x), bk.dev(theta), bk.dev(msg), bk.dev(nbr), bk.dev(coop)
    d_agg = bk.dev(np.zeros((S, N, ldb), np.float32))
    wb = WideBuffers(bk, S, N, B, hid, d)
    L = bk.lib
    forward2(bk, wb, d_x, B * in_dim, d_th, S, N, B, in_dim, hid, ldp, ldb)
    L.rcmarl_wide_consensus_head(bk.ptr(wb.a2), bk.ptr(d_th), bk.ptr(d_msg), bk.ptr(d_nbr), bk.ptr(d_coop), None,
                                 bk.ptr(wb.hmat), bk.ptr(wb.hb), bk.ptr(wb.est), bk.ptr(wb.ebuf), bk.ptr(wb.grads),
                                 bk.ptr(d_agg), S, N, B, in_dim, hid, ldp, ldb, d, H, bk.stream)
    L.rcmarl_wide_head_apply(bk.ptr(wb.grads), bk.ptr(d_th), bk.ptr(d_coop), S, N, B, in_dim, hid, ldp, bk.stream)
    th_new, agg = bk.host(d_th), bk.host(d_agg)
    # K3 alone toward the aggregate just computed, from the original heads: must land on the same W3, b3
    d_th2 = bk.dev(theta)
    L.rcmarl_wide_consensus_head(bk.ptr(wb.a2), bk.ptr(d_th2), None, None, bk.ptr(d_coop), bk.ptr(d_agg), None, None, None,
                                 bk.ptr(wb.ebuf), bk.ptr(wb.grads), None, S, N, B, in_dim, hid, ldp, ldb, d, H, bk.stream)
    L.rcmarl_wide_head_apply(bk.ptr(wb.grads), bk.ptr(d_th2), bk.ptr(d_coop), S, N, B, in_dim, hid, ldp, bk.stream)
    th_proj = bk.host(d_th2)
    # the same step with |phi|^2 per replay row handed over as per-tile parts (rcmarl_wide_consensus_head_nrm: what the packed-operand
    # forward pass leaves beside its fp32 activations): two parts here, summed in fp64 on the host
    a2h = bk.host(wb.a2).reshape(S, N, hid, ldb).astype(np.float64)
    h2 = hid // 2
    nparts = np.stack([(a2h[:, :, :h2] ** 2).sum(axis=2), (a2h[:, :, h2:] ** 2).sum(axis=2)], axis=2).astype(np.float32)     # [S][N][2][ldb]
    d_th3, d_np, d_agg3 = bk.dev(theta), bk.dev(nparts), bk.dev(np.zeros((S, N, ldb), np.float32))
    L.rcmarl_wide_consensus_head_nrm(bk.ptr(wb.a2), bk.ptr(d_np), 2, bk.ptr(d_th3), bk.ptr(d_msg), bk.ptr(d_nbr), bk.ptr(d_coop),
                                     bk.ptr(wb.hmat), bk.ptr(wb.hb), bk.ptr(wb.est), bk.ptr(wb.ebuf), bk.ptr(wb.grads), bk.ptr(d_agg3), S, N,
                                     B, in_dim, hid, ldp, ldb, d, H, bk.stream)
    L.rcmarl_wide_head_apply(bk.ptr(wb.grads), bk.ptr(d_th3), bk.ptr(d_coop), S, N, B, in_dim, hid, ldp, bk.stream)
    th_nrm, agg3 = bk.host(d_th3), bk.host(d_agg3)
    np.testing.assert_array_equal(agg3, agg)                              # (the aggregate does not depend on the norm)
    for s in range(S):
        for i in range(N):
            if not coop[i]:
                np.testing.assert_array_equal(th_new[s, i], theta[s, i])
                continue
            rel_close(th_nrm[s, i, :P], th_new[s, i, :P], 2e-6, "projection with |phi|^2 supplied as parts")
            ag = O.CoopAgent(M.init_mlp(rng, in_dim, 20, 5), live[s][i], live[s][i], 0.002, 0.01, 0.9, H)
            want_agg = ag.consensus_estimates_critic(x[s], [msgp[s][j] for j in nbr[i]])
            # fp32 summation order only: a 512-term head on top of two GEMM layers carries ~4x the roundoff of a 128-term one.
            # Layer 2 on the 16-bit matrix core (two f16 pieces per operand, the l*l product dropped: the default form): each of the
            # d + 1 estimates carries operands good to 2^-22 instead of 2^-24 -- measured 1.34e-5 on the (24 agents, 128 units, d = 10)
            # case that sits at 1e-5 in the fp32 form -> 2e-5 there
            f16 = bool(bk.lib.rcmarl_wide_f16_mode())
            rel_close(agg[s, i, :B], want_agg[:, 0], (2e-5 if f16 else 1e-5) if hid <= 128 else 4e-5, "estimate aggregate")
            ag.projection_step_critic(x[s], want_agg)
            got = unpack_row(th_new[s, i], in_dim, 1, hid)
            for k in range(4):
                np.testing.assert_array_equal(got[k], live[s][i][k])          # hidden layers frozen
            rel_close(got[4], ag.critic[4], 3e-5 if hid <= 128 else 1e-4, "W3 after projection")
            rel_close(got[5], ag.critic[5], 3e-5 if hid <= 128 else 1e-4, "b3 after projection")
            rel_close(th_proj[s, i, :P], th_new[s, i, :P], 2e-6, "projection toward a given aggregate")


# ---- dense layers on pre-split packed operands (csrc/dense_pk.hip) ----------------------------------------------------
class PkBuffers:
    """Packed operands + scratch of the rcmarl_pk_* path for S seeds x N agents of an (in_dim -> hid -> hid -> 1) net, B rows."""

    def __init__(self, bk, S, N, B, in_dim, hid):
        from rcmarl_amd import lattice as LT
        Z, Bp, JT, JK, ldb = S * N, (B + 255) // 256 * 256, hid // 128, hid // 32, pad64(B)
        self.g = g = LT.Geometry(N, in_dim, B, hid)
        self.Bp, self.JT, self.JK = Bp, JT, JK
        u8 = lambda n: bk.dev(np.zeros(int(n), np.uint8))
        f32 = lambda *sh: bk.dev(np.zeros(sh, np.float32))
        nb = LT.Geometry.nbytes
        self.kp, self.ktp = u8(S * nb(g.kp, 1)), u8(S * nb(g.ktp, 1))
        self.wp, self.dzp = u8(S * nb(g.wp, 3)), u8(S * nb(g.dzp, 3))
        self.flag = bk.dev(np.zeros(1, np.int32))
        self.bk_rt, self.kb_kt = Bp // 128, Bp // 32
        self.a1_bk, self.a1_kb = u8(Z * self.bk_rt * JK * 2 * 8192), u8(Z * JT * self.kb_kt * 2 * 8192)
        self.s1 = bk.dev(np.zeros((Z * hid, Bp // 32), np.int32))
        self.w2t, self.w2w3, self.rs = u8(Z * JT * JK * 2 * 8192), u8(Z * JT * JK * 2 * 8192), f32(Z, hid)
        self.mask_bj, self.mask_jb = u8(Z * self.bk_rt * JK * 8192), u8(Z * JT * self.kb_kt * 8192)
        self.vpart, self.npart, self.dz3 = f32(Z, JT, ldb), f32(Z, JT, ldb), f32(Z, ldb)
        self.dzv = bk.dev(np.zeros((Z, 4, Bp), np.int16))
        self.losspart = f32(Z, (B + 255) // 256)
        self.gw3part, self.q, self.gb1part = f32(Z, JT, hid), f32(Z, hid), f32(Z, (B + 127) // 128, hid)
        self.a2 = f32(S, N * hid, ldb)
        self.ovf = bk.dev(np.zeros(1, np.int32))


def pk_encode(bk, pb, d_x, x_stride, d_alpha, S, B, in_dim):
    g = pb.g
    bk.lib.rcmarl_lattice_encode(bk.ptr(d_x), x_stride, bk.ptr(d_alpha), S, B, in_dim, bk.ptr(pb.kp), g.kp[0], g.kp[1], bk.ptr(pb.ktp),
                                 g.ktp[0], g.ktp[1], bk.ptr(pb.flag), bk.stream)


def pk_forward(bk, pb, d_alpha, d_theta, S, N, B, in_dim, hid, ldp, ldb, split=True, fit=True, want_a2=False):
    """layer 1 (lattice GEMM, packed outputs) + pack of W2 + layer 2 (masks / value parts / fp32 a2)"""
    g, L = pb.g, bk.lib
    if split:
        L.rcmarl_w1_split(bk.ptr(d_theta), bk.ptr(d_alpha), bk.ptr(pb.wp), S, N, in_dim, hid, ldp, g.wp[0], g.wp[1], bk.stream)
    L.rcmarl_layer1_forward_lattice_pk(bk.ptr(pb.kp), g.kp[0], g.kp[1], bk.ptr(pb.wp), g.wp[0], g.wp[1], bk.ptr(d_theta),
                                       bk.ptr(pb.a1_bk), pb.bk_rt, bk.ptr(pb.a1_kb) if fit else None, pb.kb_kt,
                                       bk.ptr(pb.s1) if fit else None, pb.Bp // 32, bk.ptr(pb.ovf), S, N, B, in_dim, hid, ldp, bk.stream)
    L.rcmarl_pk_pack_w2(bk.ptr(d_theta), bk.ptr(pb.w2t), bk.ptr(pb.w2w3), bk.ptr(pb.rs), bk.ptr(pb.ovf), S, N, in_dim, hid, ldp, bk.stream)
    L.rcmarl_pk_forward2(bk.ptr(pb.w2t), bk.ptr(pb.a1_bk), pb.bk_rt, bk.ptr(d_theta), bk.ptr(pb.a2) if want_a2 else None,
                         bk.ptr(pb.mask_bj) if fit else None, pb.bk_rt, bk.ptr(pb.mask_jb) if fit else None, pb.kb_kt,
                         bk.ptr(pb.vpart), bk.ptr(pb.npart) if want_a2 else None, S, N, B, in_dim, hid, ldp, ldb, bk.stream)


def pk_fit_step(bk, pb, d_alpha, d_msg, d_y, d_mask, d_loss, S, N, B, in_dim, hid, ldp, ldb, lr, split, emit_wp=True):
    """one full-batch SGD step of fit() on the packed-operand path (the sequence engine._local_fit_wide_pk runs)"""
    g, L = pb.g, bk.lib
    pk_forward(bk, pb, d_alpha, d_msg, S, N, B, in_dim, hid, ldp, ldb, split=split)
    L.rcmarl_pk_head(bk.ptr(pb.vpart), bk.ptr(d_msg), bk.ptr(d_y), 0.0, 2, bk.ptr(pb.dz3), bk.ptr(pb.dzv), bk.ptr(pb.losspart), S, N, B,
                     in_dim, hid, ldp, ldb, bk.stream)
    L.rcmarl_pk_backward_data(bk.ptr(pb.mask_bj), pb.bk_rt, bk.ptr(pb.w2w3), bk.ptr(pb.rs), bk.ptr(pb.s1), pb.Bp // 32, bk.ptr(pb.dz3),
                              bk.ptr(pb.dzp), g.dzp[0], g.dzp[1], bk.ptr(pb.gb1part), bk.ptr(pb.ovf), S, N, B, hid, ldb, bk.stream)
    L.rcmarl_pk_backward_w2(bk.ptr(pb.a1_kb), pb.kb_kt, bk.ptr(pb.mask_jb), pb.kb_kt, bk.ptr(pb.dzv), bk.ptr(d_msg), bk.ptr(d_mask),
                            bk.ptr(pb.gw3part), bk.ptr(pb.q), S, N, B, in_dim, hid, ldp, lr, bk.stream)
    L.rcmarl_layer1_backward_sgd_lattice(bk.ptr(pb.ktp), g.ktp[0], g.ktp[1], bk.ptr(pb.dzp), g.dzp[0], g.dzp[1], bk.ptr(d_alpha),
                                         bk.ptr(d_msg), bk.ptr(d_mask), S, N, B, in_dim, hid, ldp, lr,
                                         bk.ptr(pb.wp) if emit_wp else None, g.wp[0], g.wp[1], bk.stream)
    L.rcmarl_pk_small_sgd(bk.ptr(pb.gw3part), bk.ptr(pb.q), bk.ptr(pb.gb1part), bk.ptr(pb.dz3), bk.ptr(pb.losspart), bk.ptr(d_msg),
                          bk.ptr(d_mask), bk.ptr(d_loss), S, N, B, in_dim, hid, ldp, ldb, lr, bk.stream)


def _wide_lattice_case(S, N, B, width, nrow, ncol, hid, seed_off=0):
    from kernel_checks import lattice_rows
    rng = np.random.default_rng(S * 1000 + N * 100 + B + width + hid + seed_off)
    in_dim = N * width
    g = wgeom(in_dim, hid)
    ldp, ldb = pad64(g["P"]), pad64(B)
    params = wide_params(rng, S, N, in_dim, hid)
    x, alpha = lattice_rows(rng, S, B, N, width, nrow, ncol)
    return rng, in_dim, g, ldp, ldb, params, x, alpha


def check_pk_forward(bk, S, N, B, width, nrow, ncol, hid):
    """values, TD targets and the fp32 layer-2 activations of the packed-operand path vs the oracle's forward pass; the packed
    activations decode to a1 (two f16 pieces of 2^6 a1) in both orientations and the sign words to [a1 > 0]."""
    from rcmarl_amd import lattice as LT
    assert bk.lib.rcmarl_pk_supported(hid) == 1
    rng, in_dim, g, ldp, ldb, params, x, alpha = _wide_lattice_case(S, N, B, width, nrow, ncol, hid)
    theta = pack_rows(params, ldp)
    r = rng.normal(size=(S, N, ldb)).astype(np.float32)
    d_x, d_al, d_th, d_r = bk.dev(x), bk.dev(alpha), bk.dev(theta), bk.dev(r)
    pb = PkBuffers(bk, S, N, B, in_dim, hid)
    pk_encode(bk, pb, d_x, B * in_dim, d_al, S, B, in_dim)
    pk_forward(bk, pb, d_al, d_th, S, N, B, in_dim, hid, ldp, ldb, want_a2=True)
    d_v, d_y = bk.dev(np.zeros((S, N, ldb), np.float32)), bk.dev(np.zeros((S, N, ldb), np.float32))
    L = bk.lib
    L.rcmarl_pk_head(bk.ptr(pb.vpart), bk.ptr(d_th), None, 0.0, 0, bk.ptr(d_v), None, None, S, N, B, in_dim, hid, ldp, ldb, bk.stream)
    L.rcmarl_pk_head(bk.ptr(pb.vpart), bk.ptr(d_th), bk.ptr(d_r), 0.9, 1, bk.ptr(d_y), None, None, S, N, B, in_dim, hid, ldp, ldb,
                     bk.stream)
    assert bk.host(pb.flag)[0] == 0 and bk.host(pb.ovf)[0] == 0
    a2, v, y = bk.host(pb.a2), bk.host(d_v), bk.host(d_y)
    Bp, JT, JK = pb.Bp, pb.JT, pb.JK
    bk16 = np.asarray(bk.host(pb.a1_bk)).view(np.uint16).reshape(S * N, -1)
    kb16 = np.asarray(bk.host(pb.a1_kb)).view(np.uint16).reshape(S * N, -1)
    s1 = np.asarray(bk.host(pb.s1)).view(np.uint32).reshape(S * N, hid, Bp // 32)
    for s in range(S):
        for n in range(N):
            p = params[s][n]
            z1 = x[s] @ p[0] + p[1]
            w1 = np.where(z1 > 0, z1, np.float32(0.1) * z1)
            z2 = w1 @ p[2] + p[3]
            w2 = np.where(z2 > 0, z2, np.float32(0.1) * z2)
            zi = s * N + n
            dec = LT.pk_unpack(bk16[zi], B, hid, JK, 2, f16=True)                   # [2][B][hid]
            rel_close((dec[0] + dec[1]) / 64.0, w1, 3e-6, "a1_bk (rows = replay row)")
            dect = LT.pk_unpack(kb16[zi], hid, B, pb.kb_kt, 2, f16=True)            # [2][hid][B]
            np.testing.assert_array_equal(dect[0], dec[0].T)
            np.testing.assert_array_equal(dect[1], dec[1].T)
            bits = (s1[zi][:, np.arange(B) >> 5] >> (np.arange(B) & 31).astype(np.uint32)) & 1
            np.testing.assert_array_equal(bits.astype(bool), (dec[0] + dec[1]).T > 0)
            rel_close(a2[s, n * hid:(n + 1) * hid, :B].T, w2, 6e-6, "layer-2 activations")
            want = M.forward(p, x[s])[:, 0]
            rel_close(v[s, n, :B], want, 8e-6, "value")
            rel_close(y[s, n, :B], r[s, n, :B] + np.float32(0.9) * want, 8e-6, "td target")


def knife_edge_ratio(p0, x, y, lr, steps):
    """fp64 replay of the oracle's full-batch fit from p0: the smallest |pre-activation| / sum |terms| over both hidden layers, all rows
    and all `steps` steps.  A LeakyReLU input within rounding of zero (ratio ~1e-7 or less) takes the other slope in one of two fp32
    chains that differ in rounding order: that unit's weights then move by about lr * 0.9 * |dz| * |x| / B -- a "knife edge"."""
    pw = M.copy_params(p0)
    worst = np.inf
    for _ in range(steps):
        p64 = [q.astype(np.float64) for q in pw]
        x64 = x.astype(np.float64)
        z1 = x64 @ p64[0] + p64[1]
        a1 = np.where(z1 > 0, z1, 0.1 * z1)
        z2 = a1 @ p64[2] + p64[3]
        s1 = np.abs(x64) @ np.abs(p64[0]) + np.abs(p64[1])
        s2 = np.abs(a1) @ np.abs(p64[2]) + np.abs(p64[3])
        worst = min(worst, float((np.abs(z1) / s1).min()), float((np.abs(z2) / s2).min()))
        M.fit_mse(pw, x, y, lr, epochs=1)
    return worst


def check_pk_fit(bk, S, N, B, width, nrow, ncol, hid, steps=2, lr=0.01, masked_agent=None, tol=2e-5, knife_tol=1e-3):
    """`steps` full-batch SGD steps on the packed-operand path vs the oracle's fit (M.fit_mse).  A network beyond `tol` must be a
    PROVEN knife edge (knife_edge_ratio below 2e-7: one of its ~B * hid * steps LeakyReLU inputs sits within fp32 rounding of zero)
    and stay within knife_tol."""
    assert bk.lib.rcmarl_pk_supported(hid) == 1
    rng, in_dim, g, ldp, ldb, params, x, alpha = _wide_lattice_case(S, N, B, width, nrow, ncol, hid, 1)
    theta = pack_rows(params, ldp)
    yv = rng.normal(size=(S, N, ldb)).astype(np.float32)
    mask = np.ones(N, np.int32)
    if masked_agent is not None:
        mask[masked_agent] = 0
    d_x, d_al, d_y, d_mask, d_msg = bk.dev(x), bk.dev(alpha), bk.dev(yv), bk.dev(mask), bk.dev(theta.copy())
    d_loss = bk.dev(np.zeros((S, N), np.float32))
    pb = PkBuffers(bk, S, N, B, in_dim, hid)
    pk_encode(bk, pb, d_x, B * in_dim, d_al, S, B, in_dim)
    for st in range(steps):
        pk_fit_step(bk, pb, d_al, d_msg, d_y, d_mask, d_loss if st == 0 else None, S, N, B, in_dim, hid, ldp, ldb, lr, split=(st == 0))
    assert bk.host(pb.flag)[0] == 0 and bk.host(pb.ovf)[0] == 0
    msg, loss = bk.host(d_msg), bk.host(d_loss)
    worst = 0.0
    for s in range(S):
        for n in range(N):
            if not mask[n]:
                np.testing.assert_array_equal(msg[s, n], theta[s, n])
                continue
            pw = M.copy_params(params[s][n])
            hist = M.fit_mse(pw, x[s], yv[s, n, :B, None], lr, epochs=steps)
            got = unpack_row(msg[s, n], in_dim, 1, hid)
            err = max(float(np.max(np.abs(got[k] - pw[k])) / max(1.0, float(np.max(np.abs(pw[k]))))) for k in range(6))
            if err > tol:
                ratio = knife_edge_ratio(params[s][n], x[s], yv[s, n, :B, None], lr, steps)
                assert ratio < 2e-7 and err <= knife_tol, "packed-operand fit, seed %d agent %d: error %.2e beyond %.1e without a knife " \
                    "edge to explain it (smallest |z| / sum|terms| %.1e)" % (s, n, err, tol, ratio)
                print("packed-operand fit: seed %d agent %d is a knife edge (|z| / sum|terms| = %.1e): error %.2e" % (s, n, ratio, err))
            else:
                worst = max(worst, err)
            assert abs(loss[s, n] - hist[0]) <= 1e-5 * max(1.0, abs(hist[0])), (loss[s, n], hist[0])
    return worst


def check_pk_range_flag(bk, S=1, N=2, B=150, width=2, nrow=5, ncol=5, hid=128):
    """Operands beyond the f16 range of the packed form: the pieces saturate (finite outputs) and the producers raise the caller's flag
    -- W2 of one agent scaled so that 2^10 |W2| > 65000 trips rcmarl_pk_pack_w2, a huge b1 trips the layer-1 epilogue."""
    rng, in_dim, g, ldp, ldb, params, x, alpha = _wide_lattice_case(S, N, B, width, nrow, ncol, hid, 3)
    for which in ("w2", "b1"):
        p2 = [[M.copy_params(p) for p in row] for row in params]
        if which == "w2":
            p2[0][1][2] *= np.float32(2000.0)
        else:
            p2[0][0][1] += np.float32(5000.0)
        d_x, d_al, d_th = bk.dev(x), bk.dev(alpha), bk.dev(pack_rows(p2, ldp))
        pb = PkBuffers(bk, S, N, B, in_dim, hid)
        pk_encode(bk, pb, d_x, B * in_dim, d_al, S, B, in_dim)
        pk_forward(bk, pb, d_al, d_th, S, N, B, in_dim, hid, ldp, ldb, want_a2=True)
        assert bk.host(pb.ovf)[0] == 1, which
        assert np.isfinite(bk.host(pb.a2)).all() and np.isfinite(bk.host(pb.vpart)).all()
