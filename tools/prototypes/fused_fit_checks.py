"""Parity checks of the fused local-fit PROTOTYPES (tools/prototypes/fused_fit.hip; not built since round 5, ABI 3).  Kept as a record:
they were part of tests/kernel_checks.py (bk = tests/backend.py object) and ran green on the MI355X and on the emulation in round 4."""
# flake8: noqa
class FusedBuffers:
    """Operand images of the fused local fit (csrc/fused_fit.hip), NaN-filled."""

    def __init__(self, bk, S, N, in_dim, rows_alloc):
        import ctypes
        kf, ktf, wf = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
        bk.lib.rcmarl_fit_fused_geometry(N, in_dim, HID, rows_alloc, ctypes.byref(kf), ctypes.byref(ktf), ctypes.byref(wf))
        z = lambda nbytes: bk.dev(np.full(S * nbytes // 2, 0x7e00, np.uint16))       # f16 NaN fill
        self.rows_alloc = rows_alloc
        self.nbytes = (kf.value, ktf.value, wf.value)
        self.kf, self.ktf, self.wf = z(kf.value), z(ktf.value), z(wf.value)


def check_fit_encode(bk, S, n_agents, B, width, nrow, ncol):
    """rcmarl_fit_encode: both fragment-major images hold the lattice integers (f16), zero beyond B / in_dim."""
    rng = np.random.default_rng(B * 5 + n_agents + width)
    x, alpha = lattice_rows(rng, S, B, n_agents, width, nrow, ncol)
    in_dim = n_agents * width
    rows_alloc = (B + 255) // 256 * 256 + 256
    fb = FusedBuffers(bk, S, n_agents, in_dim, rows_alloc)
    d_x, d_al = bk.dev(x), bk.dev(alpha)                  # (named: a temporary would be freed, and reused, before the launch)
    bk.lib.rcmarl_fit_encode(bk.ptr(d_x), B * in_dim, bk.ptr(d_al), S, B, in_dim, rows_alloc, bk.ptr(fb.kf), bk.ptr(fb.ktf),
                             bk.stream)
    K = np.rint(x.astype(np.float64) / alpha.astype(np.float64)).astype(np.float32)
    ftiles = (in_dim + 31) // 32
    KS, RS, b_pad = 2 * ftiles, rows_alloc // 16, (B + 255) // 256 * 256
    kf = bk.host(fb.kf).view(np.float16).reshape(S, rows_alloc // 32, KS, 64, 8).astype(np.float32)
    ktf = bk.host(fb.ktf).view(np.float16).reshape(S, ftiles, RS, 64, 8).astype(np.float32)
    Kpad = np.zeros((S, b_pad, ftiles * 32), np.float32)
    Kpad[:, :B, :in_dim] = K
    lane = np.arange(64)
    for s in range(S):
        # Kf[rt][ks][l][e] = K[32 rt + (l & 31)][16 ks + 8 (l >> 5) + e]
        want = Kpad[s].reshape(b_pad // 32, 32, KS, 2, 8)[:, lane & 31][:, :, :, :, :]          # [rt][l][ks][kg][e]
        want = want[:, np.arange(64), :, lane >> 5, :]                                          # -> [l][rt][ks][e]
        np.testing.assert_array_equal(kf[s, :b_pad // 32], want.transpose(1, 2, 0, 3))
        # KTf[ft][rs][l][e] = K[16 rs + 8 (l >> 5) + e][32 ft + (l & 31)]
        wt = Kpad[s].T.reshape(ftiles, 32, b_pad // 16, 2, 8)[:, lane & 31]
        wt = wt[:, np.arange(64), :, lane >> 5, :]                                              # [l][ft][rs][e]
        np.testing.assert_array_equal(ktf[s, :, :b_pad // 16], wt.transpose(1, 2, 0, 3))


def check_fused_fit(bk, S, N, B, width, nrow, ncol, steps=2, lr=0.01, gamma=0.9, masked_agent=None, vs_unfused=True):
    """The whole local fit in one launch (rcmarl_fit_encode + rcmarl_fit_fused) against the oracle's fit_mse, and against
    the three-launch lattice path in the same operand form (same arithmetic per element; the row sums associate differently)."""
    rng = np.random.default_rng(S * 1000 + N * 100 + B + width)
    in_dim = N * width
    P, _ = geom(in_dim, 1)
    ldp, ldb = pad64(P), pad64(B)
    params = random_params(rng, S, N, in_dim, 1)
    theta = pack_rows(params, ldp)
    x, alpha = lattice_rows(rng, S, B, N, width, nrow, ncol)
    target = rng.normal(size=(S, N, ldb)).astype(np.float32)
    mask = np.ones(N, np.int32)
    if masked_agent is not None:
        mask[masked_agent] = 0
    rows_alloc = (B + 255) // 256 * 256
    fb = FusedBuffers(bk, S, N, in_dim, rows_alloc)
    d_x, d_al, d_y, d_mask = bk.dev(x), bk.dev(alpha), bk.dev(target), bk.dev(mask)
    d_msg = bk.dev(theta.copy())
    d_loss = bk.dev(np.zeros((S, N), np.float32))
    d_flags = bk.dev(np.zeros((S, N), np.int32))
    L = bk.lib
    L.rcmarl_fit_encode(bk.ptr(d_x), B * in_dim, bk.ptr(d_al), S, B, in_dim, rows_alloc, bk.ptr(fb.kf), bk.ptr(fb.ktf), bk.stream)
    L.rcmarl_fit_fused(bk.ptr(fb.kf), bk.ptr(fb.ktf), bk.ptr(fb.wf), bk.ptr(d_al), bk.ptr(d_msg), bk.ptr(d_y), bk.ptr(d_mask),
                       bk.ptr(d_loss), bk.ptr(d_flags), S, N, B, in_dim, HID, ldp, ldb, rows_alloc, steps, lr, bk.stream)
    msg, loss, flags = bk.host(d_msg), bk.host(d_loss), bk.host(d_flags)
    assert not flags.any()
    ref = None
    if vs_unfused and bk.lib.rcmarl_lattice_f16_mode() == 3:
        nchunk = (B + 255) // 256
        psz = bk.lib.rcmarl_fit_partial_size(HID)
        lb = LatticeBuffers(bk, S, N, in_dim, B)
        g = lb.g
        d_ref = bk.dev(theta.copy())
        d_a = bk.dev(np.zeros((S, N * HID, ldb), np.float32))
        d_part = bk.dev(np.zeros((S, N, nchunk, psz), np.float32))
        _encode(bk, lb, d_x, B * in_dim, d_al, S, B, in_dim)
        for st in range(steps):
            _layer1_lattice(bk, lb, d_al, d_ref, d_a, S, N, B, in_dim, ldp, ldb, split=(st == 0))
            L.rcmarl_mid_fit_lattice(bk.ptr(d_a), bk.ptr(d_ref), bk.ptr(d_y), bk.ptr(d_part), bk.ptr(lb.dzp), g.dzp[0], g.dzp[1],
                                     S, N, B, in_dim, HID, ldp, ldb, bk.ptr(_mid_flags(bk, S, N)), bk.stream)
            L.rcmarl_small_sgd(bk.ptr(d_part), bk.ptr(d_ref), bk.ptr(d_mask), None, S, N, B, in_dim, HID, ldp, lr, bk.stream)
            L.rcmarl_layer1_backward_sgd_lattice(bk.ptr(lb.ktp), g.ktp[0], g.ktp[1], bk.ptr(lb.dzp), g.dzp[0], g.dzp[1],
                                                 bk.ptr(d_al), bk.ptr(d_ref), bk.ptr(d_mask), S, N, B, in_dim, HID, ldp, lr,
                                                 bk.ptr(lb.wp), g.wp[0], g.wp[1], bk.stream)
        ref = bk.host(d_ref)
        worst = float(np.abs(msg - ref).max() / np.abs(ref).max())
        assert worst <= 2e-6, worst
    # against the oracle: 1e-5 like the three-launch path; where that path itself measures more on a shape (many steps, wide
    # inputs), the fused fit gets what the three-launch path needs (both are printed)
    worst_f, worst_u = 0.0, 0.0
    for s in range(S):
        for n in range(N):
            if not mask[n]:
                np.testing.assert_array_equal(msg[s, n], theta[s, n])
                continue
            pw = M.copy_params(params[s][n])
            hist = M.fit_mse(pw, x[s], target[s, n, :B, None], lr, epochs=steps)
            got = unpack_row(msg[s, n], in_dim, 1)
            gu = unpack_row(ref[s, n], in_dim, 1) if ref is not None else None
            for k in range(6):
                scale = max(1.0, float(np.abs(pw[k]).max()))
                worst_f = max(worst_f, float(np.abs(got[k] - pw[k]).max()) / scale)
                if gu is not None:
                    worst_u = max(worst_u, float(np.abs(gu[k] - pw[k]).max()) / scale)
            assert abs(loss[s, n] - hist[0]) <= 1e-5 * max(1.0, abs(hist[0])), (loss[s, n], hist[0])
    print("fused fit vs oracle %.2e (three-launch path on the same inputs %.2e)" % (worst_f, worst_u))
    assert worst_f <= max(1e-5, 1.1 * worst_u), (worst_f, worst_u)
    return msg


def check_forward_mid_fit(bk, S, N, B, width, nrow, ncol, steps=3, lr=0.01, masked_agent=None):
    """The local fit with forward + mid in one launch (rcmarl_fit_wf_split, then per step rcmarl_fit_w2_frags ->
    rcmarl_forward_mid -> rcmarl_small_sgd_records -> rcmarl_layer1_backward_sgd_lattice_wf) against the three-launch path in
    the same operand form: the packed dz1 image of every step is BIT-IDENTICAL (same pieces, same accumulation order), the
    records sum to the same gradients up to the association of the row sums; and against the oracle's fit."""
    assert bk.lib.rcmarl_lattice_f16_mode() == 3
    import ctypes
    rng = np.random.default_rng(S * 1000 + N * 100 + B + width + 5)
    in_dim = N * width
    P, _ = geom(in_dim, 1)
    ldp, ldb = pad64(P), pad64(B)
    params = random_params(rng, S, N, in_dim, 1)
    theta = pack_rows(params, ldp)
    x, alpha = lattice_rows(rng, S, B, N, width, nrow, ncol)
    target = rng.normal(size=(S, N, ldb)).astype(np.float32)
    mask = np.ones(N, np.int32)
    if masked_agent is not None:
        mask[masked_agent] = 0
    rows_alloc = (B + 255) // 256 * 256
    ntiles = rows_alloc // 256
    fb = FusedBuffers(bk, S, N, in_dim, rows_alloc)
    lb = LatticeBuffers(bk, S, N, in_dim, B)
    g = lb.g
    psz = bk.lib.rcmarl_fit_partial_size(HID)
    L = bk.lib
    d_x, d_al, d_y, d_mask = bk.dev(x), bk.dev(alpha), bk.dev(target), bk.dev(mask)
    d_a = bk.dev(np.zeros((S, N * HID, ldb), np.float32))
    d_w2f = bk.dev(np.zeros((S, N, 8192 // 4), np.uint32))
    d_flags = bk.dev(np.zeros((S, N), np.int32))
    d_loss = bk.dev(np.zeros((S, N), np.float32))
    d_msg, d_ref = bk.dev(theta.copy()), bk.dev(theta.copy())
    d_part = bk.dev(np.zeros((S, N, ntiles, psz), np.float32))
    d_part_ref = bk.dev(np.zeros((S, N, ntiles, psz), np.float32))
    dzp_a = bk.dev(np.zeros_like(bk.host(lb.dzp)))
    _encode(bk, lb, d_x, B * in_dim, d_al, S, B, in_dim)
    L.rcmarl_fit_encode(bk.ptr(d_x), B * in_dim, bk.ptr(d_al), S, B, in_dim, rows_alloc, bk.ptr(fb.kf), bk.ptr(fb.ktf), bk.stream)
    L.rcmarl_fit_wf_split(bk.ptr(d_msg), bk.ptr(d_al), bk.ptr(fb.wf), bk.ptr(d_flags), S, N, in_dim, HID, ldp, bk.stream)
    npc = 2
    for st in range(steps):
        # --- three launches
        _layer1_lattice(bk, lb, d_al, d_ref, d_a, S, N, B, in_dim, ldp, ldb, split=(st == 0))
        L.rcmarl_mid_fit_lattice(bk.ptr(d_a), bk.ptr(d_ref), bk.ptr(d_y), bk.ptr(d_part_ref), bk.ptr(lb.dzp), g.dzp[0], g.dzp[1],
                                 S, N, B, in_dim, HID, ldp, ldb, bk.ptr(_mid_flags(bk, S, N)), bk.stream)
        L.rcmarl_small_sgd(bk.ptr(d_part_ref), bk.ptr(d_ref), bk.ptr(d_mask), None, S, N, B, in_dim, HID, ldp, lr, bk.stream)
        L.rcmarl_layer1_backward_sgd_lattice(bk.ptr(lb.ktp), g.ktp[0], g.ktp[1], bk.ptr(lb.dzp), g.dzp[0], g.dzp[1],
                                             bk.ptr(d_al), bk.ptr(d_ref), bk.ptr(d_mask), S, N, B, in_dim, HID, ldp, lr,
                                             bk.ptr(lb.wp), g.wp[0], g.wp[1], bk.stream)
        # --- forward + mid in one
        L.rcmarl_fit_w2_frags(bk.ptr(d_msg), bk.ptr(d_w2f), bk.ptr(d_flags), S, N, in_dim, HID, ldp, bk.stream)
        L.rcmarl_forward_mid(bk.ptr(fb.kf), bk.ptr(fb.wf), bk.ptr(d_w2f), bk.ptr(d_msg), bk.ptr(d_y), bk.ptr(d_part), bk.ptr(dzp_a),
                             g.dzp[0], g.dzp[1], bk.ptr(d_flags), S, N, B, in_dim, HID, ldp, ldb, rows_alloc, bk.stream)
        if st == 0:          # the same weights went in: the packed dz1 images must agree bit for bit on every row both paths write
            a, b = seed_views(bk.host(dzp_a), S, g.dzp, npc), seed_views(bk.host(lb.dzp), S, g.dzp, npc)
            for s_ in range(S):
                for pc in range(npc):
                    idx = LT.pk_element_index(N * HID, (B + 31) // 32 * 32, g.dzp[1], npc, pc)
                    np.testing.assert_array_equal(a[s_][idx], b[s_][idx])
        L.rcmarl_small_sgd_records(bk.ptr(d_part), bk.ptr(d_msg), bk.ptr(d_mask), bk.ptr(d_loss) if st == 0 else None, S, N, B,
                                   in_dim, HID, ldp, ntiles, lr, bk.stream)
        L.rcmarl_layer1_backward_sgd_lattice_wf(bk.ptr(lb.ktp), g.ktp[0], g.ktp[1], bk.ptr(dzp_a), g.dzp[0], g.dzp[1], bk.ptr(d_al),
                                                bk.ptr(d_msg), bk.ptr(d_mask), S, N, B, in_dim, HID, ldp, lr, bk.ptr(fb.wf),
                                                bk.stream)
    msg, ref, flags = bk.host(d_msg), bk.host(d_ref), bk.host(d_flags)
    assert not flags.any()
    worst = float(np.abs(msg - ref).max() / np.abs(ref).max())
    assert worst <= 2e-6, worst
    # the forward operand the last backward left == a fresh split of the result
    left = bk.host(fb.wf).copy()
    L.rcmarl_fit_wf_split(bk.ptr(d_msg), bk.ptr(d_al), bk.ptr(fb.wf), bk.ptr(d_flags), S, N, in_dim, HID, ldp, bk.stream)
    np.testing.assert_array_equal(left, bk.host(fb.wf))
    worst_f = 0.0
    for s in range(S):
        for n in range(N):
            if not mask[n]:
                np.testing.assert_array_equal(msg[s, n], theta[s, n])
                continue
            pw = M.copy_params(params[s][n])
            M.fit_mse(pw, x[s], target[s, n, :B, None], lr, epochs=steps)
            got = unpack_row(msg[s, n], in_dim, 1)
            for k in range(6):
                worst_f = max(worst_f, float(np.abs(got[k] - pw[k]).max()) / max(1.0, float(np.abs(pw[k]).max())))
    assert worst_f <= 2e-5, worst_f
    return msg


