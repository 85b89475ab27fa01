#!/usr/bin/env python
"""Where do the packed-operand fit and the oracle's fp32 fit differ?  (run on the GPU box)

    python tools/diag_pk_knife.py S N B width nrow ncol hid steps lr [masked]

Prints, per network, the largest errors of W1 / b1 / W2 / b2 / W3 / b3 and the hidden units they sit in, and -- from an fp64 replay of the
oracle's chain -- the smallest |pre-activation| / scale of both layers at every step: a LeakyReLU input within rounding of zero takes
the other slope in one of the two chains (a "knife edge"), which moves ONE unit's weights by about lr * 0.9 * |dz| * |x| / B."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wide_checks as WC  # noqa: E402
from oracle import mlp_np as M  # noqa: E402
from rcmarl_amd import capi  # noqa: E402


class Bk:
    lib = capi.load()
    stream = None
    dev = staticmethod(lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda())
    ptr = staticmethod(lambda h: None if h is None else h.data_ptr())

    @staticmethod
    def host(h):
        torch.cuda.synchronize()
        return h.cpu().numpy()


def main():
    a = sys.argv[1:]
    S, N, B, width, nrow, ncol, hid, steps = (int(v) for v in a[:8])
    lr = float(a[8])
    masked = int(a[9]) if len(a) > 9 else None
    bk = Bk()
    bk.stream = torch.cuda.current_stream().cuda_stream
    rng, in_dim, g, ldp, ldb, params, x, alpha = WC._wide_lattice_case(S, N, B, width, nrow, ncol, hid, 1)
    theta = WC.pack_rows(params, ldp)
    yv = rng.normal(size=(S, N, ldb)).astype(np.float32)
    mask = np.ones(N, np.int32)
    if masked is not None:
        mask[masked] = 0
    d_x, d_al, d_y, d_mask, d_msg = bk.dev(x), bk.dev(alpha), bk.dev(yv), bk.dev(mask), bk.dev(theta.copy())
    d_loss = bk.dev(np.zeros((S, N), np.float32))
    pb = WC.PkBuffers(bk, S, N, B, in_dim, hid)
    WC.pk_encode(bk, pb, d_x, B * in_dim, d_al, S, B, in_dim)
    for st in range(steps):
        WC.pk_fit_step(bk, pb, d_al, d_msg, d_y, d_mask, d_loss if st == 0 else None, S, N, B, in_dim, hid, ldp, ldb, lr, split=(st == 0))
    msg = bk.host(d_msg)
    names = ("W1", "b1", "W2", "b2", "W3", "b3")
    for s in range(S):
        for n in range(N):
            if not mask[n]:
                continue
            pw = M.copy_params(params[s][n])
            # fp64 replay of the oracle chain: smallest relative |z| per step
            edges = []
            for st in range(steps):
                p64 = [q.astype(np.float64) for q in pw]
                z1 = x[s].astype(np.float64) @ p64[0] + p64[1]
                a1 = np.where(z1 > 0, z1, 0.1 * z1)
                z2 = a1 @ p64[2] + p64[3]
                s1 = np.abs(x[s]).astype(np.float64) @ np.abs(p64[0]) + np.abs(p64[1])
                s2 = np.abs(a1) @ np.abs(p64[2]) + np.abs(p64[3])
                r1, r2 = np.abs(z1) / s1, np.abs(z2) / s2
                i1, i2 = np.unravel_index(r1.argmin(), r1.shape), np.unravel_index(r2.argmin(), r2.shape)
                edges.append("step %d: min|z1|/sum|terms| %.1e (row %d unit %d)  min|z2|/sum|terms| %.1e (row %d unit %d)"
                             % (st, r1.min(), i1[0], i1[1], r2.min(), i2[0], i2[1]))
                M.fit_mse(pw, x[s], yv[s, n, :B, None], lr, epochs=1)
            got = WC.unpack_row(msg[s, n], in_dim, 1, hid)
            errs = [float(np.abs(got[k] - pw[k]).max()) for k in range(6)]
            flag = max(errs) > 1e-6
            print("seed %d agent %2d  max|err| %s%s" % (s, n, " ".join("%s %.1e" % (names[k], errs[k]) for k in range(6)), "   <--" if flag else ""))
            if flag:
                e1 = np.abs(got[0] - pw[0]).max(axis=0)
                e2 = np.abs(got[2] - pw[2])
                print("    W1 columns (layer-1 units) beyond 1e-6:", np.nonzero(e1 > 1e-6)[0][:12], " W2 rows:", np.nonzero(e2.max(axis=1) > 1e-6)[0][:12],
                      " W2 columns:", np.nonzero(e2.max(axis=0) > 1e-6)[0][:12])
                for e in edges:
                    print("    " + e)


if __name__ == "__main__":
    main()
