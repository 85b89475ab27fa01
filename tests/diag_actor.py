"""Diagnostic (GPU): error distribution of the actor Adam step vs the oracle for a large shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kernel_checks as KC
from oracle import mlp_np as M
from test_kernels_gpu import GpuBackend

bk = GpuBackend()
S, N, B, in_dim, steps, lr, A = 1, 64, 1000, 128, 3, 0.002, 5
rng = np.random.default_rng(S + N + B + in_dim)
P, _ = KC.geom(in_dim, A)
ldp, ldb = KC.pad64(P), KC.pad64(B)
params = KC.random_params(rng, S, N, in_dim, A)
theta = KC.pack_rows(params, ldp)
x = rng.normal(size=(S, B, in_dim)).astype(np.float32)
act = rng.integers(0, A, size=(S, N, ldb)).astype(np.float32)
delta = rng.normal(size=(S, N, ldb)).astype(np.float32)
mask = np.ones(N, np.int32)
nchunk = (B + 255) // 256
psz = bk.lib.rcmarl_actor_partial_size(20, A)
d_x, d_th, d_act, d_delta, d_mask = bk.dev(x), bk.dev(theta), bk.dev(act), bk.dev(delta), bk.dev(mask)
d_m, d_v = bk.dev(np.zeros_like(theta)), bk.dev(np.zeros_like(theta))
d_a = bk.dev(np.zeros((S, N * 20, ldb), np.float32))
d_part = bk.dev(np.zeros((S, N, nchunk, psz), np.float32))
d_loss = bk.dev(np.zeros((S, N), np.float32))
L = bk.lib
b1, b2, eps = 0.9, 0.999, 1e-7
for t in range(1, steps + 1):
    alpha = np.float32(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    KC._layer1(bk, d_x, B * in_dim, d_th, d_a, S, N, B, in_dim, ldp, ldb)
    L.rcmarl_mid_actor(bk.ptr(d_a), bk.ptr(d_th), bk.ptr(d_act), bk.ptr(d_delta), bk.ptr(d_part), S, N, B, in_dim, 20, A, ldp, ldb, bk.stream)
    L.rcmarl_small_adam(bk.ptr(d_part), bk.ptr(d_th), bk.ptr(d_m), bk.ptr(d_v), bk.ptr(d_mask), bk.ptr(d_loss), S, N, B, in_dim, 20, A, ldp, float(alpha), float(np.float32(1 - b1)), float(np.float32(1 - b2)), float(np.float32(eps)), bk.stream)
    L.rcmarl_layer1_backward_adam(bk.ptr(d_x), B * in_dim, bk.ptr(d_a), bk.ptr(d_th), bk.ptr(d_m), bk.ptr(d_v), bk.ptr(d_mask), S, N, B, in_dim, 20, ldp, ldb, float(alpha), float(np.float32(1 - b1)), float(np.float32(1 - b2)), float(np.float32(eps)), bk.stream)
th_new = bk.host(d_th)
errs = [[] for _ in range(6)]
for n in range(N):
    pw = M.copy_params(params[0][n]); st = M.AdamState(pw, lr)
    for t in range(steps):
        M.fit_actor_ce(pw, st, x[0], act[0, n, :B], delta[0, n, :B], epochs=1)
    got = KC.unpack_row(th_new[0, n], in_dim, A)
    for k in range(6):
        errs[k].append(np.abs(got[k] - pw[k]).ravel())
for k in range(6):
    e = np.concatenate(errs[k])
    print("array", k, "n", e.size, "max %.3e" % e.max(), "p99.9 %.3e" % np.quantile(e, 0.999), "frac>1.2e-4: %.2e" % np.mean(e > 1.2e-4), "frac>1e-5 %.2e" % np.mean(e > 1e-5), "RCMARL_GEMM", os.environ.get("RCMARL_GEMM"))
