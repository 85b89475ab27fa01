#!/usr/bin/env python
"""Kernel micro-benchmarks on synthetic cfg-4 shapes (run on the GPU box through gpurun).

    python tools/kbench.py gemm      # layer-1 GEMMs (variant via RCMARL_GEMM=0|1|2)
    python tools/kbench.py k1        # consensus_params at (d,H) = (4,1), (10,4), (18,8)
    python tools/kbench.py mid       # mid_fit / consensus_head / mid_value
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcmarl_amd import capi  # noqa: E402

HID = 20


def pad64(n):
    return (n + 63) // 64 * 64


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def gemm(L, S=16, N=256, B=3000):
    st = torch.cuda.current_stream().cuda_stream
    for in_dim in (2 * N, 3 * N):
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        x = torch.randn(S, B, in_dim, device="cuda")
        theta = torch.randn(S, N, ldp, device="cuda") * 0.05
        a1t = torch.zeros(S, N * HID, ldb, device="cuda")
        dz = torch.randn(S, N * HID, ldb, device="cuda") * 1e-3
        mask = torch.ones(N, dtype=torch.int32, device="cuda")
        flops = 2.0 * S * N * HID * B * in_dim
        t = timeit(lambda: L.rcmarl_layer1_forward(x.data_ptr(), B * in_dim, theta.data_ptr(), a1t.data_ptr(), S, N, B, in_dim,
                                                   HID, ldp, ldb, st))
        print("fwd  in=%4d  %8.1f us  %6.1f TF/s" % (in_dim, t, flops / t / 1e6))
        t = timeit(lambda: L.rcmarl_layer1_backward_sgd(x.data_ptr(), B * in_dim, dz.data_ptr(), theta.data_ptr(), mask.data_ptr(),
                                                        S, N, B, in_dim, HID, ldp, ldb, 1e-6, st))
        print("bwd  in=%4d  %8.1f us  %6.1f TF/s" % (in_dim, t, flops / t / 1e6))


def k1(L, S=16, N=256):
    st = torch.cuda.current_stream().cuda_stream
    for d, H in ((4, 1), (10, 4), (18, 8), (18, 1)):
        for in_dim in (2 * N, 3 * N):
            P = in_dim * HID + HID + HID * HID + HID + HID + 1
            P_hid = P - 21
            ldp = pad64(P)
            msg = torch.randn(S, N, ldp, device="cuda")
            theta = torch.zeros(S, N, ldp, device="cuda")
            nbr = torch.tensor([[(i + k) % N for k in range(d)] for i in range(N)], dtype=torch.int32, device="cuda")
            coop = torch.ones(N, dtype=torch.int32, device="cuda")
            t = timeit(lambda: L.rcmarl_consensus_params(msg.data_ptr(), theta.data_ptr(), nbr.data_ptr(), coop.data_ptr(), S, N,
                                                         ldp, P_hid, d, H, None, None, st), iters=20)
            byts = 8.0 * S * N * P_hid
            print("K1 d=%2d H=%d P_hid=%5d  %7.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (d, H, P_hid, t, byts / t / 1e3, byts / t / 1e3 / 80))


def mid(L, S=16, N=256, B=3000):
    st = torch.cuda.current_stream().cuda_stream
    in_dim = 2 * N
    P = in_dim * HID + HID + HID * HID + HID + HID + 1
    ldp, ldb = pad64(P), pad64(B)
    theta = torch.randn(S, N, ldp, device="cuda") * 0.05
    a1t = torch.randn(S, N * HID, ldb, device="cuda")
    y = torch.randn(S, N, ldb, device="cuda")
    nchunk = (B + 255) // 256
    part = torch.zeros(S * N * nchunk * L.rcmarl_fit_partial_size(HID), device="cuda")
    t = timeit(lambda: L.rcmarl_mid_fit(a1t.data_ptr(), theta.data_ptr(), y.data_ptr(), part.data_ptr(), S, N, B, in_dim, HID, ldp,
                                        ldb, st))
    print("mid_fit   %8.1f us  (%.2f TB/s on a1t r+w)" % (t, 8.0 * S * N * HID * B / t / 1e6))
    out = torch.zeros(S, N, ldb, device="cuda")
    t = timeit(lambda: L.rcmarl_mid_value(a1t.data_ptr(), theta.data_ptr(), None, 0.9, out.data_ptr(), S, N, B, in_dim, HID, ldp,
                                          ldb, st))
    print("mid_value %8.1f us  (%.2f TB/s on a1t r)" % (t, 4.0 * S * N * HID * B / t / 1e6))
    for d, H in ((4, 1), (18, 8)):
        nbr = torch.tensor([[(i + k) % N for k in range(d)] for i in range(N)], dtype=torch.int32, device="cuda")
        coop = torch.ones(N, dtype=torch.int32, device="cuda")
        t = timeit(lambda: L.rcmarl_consensus_head(a1t.data_ptr(), theta.data_ptr(), theta.data_ptr(), nbr.data_ptr(),
                                                   coop.data_ptr(), part.data_ptr(), None, S, N, B, in_dim, HID, ldp, ldb, d, H, st))
        print("cons_head d=%d H=%d %8.1f us" % (d, H, t))


if __name__ == "__main__":
    L = capi.load()
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    print("== %s  RCMARL_GEMM=%s RCMARL_K1=%s" % (what, os.environ.get("RCMARL_GEMM"), os.environ.get("RCMARL_K1")))
    {"gemm": gemm, "k1": k1, "mid": mid}[what](L)
