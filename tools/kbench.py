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


_FL = None


def _flags():
    """caller-owned out-of-range flag buffer of the f16 mid / mini-batch kernels (int32, zero at first use)"""
    global _FL
    if _FL is None:
        _FL = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
    return _FL.data_ptr()


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
    only = os.environ.get("RCMARL_KBENCH_ONLY")       # e.g. "18": just the d=18, H=8 critic case (for counter runs)
    for d, H in ((4, 1), (10, 4), (18, 8), (18, 1)):
        if only and (d != int(only) or 2 * H + 2 != d):
            continue
        for in_dim in ((2 * N,) if only else (2 * N, 3 * N)):
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
            if L.rcmarl_consensus_params_circulant_supported(N, d, H):
                t = timeit(lambda: L.rcmarl_consensus_params_circulant(msg.data_ptr(), theta.data_ptr(), coop.data_ptr(), S, N, ldp,
                                                                       P_hid, d, H, None, None, st), iters=20)
                print("   circulant kernel     %7.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (t, byts / t / 1e3, byts / t / 1e3 / 80))


def k1_cfg5(L):
    """BASELINE configs[4] phase II-b on ONE of 8 GPUs: 1024 agents, d=66, H=32, the 512-wide critic's hidden
    parameters column-sharded 8 ways (each output column needs only its own column of the d
    neighbour rows, so the shard needs no exchange inside the step)."""
    st = torch.cuda.current_stream().cuda_stream
    N, d, H = 1024, 66, 32
    P_hid_full = 2048 * 512 + 512 + 512 * 512 + 512
    P_hid = (P_hid_full + 7) // 8
    ldp = pad64(P_hid + 1)
    msg = torch.randn(1, N, ldp, device="cuda")
    theta = torch.zeros(1, N, ldp, device="cuda")
    nbr = torch.tensor([[(i + k) % N for k in range(d)] for i in range(N)], dtype=torch.int32, device="cuda")
    coop = torch.ones(N, dtype=torch.int32, device="cuda")
    t = timeit(lambda: L.rcmarl_consensus_params(msg.data_ptr(), theta.data_ptr(), nbr.data_ptr(), coop.data_ptr(), 1, N, ldp,
                                                 P_hid, d, H, None, None, st), iters=5, warm=2)
    byts = 8.0 * N * P_hid
    print("K1 cfg5 shard: N=%d d=%d H=%d P_hid/8=%d  %9.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)  -> %.1f consensus-updates/s/GPU-shard"
          % (N, d, H, P_hid, t, byts / t / 1e3, byts / t / 1e3 / 80, N / (t * 1e-6)))
    t = timeit(lambda: L.rcmarl_consensus_params_circulant(msg.data_ptr(), theta.data_ptr(), coop.data_ptr(), 1, N, ldp, P_hid, d,
                                                           H, None, None, st), iters=5, warm=2)
    print("   circulant kernel (G=%s): %9.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (os.environ.get("RCMARL_K1_G", "default"), t,
                                                                                  byts / t / 1e3, byts / t / 1e3 / 80))


def minibatch(L, S=512, N=5, B=3000):
    """adversary mini-batch fit (fit(batch_size=32, epochs=10)) alone: one launch = 940 sequential SGD steps per net"""
    st = torch.cuda.current_stream().cuda_stream
    for in_dim in (2 * N, 3 * N):
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        x = torch.randn(S, B, in_dim, device="cuda")
        theta = torch.randn(S, N, ldp, device="cuda") * 0.05
        y = torch.randn(S, N, ldb, device="cuda")
        agents = torch.tensor([N - 1], dtype=torch.int32, device="cuda")
        perm = torch.stack([torch.stack([torch.randperm(B, device="cuda") for _ in range(10)]) for _ in range(S)]).to(torch.int32).reshape(S, 1, 10, B).contiguous()
        t = timeit(lambda: L.rcmarl_minibatch_fit(x.data_ptr(), B * in_dim, theta.data_ptr(), agents.data_ptr(), 1, y.data_ptr(), perm.data_ptr(),
                                                  S, N, B, in_dim, HID, ldp, ldb, 32, 10, 1e-4, None, _flags(), st), iters=3, warm=1)
        print("minibatch_fit in=%3d S=%d: %9.1f us per launch, %.2f us per SGD step" % (in_dim, S, t, t / (10 * ((B + 31) // 32))))


def multi(L, S=None, N=5, B=3000):
    S = int(os.environ.get("KB_S", "512")) if S is None else S
    """the Malicious agent's three chains of a consensus epoch: three rcmarl_minibatch_fit launches (one after the other / on three
    streams, as the engine did until round 4) against ONE rcmarl_minibatch_fit_multi launch"""
    st = torch.cuda.current_stream().cuda_stream
    jobs = []
    for in_dim, n_adv in ((2 * N, 1), (3 * N, 1), (2 * N, 1)):
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        jobs.append(dict(in_dim=in_dim, ldp=ldp, x=torch.randn(S, B, in_dim, device="cuda"), theta=torch.randn(S, N, ldp, device="cuda") * 0.05,
                         y=torch.randn(S, N, ldb, device="cuda"), agents=torch.tensor([N - 1], dtype=torch.int32, device="cuda"),
                         perm=torch.stack([torch.stack([torch.randperm(B, device="cuda") for _ in range(10)]) for _ in range(S)]).to(torch.int32).reshape(S, 1, 10, B).contiguous(),
                         flags=torch.zeros(S + 1, dtype=torch.int32, device="cuda")))
    ldb = pad64(B)

    def single(j, stream):
        L.rcmarl_minibatch_fit(j["x"].data_ptr(), B * j["in_dim"], j["theta"].data_ptr(), j["agents"].data_ptr(), 1, j["y"].data_ptr(),
                               j["perm"].data_ptr(), S, N, B, j["in_dim"], HID, j["ldp"], ldb, 32, 10, 1e-4, None, j["flags"].data_ptr(), stream)

    def seq():
        for j in jobs:
            single(j, st)
    side = [torch.cuda.Stream() for _ in range(2)]

    def par():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        for j, s_ in zip(jobs[1:], side):
            s_.wait_event(ev)
            single(j, s_.cuda_stream)
        single(jobs[0], st)
        for s_ in side:
            e2 = torch.cuda.Event(); e2.record(s_); cur.wait_event(e2)
    arr = (capi.MbJob * 3)(*[capi.MbJob(j["x"].data_ptr(), B * j["in_dim"], j["theta"].data_ptr(), j["agents"].data_ptr(), 1, j["in_dim"], j["ldp"], 0,
                                       j["y"].data_ptr(), j["perm"].data_ptr(), None, j["flags"].data_ptr()) for j in jobs])

    def one():
        L.rcmarl_minibatch_fit_multi(arr, 3, S, N, B, HID, ldb, 32, 10, 1e-4, st)
    for name, fn in (("three launches, one stream", seq), ("three launches, three streams", par), ("ONE multi launch", one)):
        t = timeit(fn, iters=3, warm=1)
        print("%-32s %9.1f us  (%.2f us per SGD step of a chain)" % (name, t, t / 940))
    for c in ("0", "1"):
        os.environ["RCMARL_MB_MX_COMPACT"] = c
        t = timeit(one, iters=3, warm=1)
        print("ONE multi launch, compact=%s      %9.1f us" % (c, t))
    os.environ.pop("RCMARL_MB_MX_COMPACT")


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
    only = os.environ.get("RCMARL_KBENCH_ONLY")       # e.g. "18": just the estimate-consensus kernel at d = 18; "value": just mid_value (counter runs)
    if only == "value":
        out = torch.zeros(S, N, ldb, device="cuda")
        t = timeit(lambda: L.rcmarl_mid_value(a1t.data_ptr(), theta.data_ptr(), None, 0.9, out.data_ptr(), S, N, B, in_dim, HID, ldp, ldb, st))
        print("mid_value %8.1f us  (%.2f TB/s on a1t r)" % (t, 4.0 * S * N * HID * B / t / 1e6))
        return
    if not only:
        t = timeit(lambda: L.rcmarl_mid_fit(a1t.data_ptr(), theta.data_ptr(), y.data_ptr(), part.data_ptr(), S, N, B, in_dim, HID, ldp,
                                            ldb, st))
        print("mid_fit   %8.1f us  (%.2f TB/s on a1t r+w)" % (t, 8.0 * S * N * HID * B / t / 1e6))
        out = torch.zeros(S, N, ldb, device="cuda")
        for mx in ("1", "0"):        # layer 2 on the f16 matrix core (k_mid_value_mx) | on the vector ALUs (k_mid_value)
            os.environ["RCMARL_MIDVALUE_MX"] = mx
            t = timeit(lambda: L.rcmarl_mid_value(a1t.data_ptr(), theta.data_ptr(), None, 0.9, out.data_ptr(), S, N, B, in_dim, HID, ldp,
                                                  ldb, st))
            print("mid_value RCMARL_MIDVALUE_MX=%s %8.1f us  (%.2f TB/s on a1t r)" % (mx, t, 4.0 * S * N * HID * B / t / 1e6))
        os.environ.pop("RCMARL_MIDVALUE_MX", None)
    for d, H in ((4, 1), (10, 4), (18, 8)):
        if only and d != int(only):
            continue
        nbr = torch.tensor([[(i + k) % N for k in range(d)] for i in range(N)], dtype=torch.int32, device="cuda")
        coop = torch.ones(N, dtype=torch.int32, device="cuda")
        for mx in ("1", "0"):        # layer 2 + heads on the f16 matrix core (k_consensus_head_mx) | on the vector ALUs (k_consensus_head)
            os.environ["RCMARL_K2_MX"] = mx
            t = timeit(lambda: L.rcmarl_consensus_head(a1t.data_ptr(), theta.data_ptr(), theta.data_ptr(), nbr.data_ptr(),
                                                       coop.data_ptr(), part.data_ptr(), None, S, N, B, in_dim, HID, ldp, ldb, d, H, st))
            byts = 4.0 * S * N * B * (HID + 2)
            print("cons_head d=%d H=%d RCMARL_K2_MX=%s %8.1f us  (%.2f TB/s of 4 B (h + 2) per (agent, row) = %.3f of 8 TB/s)"
                  % (d, H, mx, t, byts / t / 1e6, byts / t / 1e6 / 8.0))
        os.environ.pop("RCMARL_K2_MX", None)


def lattice(L, S=16, N=256, B=3000):
    """lattice (exact bf16x3) layer-1 GEMMs + their producers, on random lattice inputs."""
    from rcmarl_amd import lattice as LT
    st = torch.cuda.current_stream().cuda_stream
    for width in (2, 3):
        in_dim = width * N
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        g = LT.Geometry(N, in_dim, B)
        pos = torch.randint(0, 32, (S, B, in_dim), device="cuda").float()
        std = float(np.std(np.arange(32)))
        x = ((pos - 15.5) / std).contiguous()
        alpha = torch.full((in_dim,), 0.5 / std, device="cuda")
        theta = torch.randn(S, N, ldp, device="cuda") * 0.05
        a1t = torch.zeros(S, N * HID, ldb, device="cuda")
        y = torch.randn(S, N, ldb, device="cuda")
        mask = torch.ones(N, dtype=torch.int32, device="cuda")
        u8 = lambda rk, pc: torch.zeros(S * LT.Geometry.nbytes(rk, pc), dtype=torch.uint8, device="cuda")
        kp, ktp, wp, dzp = u8(g.kp, 1), u8(g.ktp, 1), u8(g.wp, 3), u8(g.dzp, 3)
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        nchunk = (B + 255) // 256
        part = torch.zeros(S * N * nchunk * L.rcmarl_fit_partial_size(HID), device="cuda")
        flops = 2.0 * S * N * HID * B * in_dim
        t = timeit(lambda: L.rcmarl_lattice_encode(x.data_ptr(), B * in_dim, alpha.data_ptr(), S, B, in_dim, kp.data_ptr(), g.kp[0],
                                                   g.kp[1], ktp.data_ptr(), g.ktp[0], g.ktp[1], flag.data_ptr(), st))
        print("encode    in=%4d  %8.1f us   flag=%d" % (in_dim, t, int(flag.item())))
        t = timeit(lambda: L.rcmarl_w1_split(theta.data_ptr(), alpha.data_ptr(), wp.data_ptr(), S, N, in_dim, HID, ldp, g.wp[0],
                                             g.wp[1], st))
        print("w1_split  in=%4d  %8.1f us  (%.2f TB/s r+w)" % (in_dim, t, 10.0 * S * N * HID * in_dim / t / 1e6))
        t = timeit(lambda: L.rcmarl_layer1_forward_lattice(kp.data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0], g.wp[1],
                                                           theta.data_ptr(), a1t.data_ptr(), S, N, B, in_dim, HID, ldp, ldb, st))
        print("fwd_lat   in=%4d  %8.1f us  %6.1f TF/s fp32-equivalent (%.0f TF/s bf16 executed)" % (in_dim, t, flops / t / 1e6,
                                                                                                   3 * flops / t / 1e6))
        t = timeit(lambda: L.rcmarl_mid_fit_lattice(a1t.data_ptr(), theta.data_ptr(), y.data_ptr(), part.data_ptr(), dzp.data_ptr(),
                                                    g.dzp[0], g.dzp[1], S, N, B, in_dim, HID, ldp, ldb, _flags(), st))
        print("mid_fit_l in=%4d  %8.1f us" % (in_dim, t))
        t = timeit(lambda: L.rcmarl_layer1_backward_sgd_lattice(ktp.data_ptr(), g.ktp[0], g.ktp[1], dzp.data_ptr(), g.dzp[0],
                                                                g.dzp[1], alpha.data_ptr(), theta.data_ptr(), mask.data_ptr(), S, N,
                                                                B, in_dim, HID, ldp, 1e-6, wp.data_ptr(), g.wp[0], g.wp[1], st))
        print("bwd_lat   in=%4d  %8.1f us  %6.1f TF/s fp32-equivalent (%.0f TF/s bf16 executed)" % (in_dim, t, flops / t / 1e6,
                                                                                                   3 * flops / t / 1e6))


def mid_ab(L, S=16, N=256, B=3000):
    """A/B of rcmarl_mid_fit_lattice: the product library (default kernel = v5, and RCMARL_MIDFIT=7 = the bf16 matrix-core form)
    against the variant builds named in RCMARL_KBENCH_LIB_B (comma-separated paths; tools/build_variant.py), interleaved."""
    from rcmarl_amd import lattice as LT
    libs = [("product", L, None), ("product-v5", L, "5")]      # default = k_mid_fit_v8 + fix-up launch; RCMARL_MIDFIT=5 = v5 alone
    for pth in [x for x in os.environ.get("RCMARL_KBENCH_LIB_B", "").split(",") if x]:
        libs.append((os.path.basename(pth).replace("lib", "").replace(".so", ""), capi.CLib(pth), None))
    st = torch.cuda.current_stream().cuda_stream
    for in_dim in (2 * N, 3 * N):
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        g = LT.Geometry(N, in_dim, B)
        theta = torch.randn(S, N, ldp, device="cuda") * 0.05
        a1t = torch.randn(S, N * HID, ldb, device="cuda")
        y = torch.randn(S, N, ldb, device="cuda")
        dzp = torch.zeros(S * LT.Geometry.nbytes(g.dzp, 3), dtype=torch.uint8, device="cuda")
        part = torch.zeros(S * N * ((B + 255) // 256) * L.rcmarl_fit_partial_size(HID), device="cuda")
        outs = {}
        for rnd in range(3):
            for name, lib, midfit in libs:
                if midfit is None:
                    os.environ.pop("RCMARL_MIDFIT", None)
                else:
                    os.environ["RCMARL_MIDFIT"] = midfit
                dzp.zero_()
                t = timeit(lambda: lib.rcmarl_mid_fit_lattice(a1t.data_ptr(), theta.data_ptr(), y.data_ptr(), part.data_ptr(), dzp.data_ptr(),
                                                              g.dzp[0], g.dzp[1], S, N, B, in_dim, HID, ldp, ldb, _flags(), st), iters=20)
                outs[name] = (dzp.clone(), part.clone())
                print("in=%d round %d  %-16s %8.1f us  (%.0f GB/s of its 160 B per row and agent)" % (in_dim, rnd, name, t, 160.0 * S * N * B / t / 1e3))
        os.environ.pop("RCMARL_MIDFIT", None)
        ref = outs["product"]
        for name, (dz, pt) in outs.items():
            rel = float(((pt - ref[1]).abs().max() / ref[1].abs().max()).item())
            print("   %-16s dz image identical to product: %s   records max rel diff %.2e" % (name, bool(torch.equal(dz, ref[0])), rel))


def lattice_ab(L, S=16, N=256, B=3000):
    """A/B of the two lattice GEMMs: the product library against the variant builds named in RCMARL_KBENCH_LIB_B (comma-separated
    paths; tools/build_variant.py), interleaved, on operands produced once by the product library."""
    from rcmarl_amd import lattice as LT
    libs = [("product", L)]
    for pth in [x for x in os.environ.get("RCMARL_KBENCH_LIB_B", "").split(",") if x]:
        libs.append((os.path.basename(pth).replace("lib", "").replace(".so", ""), capi.CLib(pth)))
    st = torch.cuda.current_stream().cuda_stream
    for width in (2, 3):
        in_dim = width * N
        P = in_dim * HID + HID + HID * HID + HID + HID + 1
        ldp, ldb = pad64(P), pad64(B)
        g = LT.Geometry(N, in_dim, B)
        std = float(np.std(np.arange(32)))
        x = ((torch.randint(0, 32, (S, B, in_dim), device="cuda").float() - 15.5) / std).contiguous()
        alpha = torch.full((in_dim,), 0.5 / std, device="cuda")
        theta = torch.randn(S, N, ldp, device="cuda") * 0.05
        a1t = torch.zeros(S, N * HID, ldb, device="cuda")
        y = torch.randn(S, N, ldb, device="cuda")
        mask = torch.ones(N, dtype=torch.int32, device="cuda")
        u8 = lambda rk, pc: torch.zeros(S * LT.Geometry.nbytes(rk, pc), dtype=torch.uint8, device="cuda")
        kp, ktp, wp, dzp = u8(g.kp, 1), u8(g.ktp, 1), u8(g.wp, 3), u8(g.dzp, 3)
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        part = torch.zeros(S * N * ((B + 255) // 256) * L.rcmarl_fit_partial_size(HID), device="cuda")
        L.rcmarl_lattice_encode(x.data_ptr(), B * in_dim, alpha.data_ptr(), S, B, in_dim, kp.data_ptr(), g.kp[0], g.kp[1], ktp.data_ptr(),
                                g.ktp[0], g.ktp[1], flag.data_ptr(), st)
        L.rcmarl_w1_split(theta.data_ptr(), alpha.data_ptr(), wp.data_ptr(), S, N, in_dim, HID, ldp, g.wp[0], g.wp[1], st)
        L.rcmarl_layer1_forward_lattice(kp.data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0], g.wp[1], theta.data_ptr(), a1t.data_ptr(),
                                        S, N, B, in_dim, HID, ldp, ldb, st)
        L.rcmarl_mid_fit_lattice(a1t.data_ptr(), theta.data_ptr(), y.data_ptr(), part.data_ptr(), dzp.data_ptr(), g.dzp[0], g.dzp[1], S, N, B,
                                 in_dim, HID, ldp, ldb, _flags(), st)
        ref = None
        for rnd in range(3):
            for name, lib in libs:
                tf = timeit(lambda: lib.rcmarl_layer1_forward_lattice(kp.data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0], g.wp[1],
                                                                      theta.data_ptr(), a1t.data_ptr(), S, N, B, in_dim, HID, ldp, ldb, st), iters=20)
                same = True if ref is None else bool(torch.equal(a1t, ref))
                if ref is None:
                    ref = a1t.clone()
                tb = timeit(lambda: lib.rcmarl_layer1_backward_sgd_lattice(ktp.data_ptr(), g.ktp[0], g.ktp[1], dzp.data_ptr(), g.dzp[0], g.dzp[1],
                                                                           alpha.data_ptr(), theta.data_ptr(), mask.data_ptr(), S, N, B, in_dim,
                                                                           HID, ldp, 0.0, wp.data_ptr(), g.wp[0], g.wp[1], st), iters=20)
                print("in=%d round %d  %-16s fwd %8.1f us   bwd %8.1f us   a1 identical: %s" % (in_dim, rnd, name, tf, tb, same))


def wide(L, S=1, N=64, B=3000, in_dim=2048, hid=512):
    """the dense-GEMM path at the cfg-5 shape (1024 agents x 512-wide critic, here a 64-agent slice)"""
    st = torch.cuda.current_stream().cuda_stream
    o_b1 = in_dim * hid
    o_W2 = o_b1 + hid
    o_b2 = o_W2 + hid * hid
    P = o_b2 + hid + hid + 1
    ldp, ldb = pad64(P), pad64(B)
    x = torch.randn(S, B, in_dim, device="cuda")
    theta = torch.randn(S, N, ldp, device="cuda") * 0.02
    a1, a2, dz1 = (torch.zeros(S, N * hid, ldb, device="cuda") for _ in range(3))
    y = torch.randn(S, N, ldb, device="cuda")
    dz3 = torch.zeros(S, N, ldb, device="cuda")
    grads = torch.zeros(S, N, L.rcmarl_wide_grad_size(hid), device="cuda")
    lp = torch.zeros(S, N, (B + 255) // 256, device="cuda")
    mask = torch.ones(N, dtype=torch.int32, device="cuda")
    f1, f2 = 2.0 * S * N * hid * B * in_dim, 2.0 * S * N * hid * B * hid
    rows = [
        ("fwd L1 (W1^T x)", f1, lambda: L.rcmarl_dense_forward(x.data_ptr(), B * in_dim, 0, 1, in_dim, theta.data_ptr(), 0, o_b1,
                                                               a1.data_ptr(), S, N, B, in_dim, hid, ldp, ldb, st)),
        ("fwd L2 (W2^T a1)", f2, lambda: L.rcmarl_dense_forward(a1.data_ptr(), N * hid * ldb, hid * ldb, 0, ldb, theta.data_ptr(),
                                                                o_W2, o_b2, a2.data_ptr(), S, N, B, hid, hid, ldp, ldb, st)),
        ("head fit", 0, lambda: L.rcmarl_wide_head_fit(a2.data_ptr(), theta.data_ptr(), y.data_ptr(), dz3.data_ptr(),
                                                       grads.data_ptr(), lp.data_ptr(), S, N, B, in_dim, hid, ldp, ldb, st)),
        ("bwd data L2 (W2 dz2)", f2, lambda: L.rcmarl_dense_backward_data(a2.data_ptr(), theta.data_ptr(), o_W2, a1.data_ptr(),
                                                                          dz1.data_ptr(), S, N, B, hid, hid, ldp, ldb, st)),
        ("bias grad", 0, lambda: L.rcmarl_wide_bias_grad(dz1.data_ptr(), grads.data_ptr(), S, N, B, hid, ldb, st)),
        ("bwd W2 (a1 dz2^T)", f2, lambda: L.rcmarl_dense_backward_sgd(a1.data_ptr(), N * hid * ldb, hid * ldb, 0, ldb, a2.data_ptr(),
                                                                       theta.data_ptr(), o_W2, mask.data_ptr(), S, N, B, hid, hid,
                                                                       ldp, ldb, 1e-9, st)),
        ("bwd W1 (x^T dz1^T)", f1, lambda: L.rcmarl_dense_backward_sgd(x.data_ptr(), B * in_dim, 0, 1, in_dim, dz1.data_ptr(),
                                                                        theta.data_ptr(), 0, mask.data_ptr(), S, N, B, in_dim, hid,
                                                                        ldp, ldb, 1e-9, st)),
    ]
    for form in (1, 0):                                  # 16-bit matrix core (two f16 pieces per operand, three passes) | fp32-input MFMA
        L.rcmarl_wide_set_f16_mode(form)
        print("-- RCMARL_WIDE_F16=%d" % form)
        tot = 0.0
        for name, fl, fn in rows:
            t = timeit(fn, iters=5, warm=2)
            tot += t
            print("%-22s %9.1f us  %s" % (name, t, ("%6.1f TF/s fp32-equivalent (%.0f%% of the 157 TF/s f32 MFMA peak)" % (fl / t / 1e6, fl / t / 1e6 / 1.573))
                                             if fl else ""))
        print("one SGD step, %d agents: %.2f ms  -> 1024 agents: %.1f ms" % (N, tot / 1e3, tot / 1e3 * 1024 / N))
    L.rcmarl_wide_set_f16_mode(-1)


def pk(L, S=1, N=int(os.environ.get("RCMARL_KBENCH_N", "256")), B=3000, width=2, hid=512):
    """the wide critic's SGD step on pre-split packed operands (csrc/dense_pk.hip) at the cfg-5 shape (1024 agents x 512 units;
    here RCMARL_KBENCH_N agents of the same 2048-input network), kernel by kernel, beside the round-4 path (`wide`)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import wide_checks as WC
    from kernel_checks import lattice_rows
    from rcmarl_amd import lattice as LT

    class Bk:
        lib = L
        stream = torch.cuda.current_stream().cuda_stream
        dev = staticmethod(lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda())
        ptr = staticmethod(lambda h: None if h is None else h.data_ptr())
    bk = Bk()
    NA = 1024                                          # the network's input width is that of the 1024-agent instance
    in_dim = NA * width
    g = WC.wgeom(in_dim, hid)
    ldp, ldb = pad64(g["P"]), pad64(B)
    rng = np.random.default_rng(0)
    x, alpha = lattice_rows(rng, S, B, NA, width, 32, 32)
    theta = (torch.randn(S, N, ldp, device="cuda") * 0.02)
    y = torch.randn(S, N, ldb, device="cuda")
    mask = torch.ones(N, dtype=torch.int32, device="cuda")
    loss = torch.zeros(S, N, device="cuda")
    pb = WC.PkBuffers(bk, S, N, B, in_dim, hid)
    d_x, d_al = bk.dev(x), bk.dev(alpha)
    WC.pk_encode(bk, pb, d_x, B * in_dim, d_al, S, B, in_dim)
    gg, st = pb.g, bk.stream
    p = lambda t: t.data_ptr()
    f1, f2 = 2.0 * S * N * hid * B * in_dim, 2.0 * S * N * hid * B * hid
    lr = 1e-9
    rows = [
        ("W1 split", 0, lambda: L.rcmarl_w1_split(p(theta), p(d_al), p(pb.wp), S, N, in_dim, hid, ldp, gg.wp[0], gg.wp[1], st)),
        ("fwd L1 lattice -> packed a1 x2 + signs", f1, lambda: L.rcmarl_layer1_forward_lattice_pk(
            p(pb.kp), gg.kp[0], gg.kp[1], p(pb.wp), gg.wp[0], gg.wp[1], p(theta), p(pb.a1_bk), pb.bk_rt, p(pb.a1_kb), pb.kb_kt, p(pb.s1),
            pb.Bp // 32, None, S, N, B, in_dim, hid, ldp, st)),
        ("fwd L1 lattice -> packed a1 (values)", f1, lambda: L.rcmarl_layer1_forward_lattice_pk(
            p(pb.kp), gg.kp[0], gg.kp[1], p(pb.wp), gg.wp[0], gg.wp[1], p(theta), p(pb.a1_bk), pb.bk_rt, None, pb.kb_kt, None,
            pb.Bp // 32, None, S, N, B, in_dim, hid, ldp, st)),
        ("fwd L1 lattice -> a1_bk + a1_kb", f1, lambda: L.rcmarl_layer1_forward_lattice_pk(
            p(pb.kp), gg.kp[0], gg.kp[1], p(pb.wp), gg.wp[0], gg.wp[1], p(theta), p(pb.a1_bk), pb.bk_rt, p(pb.a1_kb), pb.kb_kt, None,
            pb.Bp // 32, None, S, N, B, in_dim, hid, ldp, st)),
        ("fwd L1 lattice -> a1_bk + signs", f1, lambda: L.rcmarl_layer1_forward_lattice_pk(
            p(pb.kp), gg.kp[0], gg.kp[1], p(pb.wp), gg.wp[0], gg.wp[1], p(theta), p(pb.a1_bk), pb.bk_rt, None, pb.kb_kt, p(pb.s1),
            pb.Bp // 32, None, S, N, B, in_dim, hid, ldp, st)),
        ("pack W2", 0, lambda: L.rcmarl_pk_pack_w2(p(theta), p(pb.w2t), p(pb.w2w3), p(pb.rs), None, S, N, in_dim, hid, ldp, st)),
        ("fwd L2 -> masks + value parts", f2, lambda: L.rcmarl_pk_forward2(p(pb.w2t), p(pb.a1_bk), pb.bk_rt, p(theta), None, p(pb.mask_bj),
                                                                            pb.bk_rt, p(pb.mask_jb), pb.kb_kt, p(pb.vpart), None, S, N, B,
                                                                            in_dim, hid, ldp, ldb, st)),
        ("fwd L2 -> mask_bj + value parts", f2, lambda: L.rcmarl_pk_forward2(p(pb.w2t), p(pb.a1_bk), pb.bk_rt, p(theta), None, p(pb.mask_bj),
                                                                              pb.bk_rt, None, pb.kb_kt, p(pb.vpart), None, S, N, B,
                                                                              in_dim, hid, ldp, ldb, st)),
        ("fwd L2 -> mask_jb + value parts", f2, lambda: L.rcmarl_pk_forward2(p(pb.w2t), p(pb.a1_bk), pb.bk_rt, p(theta), None, None,
                                                                              pb.bk_rt, p(pb.mask_jb), pb.kb_kt, p(pb.vpart), None, S, N, B,
                                                                              in_dim, hid, ldp, ldb, st)),
        ("fwd L2 -> value parts", f2, lambda: L.rcmarl_pk_forward2(p(pb.w2t), p(pb.a1_bk), pb.bk_rt, p(theta), None, None, pb.bk_rt, None,
                                                                    pb.kb_kt, p(pb.vpart), None, S, N, B, in_dim, hid, ldp, ldb, st)),
        ("fwd L2 -> fp32 a2 (consensus)", f2, lambda: L.rcmarl_pk_forward2(p(pb.w2t), p(pb.a1_bk), pb.bk_rt, p(theta), p(pb.a2), None,
                                                                            pb.bk_rt, None, pb.kb_kt, p(pb.vpart), p(pb.npart), S, N, B, in_dim, hid,
                                                                            ldp, ldb, st)),
        ("head", 0, lambda: L.rcmarl_pk_head(p(pb.vpart), p(theta), p(y), 0.0, 2, p(pb.dz3), p(pb.dzv), p(pb.losspart), S, N, B, in_dim,
                                             hid, ldp, ldb, st)),
        ("bwd data L2 -> packed dz1", f2, lambda: L.rcmarl_pk_backward_data(p(pb.mask_bj), pb.bk_rt, p(pb.w2w3), p(pb.rs), p(pb.s1),
                                                                             pb.Bp // 32, p(pb.dz3), p(pb.dzp), gg.dzp[0], gg.dzp[1],
                                                                             p(pb.gb1part), None, S, N, B, hid, ldb, st)),
        ("bwd W2", f2, lambda: L.rcmarl_pk_backward_w2(p(pb.a1_kb), pb.kb_kt, p(pb.mask_jb), pb.kb_kt, p(pb.dzv), p(theta), p(mask),
                                                        p(pb.gw3part), p(pb.q), S, N, B, in_dim, hid, ldp, lr, st)),
        ("bwd W1 lattice", f1, lambda: L.rcmarl_layer1_backward_sgd_lattice(p(pb.ktp), gg.ktp[0], gg.ktp[1], p(pb.dzp), gg.dzp[0],
                                                                             gg.dzp[1], p(d_al), p(theta), p(mask), S, N, B, in_dim, hid,
                                                                             ldp, lr, p(pb.wp), gg.wp[0], gg.wp[1], st)),
        ("small sgd", 0, lambda: L.rcmarl_pk_small_sgd(p(pb.gw3part), p(pb.q), p(pb.gb1part), p(pb.dz3), p(pb.losspart), p(theta), p(mask),
                                                       p(loss), S, N, B, in_dim, hid, ldp, ldb, lr, st)),
    ]
    only = os.environ.get("RCMARL_KBENCH_ONLY")
    step_names = ("fwd L1 lattice -> packed a1 x2 + signs", "pack W2", "fwd L2 -> masks + value parts", "head", "bwd data L2 -> packed dz1",
                  "bwd W2", "bwd W1 lattice", "small sgd")
    if os.environ.get("RCMARL_KBENCH_STEP_ONLY"):          # counter runs: only the launches of a real fit step (no variant rows)
        rows = [r for r in rows if r[0] in step_names or r[0] == "W1 split"]
    tot = 0.0
    for name, fl, fn in rows:
        if only and only not in name:
            fn()                                       # (later kernels still need their inputs)
            continue
        t = timeit(fn, iters=5, warm=2)
        step = name in step_names
        tot += t if step else 0.0
        print("%-40s %9.1f us  %s" % (name, t, ("%7.1f TF/s fp32-equivalent (%.3f of 2.5 PF)" % (fl / t / 1e6, fl / t / 1e6 / 2500)) if fl else ""))
    print("one SGD step, %d agents: %.2f ms  -> 1024 agents: %.1f ms" % (N, tot / 1e3, tot / 1e3 * 1024 / N))


if __name__ == "__main__":
    L = capi.CLib(os.environ["RCMARL_KBENCH_LIB"]) if os.environ.get("RCMARL_KBENCH_LIB") else capi.load()     # (variant builds)
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    print("== %s  RCMARL_GEMM=%s RCMARL_K1=%s" % (what, os.environ.get("RCMARL_GEMM"), os.environ.get("RCMARL_K1")))
    {"gemm": gemm, "k1": k1, "k1_cfg5": k1_cfg5, "mid": mid, "lattice": lattice, "lattice_ab": lattice_ab, "mid_ab": mid_ab, "minibatch": minibatch, "multi": multi, "wide": wide, "pk": pk}[what](L)
