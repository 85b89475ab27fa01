#!/usr/bin/env python
"""How far the f16 matrix-core mini-batch fit (k_minibatch_mx, default) drifts from the fp32 kernel (RCMARL_MB_MX=0: the oracle's fmaf
chains) over the adversaries' real chain length -- fit(batch_size=32, epochs=10) at B = 3000 = 940 dependent SGD steps
(agents/adversarial_CAC_agents.py:121-165) -- on MANY networks: per-network max |theta_mx - theta_fp32| / max(1, |theta|), as a
histogram by decade and as quantiles.  The bar of tests/kernel_checks.py::check_minibatch_fit is set from this file's output
(profiles/r04*_knife_edge_hist.txt).

    python tools/knife_edge_hist.py [n_seeds=512]        # x 5 agents x 2 input widths = 5120 networks by default
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rcmarl_amd import capi  # noqa: E402

L = capi.load()
HID = 20
st = torch.cuda.current_stream().cuda_stream
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, B, epochs, lr = 5, 3000, 10, 0.01
torch.manual_seed(4)
allrel, allctl = [], []
for in_dim, what in ((10, "critic-shaped (10 inputs)"), (15, "team-reward-shaped (15 inputs)")):
    P = in_dim * HID + HID + HID * HID + HID + HID + 1
    ldp, ldb = (P + 63) // 64 * 64, (B + 63) // 64 * 64
    # inputs as the environment makes them: z-scored grid coordinates on a 5x5 grid (+ raw action indices), targets O(1)
    std = float(np.std(np.arange(5)))
    pos = torch.randint(0, 5, (S, B, 2 * N), device="cuda").float()
    x = (pos - 2.0) / std
    if in_dim == 15:
        act = torch.randint(0, 5, (S, B, N), device="cuda").float()
        x = torch.cat([x.reshape(S, B, N, 2), act.reshape(S, B, N, 1)], dim=-1).reshape(S, B, 3 * N)
    x = x.contiguous()
    lim1, lim2 = float(np.sqrt(6.0 / (in_dim + HID))), float(np.sqrt(6.0 / (HID + HID)))
    theta0 = torch.zeros(S, N, ldp, device="cuda")
    theta0[:, :, :in_dim * HID] = (torch.rand(S, N, in_dim * HID, device="cuda") * 2 - 1) * lim1          # Glorot-uniform, as Keras
    o = in_dim * HID + HID
    theta0[:, :, o:o + HID * HID] = (torch.rand(S, N, HID * HID, device="cuda") * 2 - 1) * lim2
    o += HID * HID + HID
    theta0[:, :, o:o + HID] = (torch.rand(S, N, HID, device="cuda") * 2 - 1) * float(np.sqrt(6.0 / (HID + 1)))
    y = torch.zeros(S, N, ldb, device="cuda")
    y[:, :, :B] = -torch.randint(0, 9, (S, N, B), device="cuda").float() / 5 + 0.9 * torch.randn(S, N, B, device="cuda")
    agents = torch.arange(N, dtype=torch.int32, device="cuda")
    perm = torch.rand(S, N, epochs, B, device="cuda").argsort(-1).to(torch.int32).contiguous()
    out = {}
    # "0u": the CONTROL -- the fp32 kernel again, started from weights that differ from theta0 by ONE ulp in ONE weight per network
    # (W3[0]): how far two fp32 runs of this chain drift apart on their own
    theta0u = theta0.clone()
    o3 = in_dim * HID + HID + HID * HID + HID
    theta0u[:, :, o3] = torch.nextafter(theta0[:, :, o3], torch.full_like(theta0[:, :, o3], 10.0))
    for mode in ("1", "0", "0u"):
        os.environ["RCMARL_MB_MX"] = mode[0]
        th = (theta0u if mode == "0u" else theta0).clone()
        fl = torch.zeros(S * N, dtype=torch.int32, device="cuda")
        L.rcmarl_minibatch_fit(x.data_ptr(), B * in_dim, th.data_ptr(), agents.data_ptr(), N, y.data_ptr(), perm.data_ptr(), S, N, B,
                               in_dim, HID, ldp, ldb, 32, epochs, lr, None, fl.data_ptr(), st)
        torch.cuda.synchronize()
        out[mode] = th[:, :, :P].double().cpu().numpy().reshape(S * N, P)
    os.environ.pop("RCMARL_MB_MX", None)
    scale = np.maximum(1.0, np.abs(out["0"]).max(axis=1))
    rel = np.abs(out["1"] - out["0"]).max(axis=1) / scale
    allrel.append(rel)
    fin = np.isfinite(out["0"]).all() and np.isfinite(out["1"]).all()
    print("%s: %d networks x %d SGD steps, lr %.3g; finite %s; moved by up to %.2f" %
          (what, S * N, epochs * ((B + 31) // 32), lr, fin, float(np.abs(out["0"] - theta0[:, :, :P].double().cpu().numpy().reshape(S * N, P)).max())))
    relc = np.abs(out["0u"] - out["0"]).max(axis=1) / scale
    allctl.append(relc)
    edges = [0, 1e-7, 1e-6, 1e-5, 2e-5, 1e-4, 1e-3, 1e-2, 1e-1, np.inf]
    h, _ = np.histogram(rel, bins=edges)
    hc, _ = np.histogram(relc, bins=edges)
    print("   max rel deviation per network            f16 matrix core vs fp32      fp32 vs fp32 started one ulp away (control)")
    for lo, hi, c, cc in zip(edges[:-1], edges[1:], h, hc):
        print("   %8.0e <= .. < %8.0e :   %5d networks (%5.2f %%)            %5d networks (%5.2f %%)" %
              (lo, hi, c, 100.0 * c / rel.size, cc, 100.0 * cc / rel.size))
    for nm, v in (("f16 vs fp32", rel), ("control   ", relc)):
        print("   %s quantiles: median %.2e, 90 %% %.2e, 99 %% %.2e, 99.9 %% %.2e, max %.2e" %
              ((nm,) + tuple(np.quantile(v, q) for q in (0.5, 0.9, 0.99, 0.999, 1.0))))
for nm, v in (("f16 matrix core vs fp32", np.concatenate(allrel)), ("fp32 vs fp32 one ulp away", np.concatenate(allctl))):
    print("all %d networks, %s: fraction within 2e-5: %.4f, within 1e-4: %.4f, within 1e-3: %.4f, within 1e-2: %.4f; max %.2e" %
          (v.size, nm, (v <= 2e-5).mean(), (v <= 1e-4).mean(), (v <= 1e-3).mean(), (v <= 1e-2).mean(), v.max()))
