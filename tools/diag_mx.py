"""Diagnostic (GPU): growth of |theta_mx - theta_fp32| with the number of mini-batch SGD steps, same start, same data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rcmarl_amd import capi
L = capi.load()
HID = 20
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for in_dim in (10, 15):
    S, N = 4, 5
    P = in_dim * HID + HID + HID * HID + HID + HID + 1
    ldp = (P + 63) // 64 * 64
    for B in (32, 320, 3000, 3200):
        ldb = (B + 63) // 64 * 64
        x = torch.randn(S, B, in_dim, device="cuda")
        theta0 = torch.zeros(S, N, ldp, device="cuda"); theta0[:, :, :P] = torch.randn(S, N, P, device="cuda") * 0.3
        y = torch.randn(S, N, ldb, device="cuda")
        advs = [1, 3]
        agents = torch.tensor(advs, dtype=torch.int32, device="cuda")
        for epochs in (1, 3):
            perm = torch.stack([torch.stack([torch.randperm(B, device="cuda") for _ in range(epochs)]) for _ in range(S * len(advs))]).to(torch.int32).reshape(S, len(advs), epochs, B).contiguous()
            out = {}
            for mode in ("1", "0"):
                os.environ["RCMARL_MB_MX"] = mode
                th = theta0.clone()
                fl = torch.zeros(S * len(advs), dtype=torch.int32, device="cuda")
                L.rcmarl_minibatch_fit(x.data_ptr(), B * in_dim, th.data_ptr(), agents.data_ptr(), len(advs), y.data_ptr(), perm.data_ptr(),
                                       S, N, B, in_dim, HID, ldp, ldb, 32, epochs, 0.01, None, fl.data_ptr(), st)
                torch.cuda.synchronize()
                out[mode] = th[:, advs, :P].double().cpu().numpy()
            d = np.abs(out["1"] - out["0"]).max()
            print("in=%2d steps=%4d  max|mx - fp32| = %.3e   (|theta| max %.2f, changed by %.3e)" % (
                in_dim, epochs * ((B + 31) // 32), d, np.abs(out["0"]).max(), np.abs(out["0"] - theta0[:, advs, :P].double().cpu().numpy()).max()), flush=True)
