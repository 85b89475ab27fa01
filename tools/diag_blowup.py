"""Diagnostic: the cfg4_shard bench engine, first block, TR local fits step by step: where does a blown-up agent turn non-finite?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
from rcmarl_amd import capi
w = bench.WORKLOADS["cfg4_shard"]
S = w["S"]
lib = capi.load()
eng = bench.make_engine(w, S, list(range(1000, 1000 + S)), lib)
L = eng.lib
real_small_sgd = L.rcmarl_small_sgd
state = {"step": 0}
class Wrap:
    def __init__(self, lib): self.__dict__["lib"] = lib
    def __getattr__(self, k):
        f = getattr(self.lib, k)
        if k not in ("rcmarl_mid_fit_lattice", "rcmarl_layer1_backward_sgd_lattice", "rcmarl_layer1_forward_lattice"):
            return f
        def g(*a):
            r = f(*a)
            torch.cuda.synchronize()
            if eng._cur_net == "tr":
                msg = eng.msg["tr"]
                bad = ~torch.isfinite(msg)
                a1 = eng.a1net["tr"]
                part = eng.partials
                fin = torch.where(bad, torch.zeros_like(msg), msg).abs()
                per_agent = fin.reshape(S * eng.N, -1).max(1).values
                top = int(per_agent.argmax())
                print("%-38s step %3d | msg bad %d absmax %.3e (seed %d agent %d) | a1 bad %d absmax %.3e | partials bad %d" % (
                    k, state["step"], int(bad.sum()), float(per_agent.max()), top // eng.N, top % eng.N,
                    int((~torch.isfinite(a1)).sum()), float(torch.nan_to_num(a1, 0, 0, 0).abs().max()),
                    int((~torch.isfinite(part)).sum())), flush=True)
            state["step"] += 1
            return r
        return g
eng.lib = Wrap(L)
orig = eng._local_fit
def lf(net, *a, **k):
    eng._cur_net = net
    return orig(net, *a, **k)
eng._local_fit = lf
eng._cur_net = None
eng.rollout_block(eng.cfg.n_ep_fixed)
eng.cfg.n_epochs = int(os.environ.get("EPOCHS", "3"))
eng.update_block()
