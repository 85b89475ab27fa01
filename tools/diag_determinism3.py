"""Diagnostic: the same engine program run twice IN SEQUENCE (second engine reuses the first one's freed memory);
reports the first snapshot that differs."""
import os, sys, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcmarl_amd.engine import EngineConfig, RPBCACEngine
N = 5; S = int(os.environ.get("S", "256")); EPOCHS = int(os.environ.get("EPOCHS", "10"))


def run():
    cfg = EngineConfig(N, ["Cooperative"] * N, [[(i + k) % N for k in range(4)] for i in range(N)], H=1, n_seeds=S, rng_mode="device")
    e = RPBCACEngine(cfg, seeds=list(range(100, 100 + S)))
    e.init_glorot(base_seed=1)
    e.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(N, 2)) for s in range(S)]))
    snaps = []
    snap = lambda name, t: snaps.append((name, t.detach().cpu().clone()))
    e.rollout_block(e.cfg.n_ep_fixed)
    B = e.B
    L = e.lib
    e._lattice_encode(B)
    rptr, rstride = e._x("r")
    L.rcmarl_team_reward(rptr, rstride, e.coop.data_ptr(), max(e.n_coop, 1), e.rcoop.data_ptr(), e.S, e.N, B, e.ldb, e.stream)
    L.rcmarl_gather_agent_major(rptr, rstride, e.rcoop.data_ptr(), e.fit_mode.data_ptr(), e.ybuf["r_fit"].data_ptr(), e.S, e.N, B, e.ldb, e.stream)
    snap("sa", e.rp["sa"]); snap("r_fit", e.ybuf["r_fit"])
    for ep in range(EPOCHS):
        e.msg["tr"].copy_(e.theta["tr"]); e.msg["critic"].copy_(e.theta["critic"])
        e._local_fit("tr", "sa", e.ybuf["r_fit"], B, e.coop)
        snap("ep%d msg tr" % ep, e.msg["tr"])
        e._value("ns", e.theta["critic"], "critic", e.ybuf["y_c"], B, r_applied=e.ybuf["r_fit"])
        e._local_fit("critic", "s", e.ybuf["y_c"], B, e.coop)
        snap("ep%d msg critic" % ep, e.msg["critic"])
        e._consensus("critic", "s", B)
        snap("ep%d theta critic" % ep, e.theta["critic"])
        e._consensus("tr", "sa", B)
        snap("ep%d theta tr" % ep, e.theta["tr"])
    e._actor_update(B)
    snap("theta actor", e.theta["actor"])
    torch.cuda.synchronize()
    return snaps


a = run()
gc.collect()
b = run()
for (na, ta), (nb, tb) in zip(a, b):
    ok = bool(torch.equal(torch.nan_to_num(ta, nan=7.0), torch.nan_to_num(tb, nan=7.0)))
    print("%-22s %s" % (na, "same" if ok else "DIFFERENT max|d|=%.3e" % float((ta.double() - tb.double()).abs().nan_to_num().max())))
