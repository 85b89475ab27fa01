"""Diagnostic: two identical engines stepped phase by phase; reports the first phase whose outputs differ bitwise."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcmarl_amd.engine import EngineConfig, RPBCACEngine
N = int(os.environ.get("N", "5")); S = int(os.environ.get("S", "256")); d = int(os.environ.get("D", "4")); H = int(os.environ.get("H", "1"))
G = int(os.environ.get("GRID", "5"))


def make():
    cfg = EngineConfig(N, ["Cooperative"] * N, [[(i + k) % N for k in range(d)] for i in range(N)], H=H, n_seeds=S, rng_mode="device",
                       nrow=G, ncol=G, fast_lr=float(os.environ.get("FAST_LR", "0.01")))
    e = RPBCACEngine(cfg, seeds=list(range(100, 100 + S)))
    e.init_glorot(base_seed=1)
    e.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(N, 2)) for s in range(S)]))
    return e


def same(name, ta, tb):
    torch.cuda.synchronize()
    ok = bool(torch.equal(ta, tb)) or bool(torch.equal(torch.nan_to_num(ta, nan=7.0), torch.nan_to_num(tb, nan=7.0)))
    print("%-28s %s" % (name, "same" if ok else "DIFFERENT  max|d|=%.3e" % float((ta.double() - tb.double()).abs().nan_to_num().max())))
    return ok


A, Bn = make(), make()
for e in (A, Bn):
    e.rollout_block(e.cfg.n_ep_fixed)
for k in A.rp:
    same("replay " + k, A.rp[k], Bn.rp[k])
B = A.B
for e in (A, Bn):
    c, L = e.cfg, e.lib
    e._lattice_encode(B)
    rptr, rstride = e._x("r")
    L.rcmarl_team_reward(rptr, rstride, e.coop.data_ptr(), max(e.n_coop, 1), e.rcoop.data_ptr(), e.S, e.N, B, e.ldb, e.stream)
    L.rcmarl_gather_agent_major(rptr, rstride, e.rcoop.data_ptr(), e.fit_mode.data_ptr(), e.ybuf["r_fit"].data_ptr(), e.S, e.N, B, e.ldb, e.stream)
same("r_fit", A.ybuf["r_fit"], Bn.ybuf["r_fit"])
for ep in range(2):
    for e in (A, Bn):
        e.msg["tr"].copy_(e.theta["tr"]); e.msg["critic"].copy_(e.theta["critic"])
        e._value("ns", e.theta["critic"], "critic", e.ybuf["y_c"], B, r_applied=e.ybuf["r_fit"])
    same("ep%d y_c" % ep, A.ybuf["y_c"], Bn.ybuf["y_c"])
    for e in (A, Bn):
        e._local_fit("tr", "sa", e.ybuf["r_fit"], B, e.coop)
    same("ep%d msg tr" % ep, A.msg["tr"], Bn.msg["tr"])
    for e in (A, Bn):
        e._local_fit("critic", "s", e.ybuf["y_c"], B, e.coop)
    same("ep%d msg critic" % ep, A.msg["critic"], Bn.msg["critic"])
    for e in (A, Bn):
        e._consensus("critic", "s", B)
    same("ep%d theta critic" % ep, A.theta["critic"], Bn.theta["critic"])
    for e in (A, Bn):
        e._consensus("tr", "sa", B)
    same("ep%d theta tr" % ep, A.theta["tr"], Bn.theta["tr"])
for e in (A, Bn):
    e._actor_update(B)
same("theta actor", A.theta["actor"], Bn.theta["actor"])
