"""Diagnostic: repeat each consensus kernel on identical inputs inside one process; report which one varies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcmarl_amd.engine import EngineConfig, RPBCACEngine, HID
N = 5; S = int(os.environ.get("S", "256"))
cfg = EngineConfig(N, ["Cooperative"] * N, [[(i + k) % N for k in range(4)] for i in range(N)], H=1, n_seeds=S, rng_mode="device")
e = RPBCACEngine(cfg, seeds=list(range(100, 100 + S)))
e.init_glorot(base_seed=1)
e.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(N, 2)) for s in range(S)]))
e.rollout_block(cfg.n_ep_fixed)
B = e.B; L = e.lib; c = cfg
e._lattice_encode(B)
rptr, rstride = e._x("r")
L.rcmarl_team_reward(rptr, rstride, e.coop.data_ptr(), max(e.n_coop, 1), e.rcoop.data_ptr(), e.S, e.N, B, e.ldb, e.stream)
L.rcmarl_gather_agent_major(rptr, rstride, e.rcoop.data_ptr(), e.fit_mode.data_ptr(), e.ybuf["r_fit"].data_ptr(), e.S, e.N, B, e.ldb, e.stream)
e.msg["critic"].copy_(e.theta["critic"])
e._value("ns", e.theta["critic"], "critic", e.ybuf["y_c"], B, r_applied=e.ybuf["r_fit"])
e._local_fit("critic", "s", e.ybuf["y_c"], B, e.coop)
net = "critic"
theta0 = e.theta[net].clone()
g_hid = e.P[net] - (HID + 1)
outs = []
for rep in range(4):
    e.theta[net].copy_(theta0)
    # pollute LDS between repetitions with an unrelated kernel
    e._value("s", e.theta["critic"], "critic", e.ybuf["v_cur"], B)
    L.rcmarl_consensus_params(e.msg[net].data_ptr(), e.theta[net].data_ptr(), e.nbr.data_ptr(), e.coop.data_ptr(), S, N, e.ldp[net], g_hid, c.d, c.H, None, None, e.stream)
    k1 = e.theta[net].clone()
    a1 = e.a1net[net]
    e._layer1("s", e.theta[net], net, B, buf=a1)
    a1c = a1.clone()
    e.partials.zero_()
    L.rcmarl_consensus_head(a1.data_ptr(), e.theta[net].data_ptr(), e.msg[net].data_ptr(), e.nbr.data_ptr(), e.coop.data_ptr(), e.partials.data_ptr(), None, S, N, B, e.in_dim[net], HID, e.ldp[net], e.ldb, c.d, c.H, e.stream)
    part = e.partials.clone()
    L.rcmarl_head_apply(e.partials.data_ptr(), e.theta[net].data_ptr(), e.coop.data_ptr(), S, N, B, e.in_dim[net], HID, e.ldp[net], e.stream)
    torch.cuda.synchronize()
    outs.append((k1, a1c, part, e.theta[net].clone()))
for rep in range(1, 4):
    print("rep %d: K1 %s | layer1 %s | head partials %s | final %s" % ((rep,) + tuple(
        "same" if torch.equal(torch.nan_to_num(outs[0][i]), torch.nan_to_num(outs[rep][i])) else "DIFF %.2e" % float((outs[0][i].double() - outs[rep][i].double()).abs().nan_to_num().max())
        for i in range(4))))
