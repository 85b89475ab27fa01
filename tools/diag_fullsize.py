"""Diagnostic: one full-size block per path; reports non-finite weights per net."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcmarl_amd.engine import EngineConfig, RPBCACEngine
n, d = 256, 18
in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
for lat in (False, True):
    cfg = EngineConfig(n, ["Cooperative"] * n, in_nodes, H=8, max_ep_len=20, n_ep_fixed=50, n_epochs=int(os.environ.get("EPOCHS", "2")),
                       buffer_size=2000, fast_lr=float(os.environ.get("FAST_LR", "0.0025")), nrow=32, ncol=32, n_seeds=2, rng_mode="device", lattice=lat)
    eng = RPBCACEngine(cfg, seeds=[1000, 1001])
    eng.init_glorot(base_seed=1)
    eng.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(n, 2)) for s in (1000, 1001)]))
    eng.train(50)
    for net in ("critic", "tr", "actor"):
        w = eng.get_all_weights(net)
        print("lattice=%s %s: nonfinite=%d absmax=%.4g" % (lat, net, int((~np.isfinite(w)).sum()), float(np.nanmax(np.abs(w)))))
