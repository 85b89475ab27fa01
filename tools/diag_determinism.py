"""Diagnostic: the same small adversarial scenario run repeatedly must give bit-identical weights."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcmarl_amd.engine import EngineConfig, RPBCACEngine
labels = ["Cooperative"] * 4 + [os.environ.get("ADV", "Malicious")]
S = int(os.environ.get("S", "256"))
ref = None
for rep in range(int(os.environ.get("REPS", "6"))):
    cfg = EngineConfig(5, labels, [[(i + k) % 5 for k in range(4)] for i in range(5)], H=1, n_seeds=S, rng_mode="device")
    eng = RPBCACEngine(cfg, seeds=list(range(100, 100 + S)))
    eng.init_glorot(base_seed=1)
    eng.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(5, 2)) for s in range(S)]))
    eng.train(int(os.environ.get("EPISODES", "100")))
    w = {k: eng.theta[k].detach().cpu().numpy().copy() for k in eng.theta}
    fin = all(np.isfinite(v).all() for v in w.values())
    if ref is None:
        ref = prev = w
        print("rep 0 finite=%s" % fin)
    else:
        print("rep %d finite=%s vs rep0 %s | vs prev %s" % (rep, fin,
              {k: "%.2e" % float(np.nanmax(np.abs(ref[k].astype(np.float64) - w[k]))) for k in w},
              {k: "%.2e" % float(np.nanmax(np.abs(prev[k].astype(np.float64) - w[k]))) for k in w}))
        prev = w
