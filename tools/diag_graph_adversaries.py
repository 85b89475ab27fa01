"""Diagnostic (GPU): one 5-agent instance with a Malicious agent, update epochs captured into a hipGraph (RCMARL_GRAPH=1).  With the
adversaries' fits on their side streams (default) the capture segfaults inside the HIP runtime (ROCm 7.2, round 4); with
RCMARL_ADV_ASYNC=0 it prints "ok 3 15".   python -X faulthandler tools/diag_graph_adversaries.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ["RCMARL_GRAPH"] = "1"
import numpy as np
import engine_checks as EC
from rcmarl_amd.engine import EngineConfig, RPBCACEngine
labels = ["Cooperative"] * 4 + ["Malicious"]
n, S = 5, 1
cfg = EngineConfig(n, labels, EC.CIRC5, H=1, n_seeds=S, rng_mode="device", max_ep_len=20, n_ep_fixed=10, n_epochs=4, buffer_size=400, nrow=5, ncol=5)
eng = RPBCACEngine(cfg, seeds=[200])
eng.init_glorot(base_seed=2)
eng.set_goals(np.stack([np.random.RandomState(s).randint(0, 5, size=(n, 2)) for s in range(S)]))
logs = eng.train(50)
print("ok", eng.graph_captures, eng.graph_replays)
