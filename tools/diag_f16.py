"""Diagnostic: the cfg4_shard bench engine block by block under RCMARL_LAT_F16 = 3 / 1 / 0: non-finite weights and largest
magnitudes per network and seed (first appearance of a divergence)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rcmarl_amd import capi
import torch

w = dict(bench.WORKLOADS[os.environ.get("WORKLOAD", "cfg4_shard")])
if os.environ.get("FAST_LR"):
    w["fast_lr"] = float(os.environ["FAST_LR"])
S = int(os.environ.get("SEEDS", w["S"]))
blocks = int(os.environ.get("BLOCKS", "8"))
lib = capi.load()
for mode in os.environ.get("MODES", "3,0").split(","):
    os.environ["RCMARL_LAT_F16"] = mode
    lib.rcmarl_lattice_set_f16_mode(int(mode))           # (the library reads the environment only once)
    seed0 = int(os.environ.get("SEED0", "1000"))
    eng = bench.make_engine(w, S, list(range(seed0, seed0 + S)), lib)
    for b in range(blocks):
        team, adv, est = eng.run_block()
        torch.cuda.synchronize()
        line = "seed0=%d " % seed0 + "mode=%s block=%d B=%d lat=%s ret=%.3f" % (mode, b, eng.B, eng.lat_active, float(np.mean(team)))
        for net in ("critic", "tr", "actor"):
            x = eng.theta[net] if hasattr(eng, "theta") else None
            x = x.detach().float().cpu().numpy()
            bad = ~np.isfinite(x)
            seeds_bad = np.nonzero(bad.reshape(x.shape[0], -1).any(1))[0]
            line += " | %s bad=%d seeds=%s absmax=%.4g" % (net, int(bad.sum()), list(seeds_bad[:6]), float(np.nanmax(np.abs(np.where(bad, 0, x)))))
        print(line, flush=True)
    del eng
    torch.cuda.empty_cache()
