#!/usr/bin/env python
"""tests/golden/reference_learning_band.json from the reference's SHIPPED result pickles (build container only):

    python tests/golden/make_learning_band.py

For every scenario {coop, faulty, greedy, malicious} x H in {0, 1} x seed in {100, 200, 300}: the mean of the last 500
episodes of `True_team_returns` / `True_adv_returns` of phase 2 (sim_data2.pkl; phase 1 = sim_data1.pkl) -- the
numbers BASELINE.md quotes -- with the run configuration logged in out.txt:6.  The pickles come from an older
revision of the reference (its logged args carry an `eps` key main.py no longer has), so this is a STATISTICAL
acceptance band for tools/learning_acceptance.py, not a golden vector."""
import ast
import json
import os

import numpy as np
import pandas as pd

REF = "/root/reference/simulation_results/raw_data"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {"source": "simulation_results/raw_data/{scenario}/H={H}/seed={seed}/sim_data{1,2}.pkl", "last_n": 500, "scenarios": {}}
    for sc in ("coop", "faulty", "greedy", "malicious"):
        for H in (0, 1):
            rec = {"seeds": {}, "args": None}
            for seed in (100, 200, 300):
                d = os.path.join(REF, sc, "H=%d" % H, "seed=%d" % seed)
                if not os.path.exists(os.path.join(d, "sim_data2.pkl")):
                    continue
                if rec["args"] is None:
                    with open(os.path.join(d, "out.txt")) as f:
                        line = [l for l in f.readlines()[:12] if l.startswith("{'n_agents'")][0]
                    a = ast.literal_eval(line[:line.index("}") + 1])
                    rec["args"] = {k: a[k] for k in ("agent_label", "in_nodes", "n_episodes", "max_ep_len", "n_ep_fixed", "n_epochs",
                                                     "slow_lr", "fast_lr", "gamma", "buffer_size", "H", "common_reward")}
                r = {}
                for ph in (1, 2):
                    df = pd.read_pickle(os.path.join(d, "sim_data%d.pkl" % ph))
                    r["phase%d" % ph] = {"episodes": int(len(df)),
                                         "team_last500": float(df["True_team_returns"].to_numpy()[-500:].mean()),
                                         "adv_last500": float(df["True_adv_returns"].to_numpy()[-500:].mean()),
                                         "team_first500": float(df["True_team_returns"].to_numpy()[:500].mean())}
                rec["seeds"][str(seed)] = r
            v = [s["phase2"]["team_last500"] for s in rec["seeds"].values()]
            rec["team_last500_mean"], rec["team_last500_min"], rec["team_last500_max"] = float(np.mean(v)), float(min(v)), float(max(v))
            out["scenarios"]["%s/H=%d" % (sc, H)] = rec
    path = os.path.join(HERE, "reference_learning_band.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)
    for k, r in out["scenarios"].items():
        print("%-14s mean %.3f  [%.3f, %.3f]" % (k, r["team_last500_mean"], r["team_last500_min"], r["team_last500_max"]))


if __name__ == "__main__":
    main()
