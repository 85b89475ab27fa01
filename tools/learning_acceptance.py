#!/usr/bin/env python
"""Statistical acceptance run against the reference's learning curves.

    gpurun -- 'python tools/learning_acceptance.py --seeds 64 --out gpurun_out/learning_r02.json'

The reference's shipped results (simulation_results/raw_data/{coop,faulty,greedy,malicious}/H={0,1}/seed={100,200,300})
were produced by its two-phase protocol (job.sh + out.txt:6): `main.py --H=h --slow_lr=0.002 --random_seed=s` for 4000
episodes, then the same command with --pretrained_agents True for 4000 more (fresh process: Adam slots and replay
start empty, networks and desired state are reloaded).  This script runs that protocol for every scenario on the
batched engine with many seeds at once and compares the mean team return of the last 500 episodes of phase 2 with
tests/golden/reference_learning_band.json (made from the shipped pickles by tests/golden/make_learning_band.py).
The shipped pickles come from an older revision of the reference (an `eps` key in its logged args): a band, not a
golden vector.  TensorFlow's Glorot draws are not reproducible here; weights are NumPy-seeded Glorot."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCEN = {"coop": "Cooperative", "faulty": "Faulty", "greedy": "Greedy", "malicious": "Malicious"}
IN_NODES = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]]


def run_scenario(name, H, seeds, n_episodes, last_n):
    from rcmarl_amd.engine import EngineConfig, RPBCACEngine
    labels = ["Cooperative"] * 4 + [SCEN[name]]
    S = len(seeds)
    cfg = EngineConfig(5, labels, IN_NODES, H=H, gamma=0.9, slow_lr=0.002, fast_lr=0.01, max_ep_len=20, n_ep_fixed=50,
                       n_epochs=10, buffer_size=2000, nrow=5, ncol=5, n_seeds=S, rng_mode="device")
    eng = RPBCACEngine(cfg, seeds=seeds)
    eng.init_glorot(base_seed=2)
    goals = []
    for s in seeds:                                  # main.py:46-48: np.random.seed(seed); s_desired = randint(0, 5, (5, 2))
        goals.append(np.random.RandomState(int(s)).randint(0, 5, size=(5, 2)))
    eng.set_goals(np.stack(goals))
    t0 = time.perf_counter()
    logs1 = eng.train(n_episodes)
    # phase 2 = a fresh process with --pretrained_agents True (main.py:52-54): networks (and the Malicious agent's private
    # critic) and desired state carry over; Adam slots, step counts and the replay lists start empty
    eng.adam_m.zero_()
    eng.adam_v.zero_()
    eng.adam_t = 0
    if hasattr(eng, "adv"):
        eng.adv.adam_t = 0
    eng.B = 0
    eng.rows_episode_aligned = True
    logs2 = eng.train(n_episodes)
    eng.sync()
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(eng.theta[k]).all().item()) for k in eng.theta)
    out = {"labels": labels, "H": H, "n_seeds": S, "seconds": round(dt, 2), "weights_finite": finite}
    for ph, lg in (("phase1", logs1), ("phase2", logs2)):
        team = lg["True_team_returns"][-last_n:].mean(axis=0)           # [S]
        adv = lg["True_adv_returns"][-last_n:].mean(axis=0)
        first = lg["True_team_returns"][:last_n].mean(axis=0)
        out[ph] = {"team_last%d_mean" % last_n: float(team.mean()), "team_last%d_std_over_seeds" % last_n: float(team.std()),
                   "team_last%d_min" % last_n: float(team.min()), "team_last%d_max" % last_n: float(team.max()),
                   "adv_last%d_mean" % last_n: float(adv.mean()), "team_first%d_mean" % last_n: float(first.mean()),
                   "per_seed_first8": [float(x) for x in team[:8]]}
    out["curve_phase2_mean_over_seeds_every100"] = [float(x) for x in
                                                    logs2["True_team_returns"].mean(axis=1).reshape(-1, 100).mean(axis=1)]
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--episodes", type=int, default=4000, help="per phase (the reference: 4000 + 4000)")
    ap.add_argument("--last", type=int, default=500)
    ap.add_argument("--scenarios", default="coop,faulty,greedy,malicious")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "learning_r02.json"))
    a = ap.parse_args()
    with open(os.path.join(ROOT, "tests", "golden", "reference_learning_band.json")) as f:
        band = json.load(f)["scenarios"]
    seeds = [100, 200, 300] + [1000 + k for k in range(max(0, a.seeds - 3))]
    seeds = seeds[:a.seeds]
    res = {"protocol": "phase 1: %d episodes from NumPy-seeded Glorot weights; phase 2: %d more with Adam slots and replay reset "
                       "(= the reference's --pretrained_agents restart); N=5, 5x5 grid, slow_lr 0.002, fast_lr 0.01, gamma 0.9, "
                       "n_ep_fixed 50, n_epochs 10, buffer 2000 (out.txt:6); %d seeds batched on one GPU, rng_mode device"
                       % (a.episodes, a.episodes, len(seeds)),
           "seeds": seeds, "scenarios": {}}
    key = "team_last%d_mean" % a.last
    for name in a.scenarios.split(","):
        for H in (0, 1):
            r = run_scenario(name, H, seeds, a.episodes, a.last)
            ref = band["%s/H=%d" % (name, H)]
            r["reference"] = {"mean": ref["team_last500_mean"], "min": ref["team_last500_min"], "max": ref["team_last500_max"],
                              "n_seeds": len(ref["seeds"])}
            sd = r["phase2"]["team_last%d_std_over_seeds" % a.last]
            r["within_reference_spread"] = bool(ref["team_last500_min"] - sd <= r["phase2"][key] <= ref["team_last500_max"] + sd)
            # the stricter reading: the standard error of OUR mean over the seeds against the reference's three draws
            se = sd / max(1.0, float(len(seeds)) ** 0.5)
            r["mean_standard_error"] = se
            r["within_reference_range_by_standard_error"] = bool(ref["team_last500_min"] - 2 * se <= r["phase2"][key] <= ref["team_last500_max"] + 2 * se)
            res["scenarios"]["%s/H=%d" % (name, H)] = r
            print("%-10s H=%d  ours %.3f +- %.3f (seed std)   reference %.3f [%.3f, %.3f]   %s   %.1f s" %
                  (name, H, r["phase2"][key], sd, ref["team_last500_mean"], ref["team_last500_min"], ref["team_last500_max"],
                   "within" if r["within_reference_spread"] else "OUTSIDE", r["seconds"]), flush=True)
    rec = {}
    for name in a.scenarios.split(","):
        if name == "coop":
            continue
        ours = res["scenarios"][name + "/H=1"]["phase2"][key] - res["scenarios"][name + "/H=0"]["phase2"][key]
        refd = band[name + "/H=1"]["team_last500_mean"] - band[name + "/H=0"]["team_last500_mean"]
        rec[name] = {"ours_H1_minus_H0": ours, "reference_H1_minus_H0": refd, "H1_recovers": bool(ours >= 0.5 * refd)}
    res["resilience"] = rec
    res["criterion"] = ("within_reference_spread: our mean over seeds lies in [reference min - s, reference max + s] with s = our "
                        "seed-to-seed std (each reference seed is one draw from a distribution of that width) -- a GENEROUS band; "
                        "within_reference_range_by_standard_error: the same with 2 standard errors of our mean instead of s.  By that "
                        "reading the adversarial H=1 scenarios sit 0.3-0.7 BELOW the reference's three shipped seeds, and so does the "
                        "CPU oracle (profiles/learning_r02a_oracle_check.json: -6.26 +- 0.35 over 6 seeds): engine = oracle; both differ "
                        "from the shipped curves under adversaries (the shipped pickles come from an older revision of the reference).  "
                        "H1_recovers: our H=1 minus H=0 gain is at least half the reference's (README.md:31-45)")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
