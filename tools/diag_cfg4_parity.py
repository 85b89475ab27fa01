#!/usr/bin/env python
"""Diagnostic (GPU + ~4 CPU-minutes of oracle): BASELINE configs[3] at the bench's hyper-parameters, two seeds, two blocks (the run of
tests/test_engine_baseline_shapes_gpu.py::cfg4_oracle) -- per-agent |w - w_oracle| / max(1, |w|max) of the critic and team-reward
nets as a DISTRIBUTION over the 512 (seed, agent) networks, in the default operand form (two f16 pieces, f16 mid kernel) and in the
exact form (three bf16 pieces, fp32 mid kernel).  Which bar does the data support, and is a miss the operand form or the
summation order?      python tools/diag_cfg4_parity.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import engine_checks as EC  # noqa: E402
from rcmarl_amd import capi  # noqa: E402


def main():
    L = capi.load()
    BENCH = len(sys.argv) > 1 and sys.argv[1] == "bench"      # the configuration bench.py times: 10 epochs, live actor, ONE block, 4 seeds
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=100, max_ep_len=20, n_ep_fixed=50, n_epochs=2, buffer_size=1000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.0)
    seeds = (1000, 1001)
    if BENCH:
        args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=50, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                            seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
        seeds = (1000, 1001, 1002, 1003)
        print("BASELINE configs[3] at the bench's configuration: 10 epochs, live actor (slow_lr 0.002), fast_lr 0.001, ONE 50-episode "
              "block (B = 1000), %d seeds x 256 agents; oracle: one process per seed" % len(seeds), flush=True)
    W, goals = EC.make_inputs(args, 32, seeds)
    o_logs, o_w = EC.run_oracle_parallel(args, 32, 32, "device", seeds, W, goals)
    for label, mode, midfit in (("default: two f16 pieces, f16 mid kernel", 3, None), ("two f16 pieces, fp32 mid kernel", 3, "5"),
                                ("exact: three bf16 pieces, fp32 mid kernel", 0, "5")):
        L.rcmarl_lattice_set_f16_mode(mode)
        os.environ["RCMARL_LAT_F16"] = str(mode)
        if midfit:
            os.environ["RCMARL_MIDFIT"] = midfit
        else:
            os.environ.pop("RCMARL_MIDFIT", None)
        eng, logs = EC.run_engine(args, 32, 32, "device", "cuda", None, seeds, W, goals)
        same_actions = all(np.array_equal(logs["True_team_returns"][:, s], o_logs[s]["True_team_returns"].to_numpy(dtype=np.float64))
                           for s in range(len(seeds)))
        for k, net in ((1, "critic"), (2, "tr")):
            errs = []
            for s in range(len(seeds)):
                for i in range(n):
                    e = 0.0
                    for a, b in zip(eng.get_weights(s, i, net), o_w[s][i][k]):
                        e = max(e, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
                    errs.append(e)
            e = np.asarray(errs)
            print("%-46s %-6s same action streams %s | per-network worst: median %.2e  90%% %.2e  99%% %.2e  max %.2e | beyond 1e-4: %d of %d, "
                  "beyond 3e-4: %d" % (label, net, same_actions, np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max(),
                                       int((e > 1e-4).sum()), e.size, int((e > 3e-4).sum())), flush=True)
        if BENCH:
            from rcmarl_amd.engine import flatten_params  # noqa: F401
            print("%-46s actor  Adam steps taken: %d" % (label, eng.adam_t), flush=True)
    L.rcmarl_lattice_set_f16_mode(-1)


if __name__ == "__main__":          # (the oracle's worker processes import this file: nothing runs there)
    main()
