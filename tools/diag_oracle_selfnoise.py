#!/usr/bin/env python
"""CONTROL for the end-of-block weight bar at BASELINE configs[3] (CPU only, no GPU, no HIP library): the ORACLE against ITSELF with
the replay rows of every full-batch local fit shuffled per epoch -- what Keras' own `fit(shuffle=True)` does to the single batch of
agents/resilient_CAC_agents.py:118,136: the gradient is the same sum, taken in another order ("shuffle only permutes
the single batch => affects summation order only").  Configuration = the one bench.py times (10 epochs, live actor, fast_lr 0.001),
one 50-episode block, one seed.  Prints the distribution over the 256 agents of the per-network worst |w_A - w_B| / max(1, |w|max):
how far two fp32 runs of the REFERENCE'S OWN arithmetic end up from each other -- the floor under any engine-vs-oracle bar.
    python tools/diag_oracle_selfnoise.py [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def run(payload):
    seed, shuffled = payload
    import threadpoolctl
    threadpoolctl.threadpool_limits(max(1, (os.cpu_count() or 2) // 2))
    import engine_checks as EC
    from oracle import mlp_np as M
    if shuffled:
        orig = M.fit_mse
        rng = np.random.default_rng(12345)

        def fit_shuffled(params, x, y, lr, epochs, batch_size=None, perms=None, sample_weight=None):
            if batch_size is None and perms is None:           # a full-batch local fit: same rows, another order every epoch
                B = np.asarray(x).shape[0]
                perms = np.stack([rng.permutation(B) for _ in range(epochs)])
            return orig(params, x, y, lr, epochs, batch_size=batch_size, perms=perms, sample_weight=sample_weight)
        M.fit_mse = fit_shuffled
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=50, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=seed, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    W, goals = EC.make_inputs(args, 32, (seed,))
    logs, w = EC.run_oracle(args, 32, 32, "device", (seed,), W, goals)
    return logs[0], w[0]


def main():
    import concurrent.futures as cf
    import multiprocessing as mp
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    with cf.ProcessPoolExecutor(max_workers=2, mp_context=mp.get_context("spawn")) as ex:
        (la, wa), (lb, wb) = list(ex.map(run, [(seed, False), (seed, True)]))
    same = np.array_equal(la["True_team_returns"].to_numpy(), lb["True_team_returns"].to_numpy())
    print("oracle vs oracle-with-shuffled-fit-rows, BASELINE configs[3] bench configuration, seed %d, one block; same action streams: %s" % (seed, same))
    for k, net in ((1, "critic"), (2, "tr")):
        e = []
        for i in range(256):
            m = 0.0
            for a, b in zip(wa[i][k], wb[i][k]):
                m = max(m, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
            e.append(m)
        e = np.asarray(e)
        print("%-6s per-network worst: median %.2e  90%% %.2e  99%% %.2e  max %.2e | beyond 1e-4: %d of %d, beyond 3e-4: %d"
              % (net, np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max(), int((e > 1e-4).sum()), e.size, int((e > 3e-4).sum())))


if __name__ == "__main__":
    main()
