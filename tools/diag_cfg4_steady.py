#!/usr/bin/env python
"""One update block of BASELINE configs[3] at the bench's STEADY STATE against the oracle from IDENTICAL state (GPU + the host's cores):
the engine trains two blocks and rolls out the third, its weights / Adam slots / 3000 replay rows go into oracle.update_block
(training/train_agents.py:100-153), then both run the third update (10 epochs, live actors, fast_lr 0.001).  Per-network worst
|w - w_oracle| / max(1, |w|max) as a DISTRIBUTION per family, default and exact operand form.

    python tools/diag_cfg4_steady.py [n_seeds] > profiles/r06_cfg4_steady_state_parity.txt"""
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
    n, d = 256, 18
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=0, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    seeds = tuple(1000 + k for k in range(nseeds))
    print("BASELINE configs[3], ONE update block at the steady state from identical state: B = 3000, 10 epochs, live actors (slow_lr 0.002), "
          "fast_lr 0.001, %d seeds x 256 agents; oracle: one process per (form, seed)" % nseeds, flush=True)
    forms = (("default: two f16 pieces, f16 mid kernel", 3, None), ("exact: three bf16 pieces, fp32 mid kernel", 0, "5"))
    engs, snaps = [], []
    for label, mode, midfit in forms:
        L.rcmarl_lattice_set_f16_mode(mode)
        os.environ.pop("RCMARL_MIDFIT", None)
        if midfit:
            os.environ["RCMARL_MIDFIT"] = midfit
        eng, sn = EC.check_block_from_injected_state(args, 32, 32, "cuda", None, seeds, blocks_before=2, oracle_later=True)
        engs.append(eng)
        snaps += sn
    o_all = EC.run_oracle_blocks_parallel(dict(args), snaps)
    for k, (label, mode, midfit) in enumerate(forms):
        L.rcmarl_lattice_set_f16_mode(mode)
        os.environ.pop("RCMARL_MIDFIT", None)
        if midfit:
            os.environ["RCMARL_MIDFIT"] = midfit
        eng, o_w = engs[k], o_all[k * nseeds:(k + 1) * nseeds]
        eng.update_block()
        eng.sync()
        for net, e in EC.network_errors(eng, o_w).items():
            print("%-44s %-6s per-network worst: median %.2e  90%% %.2e  99%% %.2e  max %.2e | beyond 1e-4: %d of %d, beyond 3e-4: %d"
                  % (label, net, np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max(), int((e > 1e-4).sum()), e.size,
                     int((e > 3e-4).sum())), flush=True)
        frac, worst = EC.actor_stat(eng, o_w, args["slow_lr"])
        print("%-44s actor  %.2e of the parameters beyond 5 %% of an Adam step, max |err| %.2e = %.2f steps; Adam steps taken %d"
              % (label, frac, worst, worst / args["slow_lr"], eng.adam_t), flush=True)
    L.rcmarl_lattice_set_f16_mode(-1)
    os.environ.pop("RCMARL_MIDFIT", None)


if __name__ == "__main__":          # (the oracle's worker processes import this file: nothing runs there)
    main()
