"""Randomised engine-vs-oracle differential runs (TEST TOOL, not collected by pytest): random team compositions (Cooperative /
Faulty / Greedy / Malicious), graphs (circulant or random, any in-degree), H, grid sizes, episode / block / buffer geometry,
RNG modes, common_reward, and -- for teams without fitting adversaries -- critic widths.  Every run must pass
engine_checks.compare (the bar of the fixed-shape tests).

    python tests/fuzz_engine.py SEED COUNT [cuda]        # default: hipemu build on the CPU
    RCMARL_FUZZ_NMAX=24 RCMARL_FUZZ_ONLY=177 python tests/fuzz_engine.py 606 250   # replay ONE draw of a cuda run on the emulation
"""
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
sys.path.insert(0, _HERE)
import engine_checks as EC  # noqa: E402

POOL = ["Cooperative", "Cooperative", "Cooperative", "Faulty", "Greedy", "Malicious"]


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    gpu = len(sys.argv) > 3 and sys.argv[3] == "cuda"
    if gpu:
        device, lib, nmax = "cuda", None, 24
    else:
        from emu_util import emu_lib
        device, lib, nmax = "cpu", emu_lib(), 8
    nmax = int(os.environ.get("RCMARL_FUZZ_NMAX", nmax))        # the draws depend on it: set it to 24 to replay a cuda run's stream
    only = os.environ.get("RCMARL_FUZZ_ONLY")
    rng = np.random.default_rng(seed)
    failed = 0
    for it in range(count):
        n = int(rng.integers(3, nmax))
        labs = [str(rng.choice(POOL)) for _ in range(n)]
        if labs.count("Cooperative") < 2:
            labs[0] = labs[1] = "Cooperative"
        hid = 20
        if rng.random() < 0.3:                                   # a wide critic runs with Cooperative and Faulty agents
            labs = [lab if lab in ("Cooperative", "Faulty") else "Cooperative" for lab in labs]
            hid = int(rng.choice([24, 40, 64, 128]))             # (128 beside the lattice layer 1: the packed-operand path, round 6)
        d = int(rng.integers(2, min(n, 12) + 1))
        H = int(rng.integers(0, (d - 1) // 2 + 1))
        circ = bool(rng.random() < 0.5)
        if circ:
            nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
        else:
            nodes = [[i] + [int(x) for x in rng.permutation([j for j in range(n) if j != i])[:d - 1]] for i in range(n)]
        mel, nef, nep = int(rng.integers(2, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
        buf = int(rng.integers(mel * nef, mel * nef * 3))
        neps = nef * int(rng.integers(1, 3)) + int(rng.integers(0, nef))
        mode = str(rng.choice(["numpy", "device"]))
        common = bool(rng.random() < 0.3)
        nrow, ncol = int(rng.integers(3, 7)), int(rng.integers(3, 7))
        lattice = "auto" if hid == 20 else bool(rng.random() < 0.5)
        args = EC.make_args(labs, H=H, n_episodes=neps, max_ep_len=mel, n_ep_fixed=nef, n_epochs=nep, buffer_size=buf,
                            seed=int(rng.integers(1000)), in_nodes=nodes, common_reward=common, fast_lr=0.005)
        desc = dict(n=n, labels="".join(lab[0] for lab in labs), d=d, H=H, circ=circ, ep=(mel, nef, nep, buf, neps), rng=mode,
                    common=common, grid=(nrow, ncol), critic_hid=hid, lattice=lattice)
        t0 = time.time()
        seeds = (int(rng.integers(100)), int(rng.integers(100, 200)))
        if only is not None and it != int(only):
            continue
        try:
            eng, logs, o_logs, o_w = EC.run_pair(args, nrow, ncol, mode, device, lib, seeds=seeds, critic_hid=hid, lattice=lattice)
            EC.compare(eng, logs, o_logs, o_w, actor="stat" if n > 12 else "strict")
            print("OK   %5.1fs %s" % (time.time() - t0, desc), flush=True)
        except Exception as e:                                   # noqa: BLE001 -- report and go on
            failed += 1
            print("FAIL %s %s" % (desc, repr(e)[:300]), flush=True)
    print("%d of %d failed" % (failed, count))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
