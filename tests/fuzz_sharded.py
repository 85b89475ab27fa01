"""Randomised runs of the agent-sharded instance (RPBCACEngine.shard_agents) against the unsharded engine, BIT FOR BIT (TEST TOOL,
not collected by pytest): world sizes 2-4 as threads of this process on one GPU (parallel.ThreadComm), random team sizes, graphs, H,
critic widths, lattice / dense layer 1, episode geometry.

    python tests/fuzz_sharded.py SEED COUNT
"""
import os
import sys
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
sys.path.insert(0, _HERE)
import engine_checks as EC  # noqa: E402
from rcmarl_amd import capi  # noqa: E402
from rcmarl_amd.parallel import ThreadComm  # noqa: E402


def snapshot(eng, logs):
    out = {"theta_" + k: v.cpu().numpy() for k, v in eng.theta.items()}
    out.update({"adam_m": eng.adam_m.cpu().numpy(), "loss_c": eng.loss["critic"].cpu().numpy(), "loss_tr": eng.loss["tr"].cpu().numpy()})
    out.update({"rp_" + k: v[:, :eng.B].cpu().numpy() for k, v in eng.rp.items()})
    out.update({"log_" + k: np.asarray(v) for k, v in logs.items()})
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rng = np.random.default_rng(seed)
    lib = capi.load()
    failed = 0
    for _ in range(count):
        world = int(rng.choice([2, 3, 4]))
        n_loc = int(rng.integers(1, 9))
        hid = int(rng.choice([24, 40, 64, 128, 512]))
        lattice = bool(rng.random() < 0.5) and (n_loc * hid) % 128 == 0
        n = world * n_loc
        if n < 3:
            continue
        d = int(rng.integers(2, min(n, 10) + 1))
        H = int(rng.integers(0, (d - 1) // 2 + 1))
        circ = bool(rng.random() < 0.5)
        if circ:
            nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
        else:
            nodes = [[i] + [int(x) for x in rng.permutation([j for j in range(n) if j != i])[:d - 1]] for i in range(n)]
        mel, nef, nep = int(rng.integers(2, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 3))
        buf = int(rng.integers(mel * nef, mel * nef * 3))
        neps = nef * int(rng.integers(1, 3)) + int(rng.integers(0, nef))
        mode = str(rng.choice(["numpy", "device"]))
        args = EC.make_args(["Cooperative"] * n, H=H, n_episodes=neps, max_ep_len=mel, n_ep_fixed=nef, n_epochs=nep, buffer_size=buf,
                            seed=int(rng.integers(1000)), in_nodes=nodes, fast_lr=0.002, common_reward=bool(rng.random() < 0.3))
        desc = dict(world=world, n=n, hid=hid, lattice=lattice, d=d, H=H, circ=circ, ep=(mel, nef, nep, buf, neps), rng=mode)
        sd = int(rng.integers(100))
        W, goals = EC.make_inputs(args, 6, (sd,), critic_hid=hid)
        try:
            ref_eng, ref_logs = EC.run_engine(args, 6, 6, mode, "cuda", lib, (sd,), W, goals, lattice=lattice, critic_hid=hid)
            ref = snapshot(ref_eng, ref_logs)
            comms, results, errors = ThreadComm.make(world), [None] * world, []

            def rank_main(r):
                try:
                    torch.cuda.set_device(0)
                    eng, logs = EC.run_engine(args, 6, 6, mode, "cuda", lib, (sd,), W, goals, lattice=lattice, critic_hid=hid,
                                              tweak=lambda e: e.shard_agents(comm=comms[r]))
                    assert eng.shard is not None and eng.shard.n_loc == n_loc
                    results[r] = snapshot(eng, logs)
                except BaseException as e:      # noqa: BLE001
                    errors.append((r, repr(e)[:200]))
                    comms[r]._sh["barrier"].abort()
            threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not errors, errors
            for r in range(world):
                for k in ref:
                    np.testing.assert_array_equal(results[r][k], ref[k], err_msg="rank %d %s" % (r, k))
            print("OK  ", desc, flush=True)
        except Exception as e:                  # noqa: BLE001
            failed += 1
            print("FAIL", desc, repr(e)[:300], flush=True)
    print("%d failed" % failed)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
