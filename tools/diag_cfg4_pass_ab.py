#!/usr/bin/env python
"""PAIRED fp64 arbitration of a kernel variant at BASELINE configs[3]'s steady state (round 6: the mid kernels with three of the four
piece products against the four-pass variant build).

The per-network error counts of the team-reward net move a lot with the STATE an update block starts from (a few per cent of the nets
sit on a LeakyReLU knife edge in any given block), so two builds are compared from the SAME state: the product library runs
`blocks_before` blocks and the next rollout, its checkpoint (every network, Adam slots, replay rows) is loaded into an engine on the
variant library and into one in the exact operand form, the oracle runs that state in fp32, fp32 with reordered fit rows and FLOAT64,
every side runs ONE update block, and every fp32 result is measured against the float64 one.

    python tools/diag_cfg4_pass_ab.py VARIANT.so [blocks_before=2] [n_seeds=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import engine_checks as EC  # noqa: E402
from rcmarl_amd import capi  # noqa: E402
from diag_cfg4_fp64_arbiter import errs_vs, stats, set_form  # noqa: E402


def main():
    variant = sys.argv[1]
    blocks_before = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    LA, LB = capi.load(), capi.CLib(variant)
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=0, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    seeds = tuple(1000 + k for k in range(nseeds))
    B = min(1000 * (blocks_before + 1), 3000)
    print("BASELINE configs[3], ONE update block from ONE state (the product build's, %d blocks before; B = %d), 10 epochs, live actors, fast_lr 0.001, "
          "%d seeds x 256 agents; every fp32 result against the oracle's loop nest in FLOAT64" % (blocks_before, B, nseeds), flush=True)
    set_form(LA, 3, None)
    engA, snaps = EC.check_block_from_injected_state(args, 32, 32, "cuda", LA, seeds, blocks_before=blocks_before, oracle_later=True)
    sd = engA.state_dict()
    set_form(LB, 3, None)
    engB, _ = EC.check_block_from_injected_state(args, 32, 32, "cuda", LB, seeds, blocks_before=0, oracle_later=True)
    engB.load_state_dict(sd)
    engC, _ = EC.check_block_from_injected_state(args, 32, 32, "cuda", LA, seeds, blocks_before=0, oracle_later=True)
    engC.load_state_dict(sd)
    snapsB = [EC.snapshot_for_oracle(engB, s) for s in range(nseeds)]
    same = all(np.array_equal(snaps[s][key], snapsB[s][key]) for s in range(nseeds) for key in ("s", "ns", "r", "a")) and \
        all(np.array_equal(x, y) for s in range(nseeds) for i in range(n) for net in ("critic", "tr", "actor")
            for x, y in zip(snaps[s]["W"][i][net], snapsB[s]["W"][i][net]))
    print("state loaded into the variant engine identical to the product engine's: %s" % same, flush=True)
    jobs, modes = [], []
    for m in ("f32", "f64", "f32_shuffled"):
        jobs += snaps
        modes += [m] * nseeds
    res = EC.run_oracle_blocks_parallel(dict(args), jobs, modes=modes)
    orc = {m: res[k * nseeds:(k + 1) * nseeds] for k, m in enumerate(("f32", "f64", "f32_shuffled"))}
    set_form(LA, 3, None)
    engA.update_block(); engA.sync()
    set_form(LB, 3, None)
    engB.update_block(); engB.sync()
    set_form(LA, 0, "5")
    engC.update_block(); engC.sync()
    set_form(LA, -1, None)
    sides = (("engine, product build (default form)", engA), ("engine, variant %s" % os.path.basename(variant), engB),
             ("engine, exact form (3 x bf16, fp32 mid)", engC))
    hdr = "%-46s %-6s median     90%%       99%%       max      >1e-4  >3e-4"
    for k_net, net in ((1, "critic"), (2, "tr")):
        for ref_name, ref in (("against the float64 oracle", orc["f64"]), ("against the fp32 oracle (the parity tests' view)", orc["f32"])):
            print()
            print(hdr % (ref_name, net))
            for label, E in sides:
                e = errs_vs(ref, lambda s, i, kk, E=E, nn=net: E.get_weights(s, i, nn), nseeds, n, k_net)
                print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((label, net) + stats(e)))
            for m, lab in (("f32", "oracle fp32"), ("f32_shuffled", "oracle fp32, fit rows reordered")):
                if ref is orc["f32"] and m == "f32":
                    continue
                e = errs_vs(ref, lambda s, i, kk, R=orc[m]: R[s][i][kk], nseeds, n, k_net)
                print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((lab, net) + stats(e)))
        eA = errs_vs([[[None, engB.get_weights(s, i, "critic"), engB.get_weights(s, i, "tr")] for i in range(n)] for s in range(nseeds)],
                     lambda s, i, kk, nn=net: engA.get_weights(s, i, nn), nseeds, n, k_net)
        print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % (("product build against the variant build", net) + stats(eA)))


if __name__ == "__main__":          # (the oracle's worker processes import this file: nothing runs there)
    main()
