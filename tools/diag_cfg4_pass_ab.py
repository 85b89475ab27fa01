#!/usr/bin/env python
"""PAIRED fp64 arbitration of a kernel variant at BASELINE configs[3]'s steady state (round 6: the mid kernels with three of the four
piece products against the four-pass variant build).

The per-network error counts of the team-reward net move a lot with the STATE an update block starts from (a few per cent of the nets
sit on a LeakyReLU knife edge in any given block), so two builds are compared from the SAME state: the product library runs
`blocks_before` blocks and the next rollout, its checkpoint (every network, Adam slots, replay rows) is loaded into an engine on the
variant library and into one in the exact operand form, the oracle runs that state in fp32, fp32 with reordered fit rows and FLOAT64,
every side runs ONE update block, and every fp32 result is measured against the float64 one.

    python tools/diag_cfg4_pass_ab.py VARIANT.so|- [blocks_before=2] [n_seeds=4]
"-": no variant; the product build with the mid step, K2 or both back in fp32 (which two-piece component moves the tail?)."""
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
    variant = sys.argv[1]                              # a variant library, or "-" : the product build with one component in fp32 at a time
    blocks_before = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    LA, LB = capi.load(), (capi.CLib(variant) if variant != "-" else None)
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=0, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    seeds = tuple(1000 + k for k in range(nseeds))
    B = min(1000 * (blocks_before + 1), 3000)
    print("BASELINE configs[3], ONE update block from ONE state (the product build's, %d blocks before; B = %d), 10 epochs, live actors, fast_lr 0.001, "
          "%d seeds x 256 agents; every fp32 result against the oracle's loop nest in FLOAT64" % (blocks_before, B, nseeds), flush=True)
    # sides: (label, library, lattice operand mode, RCMARL_MIDFIT, RCMARL_K2_MX)
    sides_def = [("engine, product build (default form)", LA, 3, None, None)]
    if LB is not None:
        sides_def.append(("engine, variant %s" % os.path.basename(variant), LB, 3, None, None))
        # ... and which of the two layer-1 GEMMs' operand forms moves the tail (bit 0: forward W'' in two f16 pieces, bit 1: backward dz'')
        sides_def += [("engine, f16 pieces in the forward GEMM only", LA, 1, None, None), ("engine, f16 pieces in the backward GEMM only", LA, 2, None, None)]
    else:                                              # which two-piece component moves the tail?  one component back to fp32 at a time
        sides_def += [("engine, default but K2 in fp32", LA, 3, None, "0"), ("engine, default but mid step in fp32", LA, 3, "5", None),
                      ("engine, default but mid step AND K2 in fp32", LA, 3, "5", "0")]
    sides_def.append(("engine, exact form (3 x bf16, fp32 mid)", LA, 0, "5", None))

    def apply(lib, mode, midfit, k2):
        set_form(lib, mode, midfit)
        os.environ.pop("RCMARL_K2_MX", None)
        if k2 is not None:
            os.environ["RCMARL_K2_MX"] = k2

    apply(LA, 3, None, None)
    engA, snaps = EC.check_block_from_injected_state(args, 32, 32, "cuda", LA, seeds, blocks_before=blocks_before, oracle_later=True)
    sd = engA.state_dict()
    engines = [engA]
    for label, lib, mode, midfit, k2 in sides_def[1:]:
        apply(lib, 3, None, None)
        e, _ = EC.check_block_from_injected_state(args, 32, 32, "cuda", lib, seeds, blocks_before=0, oracle_later=True)
        e.load_state_dict(sd)
        engines.append(e)
    engB = engines[1]
    snapsB = [EC.snapshot_for_oracle(engB, s) for s in range(nseeds)]
    same = all(np.array_equal(snaps[s][key], snapsB[s][key]) for s in range(nseeds) for key in ("s", "ns", "r", "a")) and \
        all(np.array_equal(x, y) for s in range(nseeds) for i in range(n) for net in ("critic", "tr", "actor")
            for x, y in zip(snaps[s]["W"][i][net], snapsB[s]["W"][i][net]))
    print("state loaded into the second engine identical to the product engine's: %s" % same, flush=True)
    jobs, modes = [], []
    for m in ("f32", "f64", "f32_shuffled"):
        jobs += snaps
        modes += [m] * nseeds
    res = EC.run_oracle_blocks_parallel(dict(args), jobs, modes=modes)
    orc = {m: res[k * nseeds:(k + 1) * nseeds] for k, m in enumerate(("f32", "f64", "f32_shuffled"))}
    for (label, lib, mode, midfit, k2), e in zip(sides_def, engines):
        apply(lib, mode, midfit, k2)
        e.update_block()
        e.sync()
    apply(LA, -1, None, None)
    sides = [(d[0], e) for d, e in zip(sides_def, engines)]
    hdr = "%-48s %-6s median     90%%       99%%       max      >1e-4  >3e-4"
    for k_net, net in ((1, "critic"), (2, "tr")):
        for ref_name, ref in (("against the float64 oracle", orc["f64"]), ("against the fp32 oracle (the parity tests' view)", orc["f32"])):
            print()
            print(hdr % (ref_name, net))
            for label, E in sides:
                e = errs_vs(ref, lambda s, i, kk, E=E, nn=net: E.get_weights(s, i, nn), nseeds, n, k_net)
                print("%-48s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((label, net) + stats(e)))
            for m, lab in (("f32", "oracle fp32"), ("f32_shuffled", "oracle fp32, fit rows reordered")):
                if ref is orc["f32"] and m == "f32":
                    continue
                e = errs_vs(ref, lambda s, i, kk, R=orc[m]: R[s][i][kk], nseeds, n, k_net)
                print("%-48s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((lab, net) + stats(e)))
        eA = errs_vs([[[None, engB.get_weights(s, i, "critic"), engB.get_weights(s, i, "tr")] for i in range(n)] for s in range(nseeds)],
                     lambda s, i, kk, nn=net: engA.get_weights(s, i, nn), nseeds, n, k_net)
        print("%-48s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % (("first side against the second side", net) + stats(eA)))


if __name__ == "__main__":          # (the oracle's worker processes import this file: nothing runs there)
    main()
