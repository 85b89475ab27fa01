#!/usr/bin/env python
"""An fp64 ARBITER for the end-of-block weight tolerance at BASELINE configs[3] (GPU + the host's cores).

One update block at the configuration bench.py times (10 epochs, live actors, fast_lr 0.001), from IDENTICAL state on every side
(tests/engine_checks.check_block_from_injected_state: the engine's weights, Adam slots and replay rows are handed over), same seeds
everywhere:
    engine, default form (two f16 pieces + f16 mid kernel)      engine, exact form (three bf16 pieces + fp32 mid kernel)
    oracle in fp32 (the parity reference)                       oracle in fp32 with the rows of every full-batch fit reordered
    oracle in FLOAT64 (the arbiter: the same loop nest, every array a double)
and every fp32 result is measured against the float64 one: per-network worst |w - w_f64| / max(1, |w_f64|max) as a distribution over the
(seed, agent) networks of each family.  If the engine's column matches the fp32 oracle's, the engine is exactly as far from the true
arithmetic as the reference's own fp32 chain is.

    python tools/diag_cfg4_fp64_arbiter.py [blocks_before=0] [n_seeds=4] > profiles/r06_cfg4_fp64_arbiter.txt
blocks_before = 0: the first block (B = 1000), the configuration of tools/diag_cfg4_parity.py bench; 2: the steady state (B = 3000)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import engine_checks as EC  # noqa: E402
from rcmarl_amd import capi  # noqa: E402

FORMS = (("engine, default (2 x f16 pieces, f16 mid)", 3, None), ("engine, exact (3 x bf16 pieces, fp32 mid)", 0, "5"))


def set_form(L, mode, midfit):
    L.rcmarl_lattice_set_f16_mode(mode)
    os.environ.pop("RCMARL_MIDFIT", None)
    if midfit:
        os.environ["RCMARL_MIDFIT"] = midfit


def errs_vs(ref, got_fn, S, n, k):
    """per-network worst |got - ref| / max(1, |ref|max) over the arrays of network family k (1 critic, 2 team reward)"""
    out = []
    for s in range(S):
        for i in range(n):
            e = 0.0
            for a, b in zip(got_fn(s, i, k), ref[s][i][k]):
                b = np.asarray(b, np.float64)
                e = max(e, float(np.abs(np.asarray(a, np.float64) - b).max()) / max(1.0, float(np.abs(b).max())))
            out.append(e)
    return np.asarray(out)


def stats(e):
    return (np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max(), int((e > 1e-4).sum()), int((e > 3e-4).sum()))


def main():
    L = capi.load()
    blocks_before = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=0, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    seeds = tuple(1000 + k for k in range(nseeds))
    B = min(1000 * (blocks_before + 1), 3000)
    print("BASELINE configs[3], ONE update block from identical state: B = %d (%d blocks before), 10 epochs, live actors (slow_lr 0.002), "
          "fast_lr 0.001, %d seeds x 256 agents; every fp32 result against the oracle's loop nest in FLOAT64" % (B, blocks_before, nseeds),
          flush=True)
    engs, snaps = [], []
    for label, mode, midfit in FORMS:
        set_form(L, mode, midfit)
        eng, sn = EC.check_block_from_injected_state(args, 32, 32, "cuda", None, seeds, blocks_before=blocks_before, oracle_later=True)
        engs.append(eng)
        snaps.append(sn)
    same = all(np.array_equal(snaps[0][s][key], snaps[1][s][key]) for s in range(nseeds) for key in ("s", "ns", "r", "a")) and \
        all(np.array_equal(x, y) for s in range(nseeds) for i in range(n) for net in ("critic", "tr", "actor")
            for x, y in zip(snaps[0][s]["W"][i][net], snaps[1][s]["W"][i][net]))
    print("state handed over by the two engine forms identical: %s%s" % (same, "" if same else " (each form gets its own oracle runs)"), flush=True)
    groups = [0] if same else [0, 1]
    jobs, modes = [], []
    for g in groups:
        for m in ("f32", "f64", "f32_shuffled"):
            jobs += snaps[g]
            modes += [m] * nseeds
    res = EC.run_oracle_blocks_parallel(dict(args), jobs, modes=modes)
    orc = {}
    for gi, g in enumerate(groups):
        for mi, m in enumerate(("f32", "f64", "f32_shuffled")):
            k0 = (gi * 3 + mi) * nseeds
            orc[(g, m)] = res[k0:k0 + nseeds]
    for k, (label, mode, midfit) in enumerate(FORMS):
        set_form(L, mode, midfit)
        engs[k].update_block()
        engs[k].sync()
    set_form(L, -1, None)
    hdr = "%-46s %-6s median     90%%       99%%       max      >1e-4  >3e-4"
    for k_net, net in ((1, "critic"), (2, "tr")):
        print()
        print(hdr % ("against the float64 oracle", net))
        rows = {}
        for k, (label, mode, midfit) in enumerate(FORMS):
            g = 0 if same else k
            e = errs_vs(orc[(g, "f64")], lambda s, i, kk, E=engs[k], nn=net: E.get_weights(s, i, nn), nseeds, n, k_net)
            rows[label] = stats(e)
        for g in groups:
            suffix = "" if same else " [state of form %d]" % g
            for m, lab in (("f32", "oracle fp32"), ("f32_shuffled", "oracle fp32, fit rows reordered")):
                e = errs_vs(orc[(g, "f64")], lambda s, i, kk, R=orc[(g, m)]: R[s][i][kk], nseeds, n, k_net)
                rows[lab + suffix] = stats(e)
        for lab, st in rows.items():
            print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((lab, net) + st))
        ref = rows["oracle fp32" + ("" if same else " [state of form 0]")]
        for k, (label, mode, midfit) in enumerate(FORMS):
            r = rows[label]
            ref_k = ref if same else rows["oracle fp32 [state of form %d]" % k]
            print("   %-43s / oracle fp32: median x%.2f  90%% x%.2f  99%% x%.2f  max x%.2f  (>1e-4: %d vs %d of %d)"
                  % (label, r[0] / ref_k[0], r[1] / ref_k[1], r[2] / ref_k[2], r[3] / ref_k[3], r[4], ref_k[4], nseeds * n))
        print(hdr % ("against the fp32 oracle (the parity tests' view)", net))
        for k, (label, mode, midfit) in enumerate(FORMS):
            g = 0 if same else k
            e = errs_vs(orc[(g, "f32")], lambda s, i, kk, E=engs[k], nn=net: E.get_weights(s, i, nn), nseeds, n, k_net)
            print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % ((label, net) + stats(e)))
        for g in groups:
            e = errs_vs(orc[(g, "f32")], lambda s, i, kk, R=orc[(g, "f32_shuffled")]: R[s][i][kk], nseeds, n, k_net)
            print("%-46s %-6s %.2e  %.2e  %.2e  %.2e  %5d  %5d" % (("oracle fp32, fit rows reordered" + ("" if same else " [%d]" % g), net) + stats(e)))


if __name__ == "__main__":          # (the oracle's worker processes import this file: nothing runs there)
    main()
