#!/usr/bin/env python
"""Which team-reward networks does the two-piece FORWARD operand form move, and what do they look like?  (round 6, visit r)
One update block at BASELINE configs[3]'s steady state from one state, RCMARL_LAT_F16 = 3 (both GEMMs in f16 pieces) against 2
(forward in the exact three-piece form, which profiles/r06q_* shows equal to the exact form against float64).  No oracle: GPU only.
    python tools/diag_cfg4_forward_form.py [n_seeds=2] [blocks_before=2]"""
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
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    blocks_before = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n, d = 256, 18
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=8, n_episodes=0, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000,
                        seed=1000, in_nodes=in_nodes, fast_lr=0.001, slow_lr=0.002)
    seeds = tuple(1000 + k for k in range(nseeds))
    L.rcmarl_lattice_set_f16_mode(3)
    engA, _ = EC.check_block_from_injected_state(args, 32, 32, "cuda", L, seeds, blocks_before=blocks_before, oracle_later=True)
    sd = engA.state_dict()
    before = [[engA.get_weights(s, i, "tr") for i in range(n)] for s in range(nseeds)]
    engB, _ = EC.check_block_from_injected_state(args, 32, 32, "cuda", L, seeds, blocks_before=0, oracle_later=True)
    engB.load_state_dict(sd)
    # epoch by epoch: the fitted copies (the messages) right before the consensus step, and the live nets right after it
    trace = {}

    def spy(eng, tag):
        orig = eng._consensus

        def wrapped(net, xkey, B):
            if net == "tr":
                eng.sync()
                trace.setdefault(tag, []).append([eng.msg["tr"].detach().cpu().numpy().copy(), None])
            r = orig(net, xkey, B)
            if net == "tr":
                eng.sync()
                trace[tag][-1][1] = eng.theta["tr"].detach().cpu().numpy().copy()
            return r
        eng._consensus = wrapped

    spy(engA, "A")
    spy(engB, "B")
    os.environ["RCMARL_GRAPH"] = "0"
    L.rcmarl_lattice_set_f16_mode(3)
    engA.update_block(); engA.sync()
    L.rcmarl_lattice_set_f16_mode(2)
    engB.update_block(); engB.sync()
    L.rcmarl_lattice_set_f16_mode(-1)
    print("epoch by epoch, f16-piece forward against three-piece forward from the same state (|difference| / max(1, |w|max) per net, flat parameters):")
    for ep, ((mA, tA), (mB, tB)) in enumerate(zip(trace["A"], trace["B"])):
        for what, a, b in (("fitted copies (messages)", mA, mB), ("live nets after consensus", tA, tB)):
            sc = np.maximum(1.0, np.abs(b).max(axis=2))
            e = np.abs(a - b).max(axis=2) / sc                        # [S][N]
            top = np.dstack(np.unravel_index(np.argsort(e, axis=None)[::-1][:4], e.shape))[0]
            print("  epoch %d  %-26s max %.2e  nets beyond 1e-6: %4d  1e-5: %4d  1e-4: %4d | largest: %s"
                  % (ep, what, e.max(), (e > 1e-6).sum(), (e > 1e-5).sum(), (e > 1e-4).sum(),
                     ", ".join("seed %d agent %d %.1e" % (s_, i_, e[s_, i_]) for s_, i_ in top)))
    names = ("W1", "b1", "W2", "b2", "W3", "b3")
    rows = []
    for s in range(nseeds):
        for i in range(n):
            a, b = engA.get_weights(s, i, "tr"), engB.get_weights(s, i, "tr")
            errs = [float(np.abs(x - y).max()) / max(1.0, float(np.abs(y).max())) for x, y in zip(a, b)]
            rows.append((max(errs), s, i, errs))
    rows.sort(reverse=True)
    e = np.array([r[0] for r in rows])
    print("f16-piece forward against three-piece forward, %d team-reward nets: median %.2e  90%% %.2e  99%% %.2e  max %.2e | beyond 1e-4: %d, beyond 3e-4: %d"
          % (len(e), np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max(), (e > 1e-4).sum(), (e > 3e-4).sum()))
    allW1 = np.array([np.abs(before[s][i][0]).max() for s in range(nseeds) for i in range(n)])
    print("max |W1| over all nets before the block: median %.3g  99%% %.3g  max %.3g" % (np.median(allW1), np.quantile(allW1, 0.99), allW1.max()))
    print("the 16 nets that moved most (state BEFORE the block):")
    for err, s, i, errs in rows[:16]:
        w = before[s][i]
        aft = engB.get_weights(s, i, "tr")
        print("  seed %d agent %3d err %.2e per array %s | before: max|W1| %.3g  |b1| %.3g  |W2| %.3g  |W3| %.3g  b3 %.3g | after (3-piece): max|W1| %.3g |W3| %.3g  moved by %.3g"
              % (s, i, err, " ".join("%s %.1e" % (nm, x) for nm, x in zip(names, errs)), np.abs(w[0]).max(), np.abs(w[1]).max(), np.abs(w[2]).max(),
                 np.abs(w[4]).max(), float(np.ravel(w[5])[0]), np.abs(aft[0]).max(), np.abs(aft[4]).max(), float(np.abs(aft[0] - w[0]).max())))
    print("16 nets that did not move, for contrast:")
    for err, s, i, errs in rows[-16:][::4]:
        w = before[s][i]
        aft = engB.get_weights(s, i, "tr")
        print("  seed %d agent %3d err %.2e | before: max|W1| %.3g  |b1| %.3g  |W2| %.3g  |W3| %.3g | W1 moved by %.3g"
              % (s, i, err, np.abs(w[0]).max(), np.abs(w[1]).max(), np.abs(w[2]).max(), np.abs(w[4]).max(), float(np.abs(aft[0] - w[0]).max())))


if __name__ == "__main__":
    main()
