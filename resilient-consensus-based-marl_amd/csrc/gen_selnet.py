#!/usr/bin/env python
"""Generate csrc/selnet_generated.inc: straight-line min/max selection
networks for the resilient aggregation's two order statistics.

The aggregation rule (reference agents/resilient_CAC_agents.py:48-53) needs
only sorted[H] and sorted[D-H-1] of the D neighbour values.  For each (D, H)
this script takes Batcher's merge-exchange sorting network (Knuth TAOCP 5.2.2
Algorithm M, works for any D), prunes every half-comparator that cannot reach
either wanted output (backward liveness), verifies the result with the 0-1
principle (exhaustive up to D=16, random beyond) and emits one
`SelNet<D,H>::run` specialisation of fminf/fmaxf in SSA form, so every value
lives in a VGPR.

    python gen_selnet.py > selnet_generated.inc          (--mid: selmid_generated.inc, the circulant kernel's shared networks)

Both files are BUILD PRODUCTS (git-ignored): rcmarl_amd.build and the hipemu test build call ensure_generated(), which
writes them next to this script when they are missing or older than it (about a minute, deterministic).
"""
import contextlib
import io
import os
import itertools
import random
import sys


def merge_exchange(n):
    """Batcher merge exchange comparators (i<j): a[i]<=a[j] after each."""
    if n < 2:
        return []
    t = (n - 1).bit_length()
    net = []
    p = 1 << (t - 1)
    while p > 0:
        q, r, d = 1 << (t - 1), 0, p
        while True:
            for i in range(n - d):
                if (i & p) == r:
                    net.append((i, i + d))
            if q == p:
                break
            d, q, r = q - p, q >> 1, p
        p >>= 1
    return net


def prune(net, n, outs):
    """Backward liveness at half-comparator granularity.
    Returns list of (i, j, need_min, need_max)."""
    live = set(outs)
    kept = []
    for (i, j) in reversed(net):
        nmin, nmax = i in live, j in live
        if nmin or nmax:
            kept.append((i, j, nmin, nmax))
            live |= {i, j}
    kept.reverse()
    return kept


def run_net(kept, vals, outs):
    a = list(vals)
    for (i, j, nmin, nmax) in kept:
        lo, hi = min(a[i], a[j]), max(a[i], a[j])
        if nmin:
            a[i] = lo
        else:
            a[i] = None          # dead: must never be read again
        if nmax:
            a[j] = hi
        else:
            a[j] = None
    return [a[o] for o in outs]


def verify(kept, n, outs):
    if n <= 16:
        pats = itertools.product((0, 1), repeat=n)
    else:
        rnd = random.Random(n * 131 + outs[0])
        pats = ([rnd.randint(0, 1) for _ in range(n)] for _ in range(20000))
    for p in pats:
        s = sorted(p)
        got = run_net(kept, p, outs)
        assert got == [s[o] for o in outs], (n, outs, p)
    rnd = random.Random(5)
    for _ in range(300):
        p = [rnd.uniform(-1, 1) for _ in range(n)]
        s = sorted(p)
        assert run_net(kept, p, outs) == [s[o] for o in outs]


def n_ops(kept):
    return sum(int(a) + int(b) for (_, _, a, b) in kept)


def emit(n, H, kept, out):
    lo_i, hi_i = H, n - H - 1
    out.append(f"template <> struct SelNet<{n}, {H}> {{   // {n_ops(kept)} min/max ops")
    out.append("  static constexpr bool available = true;")
    out.append(f"  static __device__ __forceinline__ void run(const float (&v)[{n}], float& lo, float& hi) {{")
    cur = {i: f"v[{i}]" for i in range(n)}
    cnt = 0
    for (i, j, nmin, nmax) in kept:
        x, y = cur[i], cur[j]
        if nmin:
            out.append(f"    const float t{cnt} = fminf({x}, {y});")
            cur[i] = f"t{cnt}"
            cnt += 1
        if nmax:
            out.append(f"    const float t{cnt} = fmaxf({x}, {y});")
            cur[j] = f"t{cnt}"
            cnt += 1
    out.append(f"    lo = {cur[lo_i]}; hi = {cur[hi_i]};")
    out.append("  }")
    out.append("};")


def emit_mid(n, lo, cnt, kept, out):
    """SelMid<n, lo, cnt>::run(v, c): c[q] = sorted(v)[lo + q], q < cnt (pruned merge-exchange network)."""
    out.append(f"template <> struct SelMid<{n}, {lo}, {cnt}> {{   // {n_ops(kept)} min/max ops")
    out.append("  static constexpr bool available = true;")
    out.append(f"  static __device__ __forceinline__ void run(const float (&v)[{n}], float (&c)[{cnt}]) {{")
    cur = {i: f"v[{i}]" for i in range(n)}
    k = 0
    for (i, j, nmin, nmax) in kept:
        x, y = cur[i], cur[j]
        if nmin:
            out.append(f"    const float t{k} = fminf({x}, {y});")
            cur[i] = f"t{k}"
            k += 1
        if nmax:
            out.append(f"    const float t{k} = fmaxf({x}, {y});")
            cur[j] = f"t{k}"
            k += 1
    out.append("    " + " ".join(f"c[{q}] = {cur[lo + q]};" for q in range(cnt)))
    out.append("  }")
    out.append("};")


def mid_combos():
    """(d, H, G) of the circulant kernel (consensus_params.hip): the d-G+1 inputs shared by G consecutive agents,
    of which the middle G+1 order statistics (ranks H-G+1 .. H+1) are needed.  d == 2H+2, H >= G-1."""
    for d, h, g in [(4, 1, 2), (6, 2, 2), (10, 4, 4), (18, 8, 4), (34, 16, 4), (66, 32, 8)]:      # = circ_group() of the kernel
        assert d == 2 * h + 2 and h >= g - 1
        yield d, h, g


def main_mid():
    out = ["// GENERATED by gen_selnet.py --mid -- do not edit.",
           "// SelMid<M,LO,CNT>::run(v, c): c[q] = sorted(v)[LO+q] for q < CNT.",
           "template <int M, int LO, int CNT> struct SelMid { static constexpr bool available = false; };"]
    table, stats = [], {}
    for d, h, g in mid_combos():
        m, lo, cnt = d - g + 1, h - g + 1, g + 1
        outs = list(range(lo, lo + cnt))
        kept = prune(merge_exchange(m), m, outs)
        verify(kept, m, outs)
        emit_mid(m, lo, cnt, kept, out)
        table.append((d, h, g))
        stats[f"{d},{h},G{g}"] = n_ops(kept)
    out.append("#define RCMARL_CIRC_COMBOS(X) \\")
    out.append(" \\\n".join(f"  X({d}, {h}, {g})" for d, h, g in table))
    sys.stdout.write("\n".join(out) + "\n")
    sys.stderr.write("generated %d mid networks; ops of the shared selection: %s\n" % (len(table), stats))


def combos():
    for d in range(1, 21):
        for h in range(0, (d - 1) // 2 + 1):
            yield d, h
    for d, h in [(24, 8), (26, 12), (32, 8), (32, 15), (34, 16), (64, 16), (66, 32)]:
        yield d, h


def main():
    out = ["// GENERATED by gen_selnet.py -- do not edit.",
           "// SelNet<D,H>::run(v, lo, hi): lo = sorted(v)[H], hi = sorted(v)[D-H-1].",
           "template <int D, int H> struct SelNet { static constexpr bool available = false; };"]
    table = []
    for d, h in combos():
        outs = sorted({h, d - h - 1})
        kept = prune(merge_exchange(d), d, outs)
        verify(kept, d, outs)
        emit(d, h, kept, out)
        table.append((d, h))
    out.append("#define RCMARL_SELNET_COMBOS(X) \\")
    out.append(" \\\n".join(f"  X({d}, {h})" for d, h in table))
    sys.stdout.write("\n".join(out) + "\n")
    sys.stderr.write("generated %d networks; ops: %s\n" % (
        len(table), {f"{d},{h}": n_ops(prune(merge_exchange(d), d, sorted({h, d - h - 1}))) for d, h in table
                     if (d, h) in [(4, 1), (10, 4), (18, 8), (18, 1), (34, 16), (66, 32)]}))


def ensure_generated(csrc_dir=None, verbose=True):
    """Write selnet_generated.inc / selmid_generated.inc into csrc_dir unless they exist and are newer than this script."""
    here = os.path.dirname(os.path.abspath(__file__))
    csrc_dir = here if csrc_dir is None else csrc_dir
    me = os.path.getmtime(os.path.abspath(__file__))
    made = []
    for name, fn in (("selnet_generated.inc", main), ("selmid_generated.inc", main_mid)):
        path = os.path.join(csrc_dir, name)
        if os.path.exists(path) and os.path.getmtime(path) >= me:
            continue
        if verbose:
            sys.stderr.write("generating %s (selection networks, verified by the 0-1 principle) ...\n" % name)
        out, err = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
            fn()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            f.write(out.getvalue())
        os.replace(tmp, path)                  # atomic: concurrent builds (pytest workers) never see half a file
        made.append(name)
    return made


if __name__ == "__main__":
    main_mid() if "--mid" in sys.argv else main()
