"""Per-kernel HIP-event timing of C-ABI launches (used by bench.py).

Wraps a capi.CLib: every launch is bracketed by two events recorded on the
stream the kernel is launched on (torch's current stream), so durations can
be read after the timed region without synchronising inside it."""
import collections

import torch

from . import capi


def _gemm_flops(a, first):          # (..., S, N, B, in_dim, hid, ...) starting at index `first`
    S, N, B, in_dim, hid = a[first:first + 5]
    return 2.0 * S * N * hid * B * in_dim, 0.0


_LIB = None          # the library the most recent TimedLib wraps (an injected / emulated one included)


def lattice_pieces(bit, lib=None):
    """16-bit pieces per value of the forward (bit 1) / backward (bit 2) lattice operand under the operand form of `lib` (default:
    the library the current TimedLib wraps, else the product library; RCMARL_LAT_F16, csrc/rcmarl_lattice.h: default 3 = two
    f16 pieces for both)."""
    lib = lib or _LIB or capi.load()
    return 2 if int(lib.rcmarl_lattice_f16_mode()) & bit else 3


# algorithmic work per launch: name -> f(args) -> (flops, bytes)   (DESIGN.md "Kernels")
WORK = {
    "rcmarl_layer1_forward": lambda a: _gemm_flops(a, 4),
    "rcmarl_layer1_backward_sgd": lambda a: _gemm_flops(a, 5),
    "rcmarl_layer1_backward_adam": lambda a: _gemm_flops(a, 7),
    # msg, theta, nbr, coop, S, N, ldp, P_hid, ...: read P_hid + write P_hid floats per (seed, agent)
    "rcmarl_consensus_params": lambda a: (0.0, 8.0 * a[4] * a[5] * a[7]),
    # msg, theta, coop, S, N, ldp, P_hid, ...
    "rcmarl_consensus_params_circulant": lambda a: (0.0, 8.0 * a[3] * a[4] * a[6]),
    # a1t, theta, y, partials, S, N, B, in_dim, hid: layers 2-3 fwd+bwd ~ (8 h^2 + 12 h) flops per (row, agent)
    "rcmarl_mid_fit": lambda a: (a[4] * a[5] * a[6] * (8.0 * a[8] ** 2 + 12.0 * a[8]), 8.0 * a[4] * a[5] * a[6] * a[8]),
    # lattice path: fp32-EQUIVALENT flops 2*M*N*K (the kernel executes 2x that on the f16 matrix core, 3x in the bf16 form)
    "rcmarl_layer1_forward_lattice": lambda a: _gemm_flops(a, 8),
    "rcmarl_layer1_backward_sgd_lattice": lambda a: _gemm_flops(a, 9),
    # a1t, theta, y, partials, dzp, rt, kt, S, N, B, in_dim, hid: reads a1 (4 B), writes the 16-bit pieces of dz1 (2 x 2 B, or 3 x 2 B)
    "rcmarl_mid_fit_lattice": lambda a: (a[7] * a[8] * a[9] * (8.0 * a[11] ** 2 + 12.0 * a[11]),
                                         (4.0 + 2.0 * lattice_pieces(2)) * a[7] * a[8] * a[9] * a[11]),
    # theta, alpha, wp, S, N, in_dim, hid: reads W1 (4 B), writes its pieces (2 x 2 B, or 3 x 2 B)
    "rcmarl_w1_split": lambda a: (0.0, (4.0 + 2.0 * lattice_pieces(1)) * a[3] * a[4] * a[5] * a[6]),
    # x, x_seed_stride, theta, agents, n_adv, y, perm, S, N, B, in_dim, hid, ..., batch_size, epochs: whole Keras fit() of
    # n_adv networks per seed: epochs x B rows x (forward + backward ~ 6 flops per weight)
    "rcmarl_minibatch_fit": lambda a: (6.0 * a[7] * a[4] * a[15] * a[9] * (a[10] * a[11] + a[11] * a[11] + a[11]), 0.0),
    # jobs (host array of capi.MbJob), njobs, S, N, B, hid, ldb, batch_size, epochs: the same count summed over the jobs
    "rcmarl_minibatch_fit_multi": lambda a: (sum(6.0 * a[2] * j.n_adv * a[8] * a[4] * (j.in_dim * a[5] + a[5] * a[5] + a[5])
                                                 for j in list(a[0])[:a[1]]) if hasattr(a[0], "__len__") else 0.0, 0.0),
    # wide path (csrc/wide_kernels.hip): dense per-agent GEMMs, 2*S*N*B*K*J flops
    # in, zs, za, rm, ld, theta, w_off, b_off, out, S, N, B, K, J
    "rcmarl_dense_forward": lambda a: (2.0 * a[9] * a[10] * a[11] * a[12] * a[13], 0.0),
    # dz_out, theta, w_off, act_in, dz_in, S, N, B, K, J
    "rcmarl_dense_backward_data": lambda a: (2.0 * a[5] * a[6] * a[7] * a[8] * a[9], 0.0),
    # in, zs, za, rm, ld, dz, theta, w_off, mask, S, N, B, K, J
    "rcmarl_dense_backward_sgd": lambda a: (2.0 * a[9] * a[10] * a[11] * a[12] * a[13], 0.0),
    # the same layers on pre-split packed operands (csrc/dense_pk.hip): fp32-equivalent 2*S*N*B*hid*hid (layer 1: the lattice GEMM)
    # kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta, a1_bk, bk_rt, a1_kb, kb_kt, s1, s1_ld, ovf_flag, S, N, B, in_dim, hid
    "rcmarl_layer1_forward_lattice_pk": lambda a: _gemm_flops(a, 14),
    # w2t, a1_bk, bk_rt, theta, a2, mask_bj, mbj_rt, mask_jb, mjb_kt, vpart, npart, S, N, B, in_dim, hid
    "rcmarl_pk_forward2": lambda a: (2.0 * a[11] * a[12] * a[13] * a[15] * a[15], 0.0),
    # mask_bj, mbj_rt, w2w3, rs, s1, s1_ld, dz3, dzp, dzp_rt, dzp_kt, gb1part, ovf_flag, S, N, B, hid
    "rcmarl_pk_backward_data": lambda a: (2.0 * a[12] * a[13] * a[14] * a[15] * a[15], 0.0),
    # a1_kb, kb_kt, mask_jb, mjb_kt, dzv, theta, mask, gw3part, q, S, N, B, in_dim, hid
    "rcmarl_pk_backward_w2": lambda a: (2.0 * a[9] * a[10] * a[11] * a[13] * a[13], 0.0),
}


class TimedLib:
    def __init__(self, lib):
        global _LIB
        self._lib = _LIB = lib
        self.enabled = False
        self._events = collections.defaultdict(list)
        self.work = {}
        for name in capi.SIGNATURES:
            fn = getattr(lib, name)
            setattr(self, name, fn if name in capi.UNCHECKED else self._wrap(name, fn))

    def _wrap(self, name, fn):
        def call(*args):
            if not self.enabled:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args)
            e1.record()
            self._events[name].append((e0, e1))
            if name in WORK:
                f, b = WORK[name](args)
                pf, pb = self.work.get(name, (0.0, 0.0))
                self.work[name] = (pf + f, pb + b)
            return r
        return call

    def reset(self):
        self._events.clear()
        self.work = {}

    def summary(self):
        """name -> (launches, total_ms, avg_us); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self._events.items():
            tot = sum(a.elapsed_time(b) for a, b in evs)
            out[name] = (len(evs), tot, 1e3 * tot / max(len(evs), 1))
        return out
