"""One-agent views over the batched kernels (S = 1, N = 1 or N = d parameter rows).

The reference's agent methods (agents/resilient_CAC_agents.py, agents/
adversarial_CAC_agents.py) operate on ONE agent's Keras models.  The drop-in
classes in ``rcmarl_amd.agents`` keep those methods; each call packs the model
weights into parameter rows, launches the very same C-ABI kernels the batched
engine uses, and unpacks the result.  This is the compatibility path (legacy
per-agent loops keep working); throughput comes from ``train_RPBCAC`` ->
``engine.RPBCACEngine``.

No CPU fallback: the ops need librcmarl_hip.so and a GPU (tests inject the hipemu
build with ``set_backend``).
"""
import math

import numpy as np
import torch

from . import capi

HID = 20
_backend = None


def pad64(n):
    return (int(n) + 63) // 64 * 64


def set_backend(lib, device):
    """Test hook: run the single-agent ops on an explicitly given C-ABI library/device."""
    global _backend
    _backend = RowOps(lib, device)
    return _backend


def get_ops():
    global _backend
    if _backend is None:
        lib = capi.load()                                   # raises when the HIP library is missing
        if not torch.cuda.is_available():
            raise capi.RcmarlError("rcmarl_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        _backend = RowOps(lib, "cuda")
    return _backend


def flat(params):
    return np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in params])


def shapes(in_dim, out_dim, hid=HID):
    return [(in_dim, hid), (hid,), (hid, hid), (hid,), (hid, out_dim), (out_dim,)]


def unflat(vec, in_dim, out_dim):
    out, o = [], 0
    for sh in shapes(in_dim, out_dim):
        n = int(np.prod(sh))
        out.append(np.array(vec[o:o + n], dtype=np.float32).reshape(sh))
        o += n
    return out


def dims_of(params):
    """(in_dim, out_dim) of a Keras-order weight list [W1,b1,W2,b2,W3,b3]."""
    W1, W3 = np.asarray(params[0]), np.asarray(params[4])
    if W1.shape[1] != HID or np.asarray(params[2]).shape != (HID, HID):
        raise capi.RcmarlError("the compiled kernels support two hidden layers of width %d (got %s)" % (HID, W1.shape))
    return int(W1.shape[0]), int(W3.shape[1])


def as_rows(x):
    """[B, N, k] (or already flat [B, in]) -> contiguous fp32 [B, in] (Keras Flatten)."""
    x = np.asarray(x, dtype=np.float32)
    return np.ascontiguousarray(x.reshape(x.shape[0], -1))


class RowOps:
    def __init__(self, lib, device):
        self.lib = lib
        self.dev = torch.device(device)
        self._mb_flags = torch.zeros(8, dtype=torch.int32, device=self.dev)      # out-of-range flags of rcmarl_minibatch_fit (one network)

    # ---- plumbing ----------------------------------------------------------------------------
    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream if self.dev.type == "cuda" else None

    def _t(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)

    def _zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.dev)

    def _rows(self, params_list, ldp):
        th = np.zeros((1, len(params_list), ldp), np.float32)
        for n, p in enumerate(params_list):
            v = flat(p)
            th[0, n, :v.size] = v
        return self._t(th)

    def _host(self, t):
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        return t.detach().cpu().numpy()

    def _layer1(self, x, B, in_dim, theta, a1t, N, ldp, ldb):
        self.lib.rcmarl_layer1_forward(x.data_ptr(), B * in_dim, theta.data_ptr(), a1t.data_ptr(), 1, N, B, in_dim, HID,
                                       ldp, ldb, self.stream)

    # ---- forward passes ----------------------------------------------------------------------
    def value(self, params, x):
        """model(x) for a linear head of width 1 -> [B, 1]   (Keras __call__, agents/...:95-97,114)."""
        in_dim, out_dim = dims_of(params)
        assert out_dim == 1
        x = as_rows(x)
        B = x.shape[0]
        ldp, ldb = pad64(flat(params).size), pad64(B)
        th, xd = self._rows([params], ldp), self._t(x)
        a1t, out = self._zeros(1, HID, ldb), self._zeros(1, 1, ldb)
        self._layer1(xd, B, in_dim, th, a1t, 1, ldp, ldb)
        self.lib.rcmarl_mid_value(a1t.data_ptr(), th.data_ptr(), None, 0.0, out.data_ptr(), 1, 1, B, in_dim, HID, ldp, ldb,
                                  self.stream)
        return self._host(out)[0, 0, :B].reshape(B, 1).copy()

    def policy(self, params, x):
        """actor.predict(x) -> [B, n_actions] softmax probabilities (agents/...:215), row by row."""
        in_dim, A = dims_of(params)
        x = as_rows(x)
        B = x.shape[0]
        ldp = pad64(flat(params).size)
        th = self._rows([params], ldp)
        out = np.zeros((B, A), np.float32)
        probs = self._zeros(1, 1, A)
        for b in range(B):
            xd = self._t(x[b:b + 1])
            self.lib.rcmarl_policy_probs(xd.data_ptr(), th.data_ptr(), probs.data_ptr(), 1, 1, in_dim, HID, A, ldp, self.stream)
            out[b] = self._host(probs)[0, 0]
        return out

    # ---- A5/A6: 5 full-batch SGD steps on a copy (critic.fit / TR.fit, agents/...:118,136) ----
    def fit_full_batch(self, params, x, r, lr, steps=5, bootstrap_x=None, gamma=0.0):
        """Returns (fitted weight list, first-step loss).  With ``bootstrap_x`` the target is
        r + gamma*model(bootstrap_x) computed once from the pre-fit weights (:114-115)."""
        in_dim, out_dim = dims_of(params)
        assert out_dim == 1
        x = as_rows(x)
        B = x.shape[0]
        ldp, ldb = pad64(flat(params).size), pad64(B)
        L = self.lib
        th, xd = self._rows([params], ldp), self._t(x)
        msg = th.clone()
        a1t = self._zeros(1, HID, ldb)
        rr = np.zeros((1, 1, ldb), np.float32)
        rr[0, 0, :B] = np.asarray(r, dtype=np.float32).reshape(B)
        y = self._t(rr)
        if bootstrap_x is not None:
            nx = self._t(as_rows(bootstrap_x))
            self._layer1(nx, B, in_dim, th, a1t, 1, ldp, ldb)
            tgt = self._zeros(1, 1, ldb)
            L.rcmarl_mid_value(a1t.data_ptr(), th.data_ptr(), y.data_ptr(), float(gamma), tgt.data_ptr(), 1, 1, B, in_dim, HID,
                               ldp, ldb, self.stream)
            y = tgt
        nchunk = (B + L.rcmarl_rows_per_chunk() - 1) // L.rcmarl_rows_per_chunk()
        part = self._zeros(nchunk * L.rcmarl_fit_partial_size(HID))
        loss = self._zeros(1, 1)
        for st in range(steps):
            self._layer1(xd, B, in_dim, msg, a1t, 1, ldp, ldb)
            L.rcmarl_mid_fit(a1t.data_ptr(), msg.data_ptr(), y.data_ptr(), part.data_ptr(), 1, 1, B, in_dim, HID, ldp, ldb,
                             self.stream)
            L.rcmarl_small_sgd(part.data_ptr(), msg.data_ptr(), None, loss.data_ptr() if st == 0 else None, 1, 1, B, in_dim,
                               HID, ldp, float(lr), self.stream)
            L.rcmarl_layer1_backward_sgd(xd.data_ptr(), B * in_dim, a1t.data_ptr(), msg.data_ptr(), None, 1, 1, B, in_dim, HID,
                                         ldp, ldb, float(lr), self.stream)
        return unflat(self._host(msg)[0, 0], in_dim, 1), float(self._host(loss)[0, 0])

    # ---- A2: hidden-layer consensus (agents/...:142-166) ---------------------------------------
    def consensus_hidden(self, msgs, H):
        """msgs: list of d weight lists, msgs[0] = own message.  Returns the 4 aggregated hidden arrays."""
        in_dim, out_dim = dims_of(msgs[0])
        d = len(msgs)
        P = flat(msgs[0]).size
        ldp = pad64(P)
        P_hid = P - (HID * out_dim + out_dim)
        msg = self._rows(msgs, ldp)
        theta = self._zeros(1, d, ldp)
        # row 0 = the agent (own message first); rows >= 1 are placeholders and are masked out by `coop`
        nbr = self._t(np.array([[(i + k) % d for k in range(d)] for i in range(d)], np.int32))
        coop = self._t(np.array([1] + [0] * (d - 1), np.int32))
        self.lib.rcmarl_consensus_params(msg.data_ptr(), theta.data_ptr(), nbr.data_ptr(), coop.data_ptr(), 1, d, ldp, P_hid,
                                         d, int(H), None, None, self.stream)
        return unflat(self._host(theta)[0, 0], in_dim, out_dim)[:4]

    # ---- A3 (+A4 residual): consensus over estimates (agents/...:168-206) ----------------------
    def consensus_estimates(self, live, x, msgs, H):
        """live: the agent's current weight list (freshly aggregated hidden layers + own head).
        Returns agg [B, 1]."""
        in_dim, _ = dims_of(live)
        x = as_rows(x)
        B = x.shape[0]
        d = len(msgs)
        ldp, ldb = pad64(flat(live).size), pad64(B)
        L = self.lib
        theta = self._rows([live] + [live] * (d - 1), ldp)
        msg = self._rows(msgs, ldp)
        nbr = self._t(np.array([[(i + k) % d for k in range(d)] for i in range(d)], np.int32))
        coop = self._t(np.array([1] + [0] * (d - 1), np.int32))
        xd = self._t(x)
        a1t = self._zeros(1, d * HID, ldb)
        nchunk = (B + L.rcmarl_rows_per_chunk() - 1) // L.rcmarl_rows_per_chunk()
        part = self._zeros(d * nchunk * (HID + 1))
        agg = self._zeros(1, d, ldb)
        self._layer1(xd, B, in_dim, theta, a1t, d, ldp, ldb)
        L.rcmarl_consensus_head(a1t.data_ptr(), theta.data_ptr(), msg.data_ptr(), nbr.data_ptr(), coop.data_ptr(),
                                part.data_ptr(), agg.data_ptr(), 1, d, B, in_dim, HID, ldp, ldb, d, int(H), self.stream)
        return self._host(agg)[0, 0, :B].reshape(B, 1).copy()

    # ---- A4: projection ("team") update of the output layer (agents/...:60-84) -----------------
    def projection_step(self, live, x, agg):
        """Returns the new (W3, b3)."""
        in_dim, _ = dims_of(live)
        x = as_rows(x)
        B = x.shape[0]
        ldp, ldb = pad64(flat(live).size), pad64(B)
        L = self.lib
        theta, xd = self._rows([live], ldp), self._t(x)
        ag = np.zeros((1, 1, ldb), np.float32)
        ag[0, 0, :B] = np.asarray(agg, dtype=np.float32).reshape(B)
        agd = self._t(ag)
        coop = self._t(np.array([1], np.int32))
        a1t = self._zeros(1, HID, ldb)
        nchunk = (B + L.rcmarl_rows_per_chunk() - 1) // L.rcmarl_rows_per_chunk()
        part = self._zeros(nchunk * (HID + 1))
        self._layer1(xd, B, in_dim, theta, a1t, 1, ldp, ldb)
        L.rcmarl_projection_residual(a1t.data_ptr(), theta.data_ptr(), agd.data_ptr(), coop.data_ptr(), part.data_ptr(), 1, 1, B,
                                     in_dim, HID, ldp, ldb, self.stream)
        L.rcmarl_head_apply(part.data_ptr(), theta.data_ptr(), coop.data_ptr(), 1, 1, B, in_dim, HID, ldp, self.stream)
        new = unflat(self._host(theta)[0, 0], in_dim, 1)
        return new[4], new[5]

    # ---- A7: one Adam step of the actor (actor.train_on_batch, agents/...:99) -------------------
    def actor_step(self, params, adam, x, labels, weights, lr):
        """adam: dict(m=flat fp32, v=flat fp32, t=int), updated in place.  Returns (new weights, loss)."""
        in_dim, A = dims_of(params)
        x = as_rows(x)
        B = x.shape[0]
        P = flat(params).size
        ldp, ldb = pad64(P), pad64(B)
        L = self.lib
        theta, xd = self._rows([params], ldp), self._t(x)
        mm, vv = np.zeros((1, 1, ldp), np.float32), np.zeros((1, 1, ldp), np.float32)
        mm[0, 0, :P], vv[0, 0, :P] = adam["m"], adam["v"]
        m, v = self._t(mm), self._t(vv)
        lab, wgt = np.zeros((1, 1, ldb), np.float32), np.zeros((1, 1, ldb), np.float32)
        lab[0, 0, :B] = np.asarray(labels, dtype=np.float32).reshape(B)
        wgt[0, 0, :B] = np.asarray(weights, dtype=np.float32).reshape(B)
        labd, wgtd = self._t(lab), self._t(wgt)
        a1t = self._zeros(1, HID, ldb)
        nchunk = (B + L.rcmarl_rows_per_chunk() - 1) // L.rcmarl_rows_per_chunk()
        part = self._zeros(nchunk * L.rcmarl_actor_partial_size(HID, A))
        loss = self._zeros(1, 1)
        adam["t"] += 1
        b1, b2, eps = 0.9, 0.999, 1e-7
        alpha = float(np.float32(lr * math.sqrt(1.0 - b2 ** adam["t"]) / (1.0 - b1 ** adam["t"])))
        omb1, omb2, epsf = float(np.float32(1 - b1)), float(np.float32(1 - b2)), float(np.float32(eps))
        self._layer1(xd, B, in_dim, theta, a1t, 1, ldp, ldb)
        L.rcmarl_mid_actor(a1t.data_ptr(), theta.data_ptr(), labd.data_ptr(), wgtd.data_ptr(), part.data_ptr(), 1, 1, B, in_dim,
                           HID, A, ldp, ldb, self.stream)
        L.rcmarl_small_adam(part.data_ptr(), theta.data_ptr(), m.data_ptr(), v.data_ptr(), None, loss.data_ptr(), 1, 1, B,
                            in_dim, HID, A, ldp, alpha, omb1, omb2, epsf, self.stream)
        L.rcmarl_layer1_backward_adam(xd.data_ptr(), B * in_dim, a1t.data_ptr(), theta.data_ptr(), m.data_ptr(), v.data_ptr(),
                                      None, 1, 1, B, in_dim, HID, ldp, ldb, alpha, omb1, omb2, epsf, self.stream)
        adam["m"], adam["v"] = self._host(m)[0, 0, :P].copy(), self._host(v)[0, 0, :P].copy()
        return unflat(self._host(theta)[0, 0], in_dim, A), float(self._host(loss)[0, 0])

    # ---- X1: the adversaries' mini-batch fits (agents/adversarial_CAC_agents.py) ---------------
    def shuffle_perms(self, seed, call, epochs, B):
        """[epochs][B] int32 permutations of the call-th mini-batch fit (csrc/shuffle.hip)."""
        seeds = self._t(np.asarray([int(seed) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64).view(np.int64))
        calls = self._t(np.asarray([int(call)], np.int32))
        out = self._zeros(1, 1, int(epochs), int(B), dtype=torch.int32)
        self.lib.rcmarl_shuffle_perms(seeds.data_ptr(), calls.data_ptr(), 1, int(epochs), int(B), out.data_ptr(), 1, self.stream)
        return self._host(out)[0, 0]

    def minibatch_fit(self, params, x, y, lr, batch_size=32, epochs=10, perms=None):
        in_dim, out_dim = dims_of(params)
        assert out_dim == 1
        x = as_rows(x)
        B = x.shape[0]
        ldp, ldb = pad64(flat(params).size), pad64(B)
        theta, xd = self._rows([params], ldp), self._t(x)
        yy = np.zeros((1, 1, ldb), np.float32)
        yy[0, 0, :B] = np.asarray(y, dtype=np.float32).reshape(B)
        yd = self._t(yy)
        agents = self._t(np.array([0], np.int32))
        pd = None if perms is None else self._t(np.asarray(perms, np.int32).reshape(1, 1, epochs, B))
        loss = self._zeros(1, 1)
        self.lib.rcmarl_minibatch_fit(xd.data_ptr(), B * in_dim, theta.data_ptr(), agents.data_ptr(), 1, yd.data_ptr(),
                                      None if pd is None else pd.data_ptr(), 1, 1, B, in_dim, HID, ldp, ldb, int(batch_size),
                                      int(epochs), float(lr), loss.data_ptr(), self._mb_flags.data_ptr(), self.stream)
        return unflat(self._host(theta)[0, 0], in_dim, 1), float(self._host(loss)[0, 0])

    def minibatch_actor(self, params, adam, x, labels, weights, lr, batch_size=200, epochs=1, perms=None):
        in_dim, A = dims_of(params)
        x = as_rows(x)
        B = x.shape[0]
        P = flat(params).size
        ldp, ldb = pad64(P), pad64(B)
        theta, xd = self._rows([params], ldp), self._t(x)
        mm, vv = np.zeros((1, 1, ldp), np.float32), np.zeros((1, 1, ldp), np.float32)
        mm[0, 0, :P], vv[0, 0, :P] = adam["m"], adam["v"]
        m, v = self._t(mm), self._t(vv)
        lab, wgt = np.zeros((1, 1, ldb), np.float32), np.zeros((1, 1, ldb), np.float32)
        lab[0, 0, :B] = np.asarray(labels, dtype=np.float32).reshape(B)
        wgt[0, 0, :B] = np.asarray(weights, dtype=np.float32).reshape(B)
        labd, wgtd = self._t(lab), self._t(wgt)
        agents = self._t(np.array([0], np.int32))
        pd = None if perms is None else self._t(np.asarray(perms, np.int32).reshape(1, 1, epochs, B))
        loss = self._zeros(1, 1)
        self.lib.rcmarl_minibatch_actor(xd.data_ptr(), B * in_dim, theta.data_ptr(), m.data_ptr(), v.data_ptr(),
                                        agents.data_ptr(), 1, labd.data_ptr(), wgtd.data_ptr(),
                                        None if pd is None else pd.data_ptr(), 1, 1, B, in_dim, HID, A, ldp, ldb,
                                        int(batch_size), int(epochs), float(lr), 0.9, 0.999, 1e-7, int(adam["t"]),
                                        loss.data_ptr(), self.stream)
        bs = min(int(batch_size), B)
        adam["t"] += epochs * ((B + bs - 1) // bs)
        adam["m"], adam["v"] = self._host(m)[0, 0, :P].copy(), self._host(v)[0, 0, :P].copy()
        return unflat(self._host(theta)[0, 0], in_dim, A), float(self._host(loss)[0, 0])
