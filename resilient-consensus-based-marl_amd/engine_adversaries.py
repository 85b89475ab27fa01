"""Adversary message generators and actor updates of the batched engine.

Replaces, for all seeds at once, what the reference does per adversarial agent in
``training/train_agents.py:107-119,149-153`` through
``agents/adversarial_CAC_agents.py``:

  Faulty    transmits its frozen critic / TR                      (:45-55)
  Greedy    fits critic / TR on its OWN reward, mini-batches of 32, 10 epochs,
            in place (no rollback) and transmits them             (:228-253)
  Malicious fits a private critic on its own reward (:137-152) and transmits a
            "compromised" critic / TR fitted on -r_coop           (:121-135, :154-165;
            train_agents.py:113-116)
  all three actor_update = TD error of their own critic, then
            fit(batch_size=200, epochs=1) with Adam               (:38-41, :111-117, :221-225)

Every Keras ``fit`` of a 20-unit network is ONE kernel launch (csrc/minibatch_fit.hip): the whole
sequence of mini-batch steps of one (seed, adversary) network runs inside one
workgroup.  A critic of any other width (BASELINE configs[4]: 512 units) takes the same
mini-batch steps through the dense per-agent GEMM entry points (rcmarl_dense_*, one step =
the eight launches of a wide local-fit step on 32 rows): correct and slow -- a Byzantine agent
beside a wide critic is what the algorithm is for, not a throughput case.  Keras shuffles with TensorFlow's RNG; here the permutations come from
a counter-based stream (csrc/shuffle.hip, generated on the device for all seeds at once; the oracle
states the same definition), consumed in the reference's call order so that oracle and engine stay in
lock step.
"""
import os

import numpy as np
import torch

HID = 20
COOP, FAULTY, GREEDY, MALICIOUS = "Cooperative", "Faulty", "Greedy", "Malicious"
FIT_BATCH, FIT_EPOCHS, ACTOR_BATCH = 32, 10, 200


class AdversaryPath:
    def __init__(self, eng):
        self.e = eng
        labels = eng.cfg.agent_label
        self.adv = [i for i, l in enumerate(labels) if l != COOP]
        self.fit = [i for i in self.adv if labels[i] in (GREEDY, MALICIOUS)]
        self.mal = [i for i in self.adv if labels[i] == MALICIOUS]
        dev = eng.dev
        i32 = dict(dtype=torch.int32, device=dev)
        self.adv_t = torch.tensor(self.adv, **i32)
        self.fit_t = torch.tensor(self.fit, **i32) if self.fit else None
        self.mal_t = torch.tensor(self.mal, **i32) if self.mal else None
        self.fit_idx = torch.tensor(self.fit, dtype=torch.long, device=dev) if self.fit else None
        self.fit_mask = torch.tensor([1 if i in self.fit else 0 for i in range(len(labels))], **i32)      # rows rcmarl_copy3d copies
        self.mal_mask = torch.tensor([1 if i in self.mal else 0 for i in range(len(labels))], **i32)
        self.mal_idx = torch.tensor(self.mal, dtype=torch.long, device=dev) if self.mal else None
        self.calls = [0] * eng.S                       # ShuffleStream.calls per seed
        self.base_dev, self._draw_bufs, self.last_draw = None, {}, 0
        self.adam_t = 0                                # Adam steps every adversary's actor has taken
        f32 = dict(dtype=torch.float32, device=dev)
        for k in ("r_own", "y_l", "y_adv", "delta_adv", "v_next_adv", "v_cur_adv"):
            eng.ybuf[k] = torch.zeros(eng.S, eng.N, eng.ldb, **f32)
        self.mode0 = torch.zeros(eng.N, **i32)
        self.a1t = torch.zeros_like(eng.a1t)            # own layer-1 scratch: phase1 may run beside the cooperative fits

    def state_dict(self):
        return {"calls": list(self.calls), "adam_t": self.adam_t}

    def load_state_dict(self, sd):
        self.calls, self.adam_t = [int(x) for x in sd["calls"]], int(sd["adam_t"])

    # -- the shuffle stream (csrc/shuffle.hip; oracle: ShuffleStream), all seeds in one launch ------------
    def _draw(self, plan, epochs, B):
        """plan: list of (agent, key) in the reference's call order -> {key: int32 tensor [S][n_key][epochs][B]}.
        Every seed makes the same sequence of fit calls, so the call numbers are shared.
        hipGraph-safe: the call counter also lives on the device (base_dev), the call numbers of a draw are formed there (constant
        offsets + base_dev) and the counter is advanced by an in-stream add, the permutation buffers are persistent -- a replayed
        epoch draws the NEXT permutations; the host counter follows through replayed()."""
        e = self.e
        dev = e.dev
        if self.base_dev is None:
            self.base_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.base_dev.fill_(int(self.calls[0]))          # (eager: the host counter is the truth, e.g. after load_state_dict)
        sig = (tuple(plan), int(epochs), int(B))
        bufs = self._draw_bufs.get(sig)
        if bufs is None:
            by_key = {}
            for pos, (agent, key) in enumerate(plan):
                by_key.setdefault(key, []).append(pos)
            bufs = {}
            for key, offs in by_key.items():
                bufs[key] = (torch.tensor(offs, dtype=torch.int32, device=dev), torch.empty(len(offs), dtype=torch.int32, device=dev),
                             torch.empty(e.S, len(offs), epochs, B, dtype=torch.int32, device=dev))
            if len(self._draw_bufs) >= 16:                    # growing replay buffer: B changes every block until steady state
                self._draw_bufs.pop(next(iter(self._draw_bufs)))
            self._draw_bufs[sig] = bufs
        out = {}
        for key, (offs, calls, perm) in bufs.items():
            torch.add(offs, self.base_dev, out=calls)
            e.lib.rcmarl_shuffle_perms(e.seeds_dev.data_ptr(), calls.data_ptr(), offs.numel(), epochs, B, perm.data_ptr(), e.S, e.stream)
            out[key] = perm
        self.base_dev.add_(len(plan))
        self.calls = [c + len(plan) for c in self.calls]
        self.last_draw = len(plan)
        return out

    def replayed(self, draws):
        """an epoch that drew `draws` permutations was replayed from a hipGraph: the device-side counter moved, the host's follows"""
        self.calls = [c + draws for c in self.calls]

    # -- Keras fit(batch_size=32, epochs=10) of a critic-family network, any width ----------------------------
    def _fit_critic_family(self, theta, agents_t, agents, y, perm, B, loss_out):
        """theta [S][N][ldp] (fitted in place, rows `agents`), targets y [S][N][ldb], perm int32 [S][len(agents)][epochs][B]
        (agents/adversarial_CAC_agents.py:131-135,146-152,237-241)."""
        e, L = self.e, self.e.lib
        hid = e.hid["critic"]
        xptr, xstride = e._x("s")
        if hid == HID:
            L.rcmarl_minibatch_fit(xptr, xstride, theta.data_ptr(), agents_t.data_ptr(), len(agents), y.data_ptr(),
                                   perm.data_ptr(), e.S, e.N, B, e.in_c, HID, e.ldp["critic"], e.ldb, FIT_BATCH, FIT_EPOCHS,
                                   e.cfg.fast_lr, None if loss_out is None else loss_out.data_ptr(),
                                   e.ovf_flags(("adv", theta.data_ptr()), e.S * len(agents)).data_ptr(), e.stream)
            return
        # wide critic: one SGD step = forward L1, forward L2, head fit (dz3, dz2 in place, gW3/gb3/gb2), backward-data L2
        # (-> dz1), bias grad, backward-SGD W2, backward-SGD W1, small SGD -- every gradient from the pre-step weights, as
        # RPBCACEngine._local_fit_wide does for a full batch; here on (seed, agent) = (1, 1) views and 32 permuted rows
        in_dim, ldp, lr, st = e.in_c, e.ldp["critic"], e.cfg.fast_lr, e.stream
        o_b1 = in_dim * hid
        o_W2 = o_b1 + hid
        o_b2 = o_W2 + hid * hid
        ldm = 64                                        # row stride of the mini-batch scratch (>= FIT_BATCH, % 64 == 0)
        if getattr(self, "_mb", None) is None or self._mb["hid"] != hid:
            f32 = dict(dtype=torch.float32, device=e.dev)
            self._mb = {"hid": hid, "a1": torch.zeros(hid, ldm, **f32), "a2": torch.zeros(hid, ldm, **f32),
                        "dz1": torch.zeros(hid, ldm, **f32), "dz3": torch.zeros(ldm, **f32),
                        "grads": torch.zeros(L.rcmarl_wide_grad_size(hid), **f32), "lp": torch.zeros(4, **f32),
                        "loss": torch.zeros(1, **f32), "one": torch.ones(1, dtype=torch.int32, device=e.dev)}
        m = self._mb
        a1, a2, dz1, dz3, grads, lp, one = (m[k].data_ptr() for k in ("a1", "a2", "dz1", "dz3", "grads", "lp", "one"))
        X = e.rp["s"]
        for s_ in range(e.S):
            for q, ag in enumerate(agents):
                th = theta.data_ptr() + 4 * ((s_ * e.N + ag) * ldp)
                first = torch.zeros(1, dtype=torch.float32, device=e.dev)
                for ep in range(FIT_EPOCHS):
                    idx = perm[s_, q, ep].long()
                    Xp = X[s_, :B].index_select(0, idx).contiguous()              # [B][in_dim], batches are row slices now
                    yp = y[s_, ag, :B].index_select(0, idx).contiguous()
                    for lo in range(0, B, FIT_BATCH):
                        nb = min(FIT_BATCH, B - lo)
                        xb, yb = Xp.data_ptr() + 4 * lo * in_dim, yp.data_ptr() + 4 * lo
                        L.rcmarl_dense_forward(xb, 0, 0, 1, in_dim, th, 0, o_b1, a1, 1, 1, nb, in_dim, hid, ldp, ldm, st)
                        L.rcmarl_dense_forward(a1, hid * ldm, hid * ldm, 0, ldm, th, o_W2, o_b2, a2, 1, 1, nb, hid, hid, ldp, ldm, st)
                        L.rcmarl_wide_head_fit(a2, th, yb, dz3, grads, lp, 1, 1, nb, in_dim, hid, ldp, ldm, st)     # a2 <- dz2
                        L.rcmarl_dense_backward_data(a2, th, o_W2, a1, dz1, 1, 1, nb, hid, hid, ldp, ldm, st)
                        L.rcmarl_wide_bias_grad(dz1, grads, 1, 1, nb, hid, ldm, st)
                        L.rcmarl_dense_backward_sgd(a1, hid * ldm, hid * ldm, 0, ldm, a2, th, o_W2, one, 1, 1, nb, hid, hid, ldp,
                                                    ldm, lr, st)
                        L.rcmarl_dense_backward_sgd(xb, 0, 0, 1, in_dim, dz1, th, 0, one, 1, 1, nb, in_dim, hid, ldp, ldm, lr, st)
                        L.rcmarl_wide_small_sgd(grads, lp, th, one, m["loss"].data_ptr() if ep == 0 and loss_out is not None else None,
                                                1, 1, nb, in_dim, hid, ldp, lr, st)
                        if ep == 0 and loss_out is not None:
                            first += m["loss"] * float(nb)            # Keras History: sample-weighted mean of the batch losses
                if loss_out is not None:
                    loss_out[s_, ag] = first[0] / float(B)

    # -- phase I of every consensus epoch ----------------------------------------------------
    def phase1(self, B):
        """One consensus epoch's message generators (targets of the transmitted critic: the engine's y_c rows)."""
        e, L = self.e, self.e.lib
        if not self.fit:
            return                                     # only Faulty agents: msg rows already = frozen theta rows
        y_c = e.ybuf["y_c"]
        labels = e.cfg.agent_label
        plan = []
        for i in self.fit:                             # train_agents.py:105-119, agent index order
            if labels[i] == MALICIOUS:
                plan.append((i, "local"))
            plan.append((i, "tr"))
            plan.append((i, "critic"))
        perms = self._draw(plan, FIT_EPOCHS, B)
        S, N = e.S, e.N
        if self._multi_ok():
            return self._phase1_multi(B, perms)
        # The (up to) three fits are independent networks and each is ONE latency-bound workgroup per (seed, adversary):
        # on a GPU they run side by side on three streams (forked from / joined to the current one).
        par = e.dev.type == "cuda" and not e.wide and os.environ.get("RCMARL_ADV_ASYNC", "1") not in ("0", "false")
        cur = torch.cuda.current_stream() if par else None
        if par and not hasattr(self, "fit_streams"):
            self.fit_streams = [torch.cuda.Stream(device=e.dev) for _ in range(2)]
        fork = None
        if par:
            fork = torch.cuda.Event()
            fork.record(cur)
        joins = []

        def on(idx):
            """context for fit #idx: stream idx-1 of the pool for idx >= 1, the current stream for idx == 0"""
            import contextlib
            if not par or idx == 0:
                return contextlib.nullcontext()
            st = self.fit_streams[idx - 1]
            st.wait_event(fork)
            return torch.cuda.stream(st)

        def done(idx):
            if par and idx > 0:
                ev = torch.cuda.Event()
                ev.record(self.fit_streams[idx - 1])
                joins.append(ev)
        if self.mal:                                   # private critic: own reward, own bootstrap (:137-152)
            with on(2):
                self._local_targets(B)
                self._fit_critic_family(e.theta["critic_local"], self.mal_t, self.mal, e.ybuf["y_l"], perms["local"], B, None)
                done(2)
        # transmitted TR: targets r_fit (own reward for Greedy, -r_coop for Malicious)
        with on(1):
            xptr, xstride = e._x("sa")
            L.rcmarl_minibatch_fit(xptr, xstride, e.theta["tr"].data_ptr(), self.fit_t.data_ptr(), len(self.fit),
                                   e.ybuf["r_fit"].data_ptr(), perms["tr"].data_ptr(), S, N, B, e.in_r, HID, e.ldp["tr"],
                                   e.ldb, FIT_BATCH, FIT_EPOCHS, e.cfg.fast_lr, e.loss["tr"].data_ptr(),
                                   e.ovf_flags(("adv", "tr"), S * len(self.fit)).data_ptr(), e.stream)
            self._publish("tr")                        # the fitted net IS the message
            done(1)
        # transmitted critic: targets y_c = r_fit + gamma*V_theta(ns), computed from the pre-fit weights
        self._fit_critic_family(e.theta["critic"], self.fit_t, self.fit, y_c, perms["critic"], B, e.loss["critic"])
        self._publish("critic")
        for ev in joins:
            cur.wait_event(ev)

    def _local_targets(self, B):
        """targets of the Malicious agents' private critic: own reward + gamma V_local(ns) (:137-152)"""
        e, L = self.e, self.e.lib
        rptr, rstride = e._x("r")
        L.rcmarl_gather_agent_major(rptr, rstride, None, None, e.ybuf["r_own"].data_ptr(), e.S, e.N, B, e.ldb, e.stream)
        e._value("ns", e.theta["critic_local"], "critic", e.ybuf["y_l"], B, r_applied=e.ybuf["r_own"], scratch=self.a1t)

    def _publish(self, net):
        """msg[net] rows of the fitting adversaries <- their theta rows (one masked strided copy, no framework kernels)"""
        e = self.e
        ldp = e.ldp[net]
        e.lib.rcmarl_copy3d(e.theta[net].data_ptr(), e.N * ldp, ldp, e.msg[net].data_ptr(), e.N * ldp, ldp, e.S, e.N, ldp,
                            self.fit_mask.data_ptr(), e.stream)

    def _multi_ok(self):
        """all fits of an epoch as ONE rcmarl_minibatch_fit_multi launch?  20-unit networks of at most 20 inputs in one input class
        (the reference's own 5-agent scenarios); RCMARL_ADV_MULTI=0: the per-family launches (on side streams on a GPU)."""
        e = self.e
        if e.wide or os.environ.get("RCMARL_ADV_MULTI", "1") in ("0", "false") or not hasattr(e.lib, "rcmarl_minibatch_fit_multi"):
            return False
        return max(e.in_c, e.in_r) <= 20 and (e.in_c <= 16) == (e.in_r <= 16)

    def _phase1_multi(self, B, perms):
        from . import capi
        e, L = self.e, self.e.lib
        S, N = e.S, e.N
        jobs = []

        def job(xkey, theta, agents_t, n, y, perm, in_dim, ldp, loss, flag_key):
            xptr, xstride = e._x(xkey)
            jobs.append(capi.MbJob(xptr, xstride, theta.data_ptr(), agents_t.data_ptr(), n, in_dim, ldp, 0, y.data_ptr(), perm.data_ptr(),
                                   None if loss is None else loss.data_ptr(), e.ovf_flags(("adv", flag_key), S * n).data_ptr()))
        if self.mal:
            self._local_targets(B)
            job("s", e.theta["critic_local"], self.mal_t, len(self.mal), e.ybuf["y_l"], perms["local"], e.in_c, e.ldp["critic"], None,
                "local")
        job("sa", e.theta["tr"], self.fit_t, len(self.fit), e.ybuf["r_fit"], perms["tr"], e.in_r, e.ldp["tr"], e.loss["tr"], "tr")
        job("s", e.theta["critic"], self.fit_t, len(self.fit), e.ybuf["y_c"], perms["critic"], e.in_c, e.ldp["critic"], e.loss["critic"],
            "critic")
        arr = (capi.MbJob * len(jobs))(*jobs)
        L.rcmarl_minibatch_fit_multi(arr, len(jobs), S, N, B, HID, e.ldb, FIT_BATCH, FIT_EPOCHS, e.cfg.fast_lr, e.stream)
        self._publish("tr")
        self._publish("critic")

    # -- phase III ---------------------------------------------------------------------------
    def actor_updates(self, B):
        e, L = self.e, self.e.lib
        S, N, nl = e.S, e.N, e.n_last
        row0 = B - nl
        shuffle = nl > ACTOR_BATCH
        perms = self._draw([(i, "actor") for i in self.adv], 1, nl)["actor"] if shuffle else None
        own = e.theta["critic"]
        if self.mal:                                   # a Malicious agent's TD error comes from its PRIVATE critic (:111-115)
            if getattr(self, "_own", None) is None or self._own.shape != e.theta["critic"].shape or self._own.device != e.theta["critic"].device:
                self._own = torch.empty_like(e.theta["critic"])      # (re-made if the parameter matrix was re-allocated in another shape)
            own, ldp = self._own, e.ldp["critic"]
            L.rcmarl_copy3d(e.theta["critic"].data_ptr(), N * ldp, ldp, own.data_ptr(), N * ldp, ldp, S, N, ldp, None, e.stream)
            L.rcmarl_copy3d(e.theta["critic_local"].data_ptr(), N * ldp, ldp, own.data_ptr(), N * ldp, ldp, S, N, ldp,
                            self.mal_mask.data_ptr(), e.stream)
        rptr, rstride = e._x("r", row0)
        L.rcmarl_gather_agent_major(rptr, rstride, None, None, e.ybuf["r_own"].data_ptr(), S, N, nl, e.ldb, e.stream)
        e._value("ns", own, "critic", e.ybuf["v_next_adv"], nl, row0, value_f32=True)
        e._value("s", own, "critic", e.ybuf["v_cur_adv"], nl, row0, value_f32=True)
        L.rcmarl_td_error(e.ybuf["r_own"].data_ptr(), e.ybuf["v_next_adv"].data_ptr(), e.ybuf["v_cur_adv"].data_ptr(),
                          e.cfg.gamma, e.ybuf["delta_adv"].data_ptr(), S * N * e.ldb, e.stream)
        sptr, sstride = e._x("s", row0)
        L.rcmarl_minibatch_actor(sptr, sstride, e.theta["actor"].data_ptr(), e.adam_m.data_ptr(), e.adam_v.data_ptr(),
                                 self.adv_t.data_ptr(), len(self.adv), e.ybuf["act_t"].data_ptr(),
                                 e.ybuf["delta_adv"].data_ptr(), None if perms is None else perms.data_ptr(), S, N, nl,
                                 e.in_c, HID, e.cfg.n_actions, e.ldp["actor"], e.ldb, ACTOR_BATCH, 1, e.cfg.slow_lr,
                                 0.9, 0.999, 1e-7, self.adam_t, e.loss["actor"].data_ptr(), e.stream)
        self.adam_t += (nl + ACTOR_BATCH - 1) // ACTOR_BATCH


def attach(eng):
    eng.adv = AdversaryPath(eng)
