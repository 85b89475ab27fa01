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

Every Keras ``fit`` is ONE kernel launch (csrc/minibatch_fit.hip): the whole
sequence of mini-batch steps of one (seed, adversary) network runs inside one
workgroup.  Keras shuffles with TensorFlow's RNG; here the permutations come from
a counter-based stream (csrc/shuffle.hip, generated on the device for all seeds at once; the oracle
states the same definition), consumed in the reference's call order so that oracle and engine stay in
lock step.
"""
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
        self.mal_idx = torch.tensor(self.mal, dtype=torch.long, device=dev) if self.mal else None
        self.calls = [0] * eng.S                       # ShuffleStream.calls per seed
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
        Every seed makes the same sequence of fit calls, so the call numbers are shared."""
        e = self.e
        base = self.calls[0]
        by_key = {}
        for pos, (agent, key) in enumerate(plan):
            by_key.setdefault(key, []).append(base + pos)
        out = {}
        for key, call_ids in by_key.items():
            calls = torch.tensor(call_ids, dtype=torch.int32, device=e.dev)
            perm = torch.empty(e.S, len(call_ids), epochs, B, dtype=torch.int32, device=e.dev)
            e.lib.rcmarl_shuffle_perms(e.seeds_dev.data_ptr(), calls.data_ptr(), len(call_ids), epochs, B, perm.data_ptr(),
                                       e.S, e.stream)
            out[key] = perm
            self._keep = (calls, perm)                 # keep the small call table alive until the launch has run
        self.calls = [c + len(plan) for c in self.calls]
        return out

    # -- phase I of every consensus epoch ----------------------------------------------------
    def phase1(self, B, y_c=None, write_msg=True):
        """One consensus epoch's message generators.  y_c: the transmitted critic's targets (default: the engine's y_c rows);
        write_msg=False leaves engine.msg alone (chain_async hands the rows over epoch by epoch instead)."""
        e, L = self.e, self.e.lib
        if not self.fit:
            return                                     # only Faulty agents: msg rows already = frozen theta rows
        y_c = e.ybuf["y_c"] if y_c is None else y_c
        labels = e.cfg.agent_label
        plan = []
        for i in self.fit:                             # train_agents.py:105-119, agent index order
            if labels[i] == MALICIOUS:
                plan.append((i, "local"))
            plan.append((i, "tr"))
            plan.append((i, "critic"))
        perms = self._draw(plan, FIT_EPOCHS, B)
        self._keep_perms = perms                       # other streams read them: keep the memory until the next call
        S, N = e.S, e.N
        # The (up to) three fits are independent networks and each is ONE latency-bound workgroup per (seed, adversary):
        # on a GPU they run side by side on three streams (forked from / joined to the current one).
        par = e.dev.type == "cuda" and __import__("os").environ.get("RCMARL_ADV_ASYNC", "1") not in ("0", "false")
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
                rptr, rstride = e._x("r")
                L.rcmarl_gather_agent_major(rptr, rstride, None, None, e.ybuf["r_own"].data_ptr(), S, N, B, e.ldb, e.stream)
                e._value("ns", e.theta["critic_local"], "critic", e.ybuf["y_l"], B, r_applied=e.ybuf["r_own"], scratch=self.a1t)
                xptr, xstride = e._x("s")
                L.rcmarl_minibatch_fit(xptr, xstride, e.theta["critic_local"].data_ptr(), self.mal_t.data_ptr(), len(self.mal),
                                       e.ybuf["y_l"].data_ptr(), perms["local"].data_ptr(), S, N, B, e.in_c, HID,
                                       e.ldp["critic"], e.ldb, FIT_BATCH, FIT_EPOCHS, e.cfg.fast_lr, None, e.stream)
                done(2)
        # transmitted TR: targets r_fit (own reward for Greedy, -r_coop for Malicious)
        with on(1):
            xptr, xstride = e._x("sa")
            L.rcmarl_minibatch_fit(xptr, xstride, e.theta["tr"].data_ptr(), self.fit_t.data_ptr(), len(self.fit),
                                   e.ybuf["r_fit"].data_ptr(), perms["tr"].data_ptr(), S, N, B, e.in_r, HID, e.ldp["tr"],
                                   e.ldb, FIT_BATCH, FIT_EPOCHS, e.cfg.fast_lr, e.loss["tr"].data_ptr(), e.stream)
            if write_msg:
                e.msg["tr"].index_copy_(1, self.fit_idx, e.theta["tr"].index_select(1, self.fit_idx))   # the fitted net IS the message
            done(1)
        # transmitted critic: targets y_c = r_fit + gamma*V_theta(ns), computed from the pre-fit weights
        xptr, xstride = e._x("s")
        L.rcmarl_minibatch_fit(xptr, xstride, e.theta["critic"].data_ptr(), self.fit_t.data_ptr(), len(self.fit),
                               y_c.data_ptr(), perms["critic"].data_ptr(), S, N, B, e.in_c, HID,
                               e.ldp["critic"], e.ldb, FIT_BATCH, FIT_EPOCHS, e.cfg.fast_lr, e.loss["critic"].data_ptr(),
                               e.stream)
        if write_msg:
            e.msg["critic"].index_copy_(1, self.fit_idx, e.theta["critic"].index_select(1, self.fit_idx))
        for ev in joins:
            cur.wait_event(ev)

    # -- all consensus epochs of a block, ahead of the cooperative agents ----------------------------------
    def chain_async(self, B, n_epochs):
        """Greedy / Malicious agents never look at their neighbours (adversarial_CAC_agents.py: their fits use their own
        nets and their own targets only), so within an update block their n_epochs x (10 Keras epochs of mini-batch SGD)
        form ONE dependent chain that does not wait for the cooperative agents' consensus steps.  Launched here in full
        on a side stream; the cooperative side only waits, epoch by epoch, for the message rows it is about to
        aggregate.  OPT-IN (RCMARL_ADV_CHAIN=1): same results, but measured SLOWER than launching epoch by epoch (233 vs
        200 ms per block at 512 seeds x (4 + 1 Malicious)) -- the limit is not the false dependency but that the 1536
        one-wavefront fits of an epoch hold more registers than the chip has (448 each), so whenever they are resident
        the cooperative kernels are not; running ahead keeps them resident all the time.
        Returns a list of (event, {net: rows [S][n_fit][ldp]}) per epoch, or None when there is nothing to run ahead."""
        e = self.e
        if not self.fit or n_epochs <= 0 or e.dev.type != "cuda":
            return None
        if __import__("os").environ.get("RCMARL_ADV_CHAIN", "0") in ("0", "false", ""):
            return None
        if not hasattr(self, "chain_stream"):
            self.chain_stream = torch.cuda.Stream(device=e.dev)
        main = torch.cuda.current_stream(e.dev)
        fork = torch.cuda.Event()
        fork.record(main)
        out = []
        with torch.cuda.stream(self.chain_stream):
            self.chain_stream.wait_event(fork)
            for _ in range(n_epochs):
                # targets of the transmitted critic from ITS current weights (agents/...:128-131), in a buffer of our own:
                # the cooperative side rewrites y_c every epoch
                e._value("ns", e.theta["critic"], "critic", e.ybuf["y_adv"], B, r_applied=e.ybuf["r_fit"], scratch=self.a1t)
                self.phase1(B, y_c=e.ybuf["y_adv"], write_msg=False)
                rows = {net: e.theta[net].index_select(1, self.fit_idx) for net in ("critic", "tr")}
                ev = torch.cuda.Event()
                ev.record(self.chain_stream)
                out.append((ev, rows))
        self._keep_chain = out                         # (rows are read on the main stream: keep them until the next block)
        return out

    def consume(self, item):
        """the cooperative side, before a consensus step: wait for that epoch's messages and put them in place"""
        ev, rows = item
        torch.cuda.current_stream(self.e.dev).wait_event(ev)
        for net, r in rows.items():
            self.e.msg[net].index_copy_(1, self.fit_idx, r)

    # -- phase III ---------------------------------------------------------------------------
    def actor_updates(self, B):
        e, L = self.e, self.e.lib
        S, N, nl = e.S, e.N, e.n_last
        row0 = B - nl
        shuffle = nl > ACTOR_BATCH
        perms = self._draw([(i, "actor") for i in self.adv], 1, nl)["actor"] if shuffle else None
        own = e.theta["critic"]
        if self.mal:
            own = own.clone()
            own.index_copy_(1, self.mal_idx, e.theta["critic_local"].index_select(1, self.mal_idx))
        rptr, rstride = e._x("r", row0)
        L.rcmarl_gather_agent_major(rptr, rstride, None, None, e.ybuf["r_own"].data_ptr(), S, N, nl, e.ldb, e.stream)
        e._value("ns", own, "critic", e.ybuf["v_next_adv"], nl, row0)
        e._value("s", own, "critic", e.ybuf["v_cur_adv"], nl, row0)
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
