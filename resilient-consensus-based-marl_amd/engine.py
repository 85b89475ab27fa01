"""Batched RPBCAC training engine (host side).

One engine instance = S independent seeds x N agents resident on ONE GPU.  It
replaces the body of the reference's ``train_RPBCAC`` (training/train_agents.py:
15-184): the rollout loop (:46-80), the update block (:86-163: I local fits,
II resilient consensus, III actor step, IV replay trim) and the episode
summaries (:168-180) -- with every per-agent Python/Keras loop turned into a
batched HIP kernel launch through the C-ABI of include/rcmarl.h.

PyTorch is used for device memory, copies and streams only; all arithmetic of
the hot path runs in csrc/*.hip.  There is no CPU fallback: the constructor
raises if the HIP library or a GPU is missing (tests inject the hipemu build
explicitly via ``lib=``/``device='cpu'``).
"""
import contextlib
import gc
import math
import os
import time

import numpy as np
import torch

from . import capi
from . import lattice as LT

HID = 20
COOP, FAULTY, GREEDY, MALICIOUS = "Cooperative", "Faulty", "Greedy", "Malicious"


def pad64(n):
    return (int(n) + 63) // 64 * 64


def net_numel(in_dim, out_dim, hid=HID):
    return in_dim * hid + hid + hid * hid + hid + hid * out_dim + out_dim


def net_shapes(in_dim, out_dim, hid=HID):
    return [(in_dim, hid), (hid,), (hid, hid), (hid,), (hid, out_dim), (out_dim,)]


def flatten_params(params):
    return np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in params])


def unflatten_params(vec, in_dim, out_dim, hid=HID):
    out, o = [], 0
    for sh in net_shapes(in_dim, out_dim, hid):
        n = int(np.prod(sh))
        out.append(np.array(vec[o:o + n], dtype=np.float32).reshape(sh))
        o += n
    return out


class EngineConfig:
    """Same keys as the reference's ``args`` dict (main.py:26-44) plus the
    grid size, seed count and RNG mode."""

    def __init__(self, n_agents, agent_label, in_nodes, H=0, gamma=0.9, slow_lr=0.002, fast_lr=0.01, n_actions=5,
                 n_states=2, max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000, common_reward=False,
                 nrow=5, ncol=5, n_seeds=1, rng_mode="device", mu=0.1, scaling=True, randomize_state=True,
                 local_fit_steps=5, lattice="auto", critic_hid=HID):
        self.n_agents, self.agent_label = int(n_agents), list(agent_label)
        self.in_nodes = [list(map(int, row)) for row in in_nodes]
        self.H, self.gamma, self.slow_lr, self.fast_lr = int(H), float(gamma), float(slow_lr), float(fast_lr)
        self.n_actions, self.n_states = int(n_actions), int(n_states)
        self.max_ep_len, self.n_ep_fixed, self.n_epochs = int(max_ep_len), int(n_ep_fixed), int(n_epochs)
        self.buffer_size, self.common_reward = int(buffer_size), bool(common_reward)
        self.nrow, self.ncol, self.n_seeds = int(nrow), int(ncol), int(n_seeds)
        self.rng_mode, self.mu, self.scaling, self.randomize_state = rng_mode, float(mu), bool(scaling), bool(randomize_state)
        self.local_fit_steps = int(local_fit_steps)
        self.lattice = lattice                # layer-1 GEMMs on the exact bf16x3 path: "auto" | True | False
        # hidden width of the critic (the reference builds 20-unit networks, main.py:59-82; BASELINE configs[4] widens
        # the critic to 512 units): any other width runs the dense-GEMM path of csrc/wide_kernels.hip
        self.critic_hid = int(critic_hid)
        assert len(self.agent_label) == self.n_agents and len(self.in_nodes) == self.n_agents
        d = len(self.in_nodes[0])
        for i, row in enumerate(self.in_nodes):
            if len(row) != d:
                raise ValueError("all in-neighbourhoods must have the same size d (got %d and %d)" % (d, len(row)))
            if row[0] != i:
                raise ValueError("in_nodes[i][0] must be i (own value first, agents/resilient_CAC_agents.py:49)")
        if d < 2 * self.H + 1:
            raise ValueError("need d >= 2H+1")
        if self.n_states != 2 or self.n_actions != 5:
            raise ValueError("the grid-world has 2 state dims and 5 actions per agent")
        if rng_mode not in ("device", "numpy"):
            raise ValueError("rng_mode must be 'device' or 'numpy'")
        for lab in self.agent_label:
            if lab not in (COOP, FAULTY, GREEDY, MALICIOUS):
                raise ValueError("unknown agent label %r" % lab)
        if self.critic_hid < 1:
            raise ValueError("critic_hid must be positive")

    @property
    def d(self):
        return len(self.in_nodes[0])


class RPBCACEngine:
    def __init__(self, cfg, seeds=None, device="cuda", lib=None):
        self.cfg = cfg
        if lib is None:
            lib = capi.load()                      # raises if librcmarl_hip.so is missing
            if device == "cuda" and not torch.cuda.is_available():
                raise capi.RcmarlError("rcmarl_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.lib = lib
        self.dev = torch.device(device)
        c = cfg
        S, N = c.n_seeds, c.n_agents
        self.S, self.N = S, N
        self.in_c, self.in_r = N * c.n_states, N * (c.n_states + 1)
        self.hid = {"actor": HID, "critic": c.critic_hid, "tr": HID}
        self.P = {"actor": net_numel(self.in_c, c.n_actions), "critic": net_numel(self.in_c, 1, c.critic_hid),
                  "tr": net_numel(self.in_r, 1)}
        self.in_dim = {"actor": self.in_c, "critic": self.in_c, "tr": self.in_r}
        self.out_dim = {"actor": c.n_actions, "critic": 1, "tr": 1}
        self.in_dim_x = {"s": self.in_c, "ns": self.in_c, "sa": self.in_r}      # width of each replay tensor
        self.ldp = {k: pad64(v) for k, v in self.P.items()}
        self.n_last = c.max_ep_len * c.n_ep_fixed
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.theta = {k: torch.zeros(S, N, self.ldp[k], **f32) for k in ("actor", "critic", "tr")}
        if MALICIOUS in c.agent_label:
            # Malicious agents keep a private critic for their own actor (adversarial_CAC_agents.py:101);
            # only their rows of this matrix are meaningful
            for d_ in (self.P, self.in_dim, self.out_dim, self.ldp, self.hid):
                d_["critic_local"] = d_["critic"]
            self.theta["critic_local"] = torch.zeros(S, N, self.ldp["critic"], **f32)
        self.msg = {k: torch.zeros(S, N, self.ldp[k], **f32) for k in ("critic", "tr")}
        self.adam_m = torch.zeros(S, N, self.ldp["actor"], **f32)
        self.adam_v = torch.zeros(S, N, self.ldp["actor"], **f32)
        self.adam_t = 0
        self.a1_cached = {"critic": False, "tr": False}
        self.a2_cached = False                # wide critic: self.w_a2 holds the fp32 layer-2 activations of the LIVE critic on the s rows
        self._graphs, self.graph_captures, self.graph_replays = {}, 0, 0       # captured update epochs (see _epoch)
        self.loss = {k: torch.zeros(S, N, **f32) for k in ("actor", "critic", "tr")}
        self.rp, self.ybuf = None, {k: None for k in ("r_fit", "y_c", "v_tr", "v_next", "v_cur", "delta", "act_t")}
        self._alloc_row_buffers(c.buffer_size + self.n_last)
        self.B = 0
        # environment
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.pos = [torch.zeros(S, N, 2, **i32) for _ in range(2)]
        self.xs = [torch.zeros(S, 2 * N, **f32) for _ in range(2)]
        self.cur = 0
        # episode-minor state of the episode-parallel rollout (rng_mode 'device'): lanes = episodes
        self.EP = pad64(c.n_ep_fixed)
        if c.rng_mode == "device":
            self.posT = [torch.zeros(S, N, 2, self.EP, **i32) for _ in range(2)]
            self.xsT = [torch.zeros(S, 2 * N, self.EP, **f32) for _ in range(2)]
            self.retT = torch.zeros(S, N, self.EP, dtype=torch.float64, device=self.dev)
        self.goal = torch.zeros(S, N, 2, **i32)
        self.ret = torch.zeros(S, N, dtype=torch.float64, device=self.dev)
        self.ret_hist = torch.zeros(c.n_ep_fixed, S, N, dtype=torch.float64, device=self.dev)
        self.est = torch.zeros(S, N, **f32)
        self.est_hist = torch.zeros(c.n_ep_fixed, S, N, **f32)
        self.probs = torch.zeros(S, N, c.n_actions, **f32)
        self.act_i32 = torch.zeros(S, N, **i32)
        if c.scaling:
            xr, yr = np.arange(c.nrow), np.arange(c.ncol)
            scale = [np.mean(xr), np.mean(yr), np.std(xr), np.std(yr)]     # environments/grid_world.py:29-33
        else:
            scale = [0.0, 0.0, 1.0, 1.0]
        self.scale = torch.tensor(scale, dtype=torch.float64, device=self.dev)
        seeds = list(range(S)) if seeds is None else [int(x) for x in seeds]
        assert len(seeds) == S
        self.seeds = seeds
        self.seeds_dev = torch.tensor(np.asarray(seeds, dtype=np.uint64).view(np.int64), dtype=torch.int64, device=self.dev)
        # graph / roles
        self.nbr = torch.tensor(np.asarray(c.in_nodes, dtype=np.int32), **i32)
        # circulant in-graph (the reference's own pattern, main.py:28) with d = 2H+2: K1 shares one selection
        # network among consecutive agents (rcmarl_consensus_params_circulant; RCMARL_K1_CIRC=0 disables)
        circ = all(row == [(i + k) % N for k in range(c.d)] for i, row in enumerate(c.in_nodes))
        self.k1_circulant = bool(circ and os.environ.get("RCMARL_K1_CIRC", "1") not in ("0", "false")
                                 and lib.rcmarl_consensus_params_circulant_supported(N, c.d, c.H))
        coop = np.array([1 if l == COOP else 0 for l in c.agent_label], dtype=np.int32)
        self.coop_np = coop
        self.n_coop = int(coop.sum())
        self.coop = torch.tensor(coop, dtype=torch.int32, device=self.dev)
        self.coop_idx = torch.tensor([i for i in range(self.N) if coop[i]], dtype=torch.int32, device=self.dev)
        # reward each agent fits on (train_agents.py:106-116): own / r_coop (common_reward) / -r_coop (Malicious)
        mode = np.array([(1 if c.common_reward else 0) if l == COOP else (2 if l == MALICIOUS else 0)
                         for l in c.agent_label], dtype=np.int32)
        self.fit_mode = torch.tensor(mode, **i32)
        self.episode = 0                      # global episode counter
        self.rows_episode_aligned = True      # every replay row so far came from a full max_ep_len-step episode of ours
        self.timers = {"rollout": 0.0, "phase1": 0.0, "phase2": 0.0, "phase3": 0.0, "blocks": 0}
        self.gpow = [float(c.gamma ** j) for j in range(c.max_ep_len)]
        self.initial_state = None             # used when randomize_state is False
        self.np_rngs = None                   # rng_mode='numpy': one RandomState-like object per seed
        self.shard = None                     # shard_agents(): ONE instance over several GPUs (wide critic)
        self._windowed = False

    # ---- everything sized by the replay capacity ---------------------------------------------------
    def _alloc_row_buffers(self, cap):
        """(Re)allocate every buffer whose size follows the replay capacity `cap` (rows per seed): the replay tensors
        (contents kept), per-row vectors, activation scratch, gradient records, lattice operands, wide-critic scratch."""
        c, S, N, lib = self.cfg, self.S, self.N, self.lib
        f32 = dict(dtype=torch.float32, device=self.dev)
        old_rp, old_B = self.rp, getattr(self, "B", 0)
        self.cap = int(cap)
        self.ldb = pad64(self.cap)
        self.a1t = torch.zeros(S, N * HID, self.ldb, **f32)          # scratch layer-1 activations
        # per-net activation buffers: the consensus step leaves layer-1 activations of the live nets on the
        # fit inputs there, which are exactly what step 0 of the next epoch's local fit needs (the fit starts
        # from a copy of the live net and only W3,b3 moved since) -> one forward GEMM per net per epoch saved
        self.a1net = {k: torch.zeros(S, N * self.hid[k], self.ldb, **f32) for k in ("critic", "tr")}
        self.a1_cached["critic"] = self.a1_cached["tr"] = self.a2_cached = False
        self._init_wide()
        self.ybuf = {k: torch.zeros(S, N, self.ldb, **f32) for k in self.ybuf}
        self.rcoop = torch.zeros(S, self.ldb, **f32)
        nchunk_max = (self.cap + 255) // 256
        psz = max(lib.rcmarl_fit_partial_size(HID), lib.rcmarl_actor_partial_size(HID, c.n_actions))
        self.partials = torch.zeros(S * N * nchunk_max * psz, **f32)
        # replay buffers [S][cap][w*N]
        self.rp = {k: torch.zeros(S, self.cap, w * N, **f32) for k, w in (("s", 2), ("ns", 2), ("sa", 3), ("a", 1), ("r", 1))}
        if old_rp is not None and old_B:
            for k in self.rp:
                self.rp[k][:, :old_B] = old_rp[k][:, :old_B]
        self._init_lattice()
        self._init_wide_pk()
        self._ns_term = None                  # scratch: the last next-state row of every episode, gathered (rcmarl_gather_rows)
        self._graphs = {}                     # captured epochs point into the buffers replaced here
        if hasattr(self, "adv"):
            self.adv.a1t = torch.zeros_like(self.a1t)

    def _reserve_rows(self, n_new):
        """The reference's replay lists grow to any length before the post-update trim (train_agents.py:76-80,158-163): a
        caller-supplied exp_buffer, trailing episodes of a previous train() call or a second train() on the same engine
        can push B + n_new past buffer_size + n_ep_fixed*max_ep_len.  Grow the row-sized buffers instead of failing."""
        self._ensure_cap(self.B + int(n_new))

    def _ensure_cap(self, need):
        """Grow geometrically (at least one block of rows, at least 1.5x): growing to exactly `need` made every further
        episode of an over-long buffer re-allocate and copy every row-sized tensor (O(n^2) copies, multi-GB at the
        BASELINE configs[3..4] shapes)."""
        if need > self.cap:
            self.sync()
            self._alloc_row_buffers(max(int(need), self.cap + self.n_last, int(1.5 * self.cap)))

    # ---- wide critic (hid != 20): dense-GEMM path, csrc/wide_kernels.hip ---------------------
    def _init_wide(self):
        """Scratch of the dense-GEMM path: layer-1/2 activations and dz1 of one network family, the head's
        gradient records and the neighbour-estimate matrix of the consensus step."""
        self.wide = self.hid["critic"] != HID
        if not self.wide:
            return
        S, N, L, hid, d = self.S, self.N, self.lib, self.hid["critic"], self.cfg.d
        f32 = dict(dtype=torch.float32, device=self.dev)
        assert self.ldb >= self.EP_pad()
        self.w_a1, self.w_a2, self.w_dz1 = (torch.zeros(S, N * hid, self.ldb, **f32) for _ in range(3))
        self.w_dz3, self.w_ebuf, self.w_v = (torch.zeros(S, N, self.ldb, **f32) for _ in range(3))
        self.w_grads = torch.zeros(S, N, L.rcmarl_wide_grad_size(hid), **f32)
        self.w_losspart = torch.zeros(S, N, (self.cap + L.rcmarl_wide_rows_per_chunk() - 1) // L.rcmarl_wide_rows_per_chunk(), **f32)
        self.w_hmat, self.w_hb = torch.zeros(S, N, d + 1, hid, **f32), torch.zeros(S, N, d + 1, **f32)
        self.w_est = torch.zeros(S, N, d + 1, self.ldb, **f32)

    def EP_pad(self):
        return pad64(self.cfg.n_ep_fixed)

    # ---- wide critic on PRE-SPLIT packed operands (hid % 128 == 0, two-piece f16 form, lattice layer 1): csrc/dense_pk.hip ----
    def _init_wide_pk(self):
        """Packed operands of the rcmarl_pk_* path (include/rcmarl.h has the layouts): the layer-1 activations of the network a fit /
        the consensus step works on in both orientations + their sign bits (`net`: what step 0 of the next local fit re-uses), a
        second replay-row-major image for value passes, the two packed forms of W2, the layer-2 masks, and the small reduction parts.
        RCMARL_WIDE_PK=0 keeps the round-4 path (fp32 operands split while they are staged, wide_kernels.hip)."""
        self.pk = None
        if not (self.wide and getattr(self, "lat_enabled", False)) or self.hid["critic"] % 128:
            return
        if os.environ.get("RCMARL_WIDE_PK", "1") in ("0", "false"):
            return
        S, N, hid = self.S, self.N, self.hid["critic"]
        Z, Bp, JT, JK = S * N, (self.cap + 255) // 256 * 256, hid // 128, hid // 32

        class _Pk:
            pass
        pk = _Pk()
        u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=self.dev)
        f32 = dict(dtype=torch.float32, device=self.dev)
        pk.Bp, pk.bk_rt, pk.kb_kt = Bp, Bp // 128, Bp // 32
        pk.a1_bk = {"net": u8(Z * pk.bk_rt * JK * 2 * LT.PK_BLOCK), "scratch": u8(Z * pk.bk_rt * JK * 2 * LT.PK_BLOCK)}
        pk.a1_kb = u8(Z * JT * pk.kb_kt * 2 * LT.PK_BLOCK)
        pk.s1 = torch.zeros(Z * hid, Bp // 32, dtype=torch.int32, device=self.dev)
        pk.w2t, pk.w2w3 = u8(Z * JT * JK * 2 * LT.PK_BLOCK), u8(Z * JT * JK * 2 * LT.PK_BLOCK)
        pk.rs = torch.zeros(Z, hid, **f32)
        pk.mask_bj, pk.mask_jb = u8(Z * pk.bk_rt * JK * LT.PK_BLOCK), u8(Z * JT * pk.kb_kt * LT.PK_BLOCK)
        pk.vpart, pk.npart = torch.zeros(Z, JT, self.ldb, **f32), torch.zeros(Z, JT, self.ldb, **f32)
        pk.dzv = torch.zeros(Z, 4, Bp, dtype=torch.int16, device=self.dev)
        pk.gw3part, pk.q = torch.zeros(Z, JT, hid, **f32), torch.zeros(Z, hid, **f32)
        pk.gb1part = torch.zeros(Z, (self.cap + 127) // 128, hid, **f32)
        pk.ovf = torch.zeros(1, dtype=torch.int32, device=self.dev)     # set by the kernels when an operand leaves the f16 range
        self.pk = pk

    def _pk_ok(self, net, xkey, B, row0=0):
        """does this pass of a wide net run on the packed-operand path?  (lattice layer 1 on these rows, both operand forms f16)"""
        return (self.pk is not None and self.hid[net] != HID and self._lattice_ok(xkey, B, row0)
                and self.lib.rcmarl_pk_supported(self.hid[net]) == 1 and self.lib.rcmarl_wide_f16_mode() == 1)

    def _wide_forward_pk(self, xkey, theta, net, B, which, fit=False, want_a2=False, skip_layer1=False, wp_fresh=False):
        """layers 1 and 2 of a wide net on packed operands.  which: 'net' (the image a fit / the consensus step leaves behind) or
        'scratch' (value passes).  fit: also a1_kb, s1, the layer-2 masks.  Always leaves the head's value parts in pk.vpart;
        want_a2: the fp32 layer-2 activations in self.w_a2 (phi of the estimate consensus)."""
        L, S, N, hid, pk, st = self.lib, self.S, self.N, self.hid[net], self.pk, self.stream
        in_dim, ldp = self.in_dim[net], self.ldp[net]
        g, wp = self.lat_geom[xkey], self.lat_wp_f[xkey]
        a1_bk = pk.a1_bk[which]
        if want_a2:
            self.a2_cached = False
        full = which == "net"                  # the cached image always carries both orientations and the sign bits
        if not skip_layer1:
            if not wp_fresh:
                L.rcmarl_w1_split(theta.data_ptr(), self.lat_alpha[xkey].data_ptr(), wp.data_ptr(), S, N, in_dim, hid, ldp, g.wp[0],
                                  g.wp[1], st)
            L.rcmarl_layer1_forward_lattice_pk(self.lat_kp[xkey].data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0], g.wp[1],
                                               theta.data_ptr(), a1_bk.data_ptr(), pk.bk_rt, pk.a1_kb.data_ptr() if full else None,
                                               pk.kb_kt, pk.s1.data_ptr() if full else None, pk.Bp // 32, pk.ovf.data_ptr(), S, N, B,
                                               in_dim, hid, ldp, st)
        L.rcmarl_pk_pack_w2(theta.data_ptr(), pk.w2t.data_ptr(), pk.w2w3.data_ptr(), pk.rs.data_ptr(), pk.ovf.data_ptr(), S, N, in_dim,
                            hid, ldp, st)
        L.rcmarl_pk_forward2(pk.w2t.data_ptr(), a1_bk.data_ptr(), pk.bk_rt, theta.data_ptr(), self.w_a2.data_ptr() if want_a2 else None,
                             pk.mask_bj.data_ptr() if fit else None, pk.bk_rt, pk.mask_jb.data_ptr() if fit else None, pk.kb_kt,
                             pk.vpart.data_ptr(), pk.npart.data_ptr() if want_a2 else None, S, N, B, in_dim, hid, ldp, self.ldb, st)

    def _local_fit_wide_pk(self, net, xkey, y, B, mask):
        """_local_fit_wide with every GEMM operand pre-split and packed by its producer (csrc/dense_pk.hip): per step
        layer 1 -> [pack W2] -> layer 2 -> head -> dz1 -> W2 step -> W1 step -> small parameters."""
        L, S, N, hid, in_dim, pk = self.lib, self.S, self.N, self.hid[net], self.in_dim[net], self.pk
        msg, ldp, ldb, st, lr = self.msg[net], self.ldp[net], self.ldb, self.stream, self.cfg.fast_lr
        g, dzp, wp = self.lat_geom[xkey], self.lat_dzp_f[xkey], self.lat_wp_f[xkey]
        wp_fresh = False
        for step in range(self.cfg.local_fit_steps):
            last = step == self.cfg.local_fit_steps - 1
            self._wide_forward_pk(xkey, msg, net, B, "net", fit=True, skip_layer1=(step == 0 and self.a1_cached[net] == "pk"),
                                  wp_fresh=wp_fresh)
            L.rcmarl_pk_head(pk.vpart.data_ptr(), msg.data_ptr(), y.data_ptr(), 0.0, 2, self.w_dz3.data_ptr(), pk.dzv.data_ptr(),
                             self.w_losspart.data_ptr(), S, N, B, in_dim, hid, ldp, ldb, st)
            L.rcmarl_pk_backward_data(pk.mask_bj.data_ptr(), pk.bk_rt, pk.w2w3.data_ptr(), pk.rs.data_ptr(), pk.s1.data_ptr(),
                                      pk.Bp // 32, self.w_dz3.data_ptr(), dzp.data_ptr(), g.dzp[0], g.dzp[1], pk.gb1part.data_ptr(),
                                      pk.ovf.data_ptr(), S, N, B, hid, ldb, st)
            # every gradient comes from the pre-step weights: the W2 step reads W3, the small step updates it afterwards
            L.rcmarl_pk_backward_w2(pk.a1_kb.data_ptr(), pk.kb_kt, pk.mask_jb.data_ptr(), pk.kb_kt, pk.dzv.data_ptr(), msg.data_ptr(),
                                    mask.data_ptr(), pk.gw3part.data_ptr(), pk.q.data_ptr(), S, N, B, in_dim, hid, ldp, lr, st)
            L.rcmarl_layer1_backward_sgd_lattice(self.lat_ktp[xkey].data_ptr(), g.ktp[0], g.ktp[1], dzp.data_ptr(), g.dzp[0], g.dzp[1],
                                                 self.lat_alpha[xkey].data_ptr(), msg.data_ptr(), mask.data_ptr(), S, N, B, in_dim,
                                                 hid, ldp, lr, None if last else wp.data_ptr(), g.wp[0], g.wp[1], st)
            wp_fresh = True
            L.rcmarl_pk_small_sgd(pk.gw3part.data_ptr(), pk.q.data_ptr(), pk.gb1part.data_ptr(), self.w_dz3.data_ptr(),
                                  self.w_losspart.data_ptr(), msg.data_ptr(), mask.data_ptr(),
                                  self.loss[net].data_ptr() if step == 0 else None, S, N, B, in_dim, hid, ldp, ldb, lr, st)
        self.a1_cached[net] = False

    def _wide_offsets(self, net):
        in_dim, hid = self.in_dim[net], self.hid[net]
        o_b1 = in_dim * hid
        o_W2 = o_b1 + hid
        return o_b1, o_W2, o_W2 + hid * hid

    def _wide_forward(self, xkey, theta, net, B, row0=0, a1=None, skip_layer1=False, x=None, wp_fresh=False):
        """layers 1 and 2 of a wide net for every (seed, agent): a1 (default: scratch), then self.w_a2.
        x: (ptr, seed_stride, row_major, ld) of an input other than a replay tensor (rollout start states)."""
        L, S, N, hid = self.lib, self.S, self.N, self.hid[net]
        o_b1, o_W2, o_b2 = self._wide_offsets(net)
        self.a2_cached = False
        a1 = self.w_a1 if a1 is None else a1
        if not skip_layer1:
            if x is None and self._lattice_ok(xkey, B, row0):
                # layer 1 on the exact bf16x3 kernels (80 % of a 2048 -> 512 -> 512 -> 1 critic's flops)
                g, wp = self.lat_geom[xkey], self.lat_wp_f[xkey]
                if not wp_fresh:
                    L.rcmarl_w1_split(theta.data_ptr(), self.lat_alpha[xkey].data_ptr(), wp.data_ptr(), S, N,
                                      self.in_dim[net], hid, self.ldp[net], g.wp[0], g.wp[1], self.stream)
                L.rcmarl_layer1_forward_lattice(self.lat_kp[xkey].data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0],
                                                g.wp[1], theta.data_ptr(), a1.data_ptr(), S, N, B, self.in_dim[net], hid,
                                                self.ldp[net], self.ldb, self.stream)
            else:
                if x is None:
                    ptr, stride = self._x(xkey, row0)
                    x = (ptr, stride, 1, self.in_dim_x[xkey])
                L.rcmarl_dense_forward(x[0], x[1], 0, x[2], x[3], theta.data_ptr(), 0, o_b1, a1.data_ptr(), S, N, B,
                                       self.in_dim[net], hid, self.ldp[net], self.ldb, self.stream)
        L.rcmarl_dense_forward(a1.data_ptr(), N * hid * self.ldb, hid * self.ldb, 0, self.ldb, theta.data_ptr(), o_W2, o_b2,
                               self.w_a2.data_ptr(), S, N, B, hid, hid, self.ldp[net], self.ldb, self.stream)

    def _local_fit_wide(self, net, xkey, y, B, mask):
        """_local_fit for a wide net: every layer of every step is a dense GEMM per agent (f32 MFMA)."""
        if self._pk_ok(net, xkey, B) and xkey in self.lat_ktp:
            return self._local_fit_wide_pk(net, xkey, y, B, mask)
        L, S, N, hid, in_dim = self.lib, self.S, self.N, self.hid[net], self.in_dim[net]
        msg, a1, a2, dz1 = self.msg[net], self.a1net[net], self.w_a2, self.w_dz1
        ldp, ldb, st, lr = self.ldp[net], self.ldb, self.stream, self.cfg.fast_lr
        _, o_W2, _ = self._wide_offsets(net)
        ptr, stride = self._x(xkey)
        lat = self._lattice_ok(xkey, B, 0) and xkey in self.lat_ktp
        g = self.lat_geom[xkey] if lat else None
        wp_fresh = False
        for step in range(self.cfg.local_fit_steps):
            self._wide_forward(xkey, msg, net, B, a1=a1, skip_layer1=(step == 0 and self.a1_cached[net] is True), wp_fresh=wp_fresh)
            L.rcmarl_wide_head_fit(a2.data_ptr(), msg.data_ptr(), y.data_ptr(), self.w_dz3.data_ptr(), self.w_grads.data_ptr(),
                                   self.w_losspart.data_ptr(), S, N, B, in_dim, hid, ldp, ldb, st)      # a2 now holds dz2
            L.rcmarl_dense_backward_data(a2.data_ptr(), msg.data_ptr(), o_W2, a1.data_ptr(), dz1.data_ptr(), S, N, B, hid, hid,
                                         ldp, ldb, st)
            if not lat:                                            # (lattice path: the row sums ride on the pack pass below)
                L.rcmarl_wide_bias_grad(dz1.data_ptr(), self.w_grads.data_ptr(), S, N, B, hid, ldb, st)
            # every gradient above came from the pre-step weights; now the updates
            L.rcmarl_dense_backward_sgd(a1.data_ptr(), N * hid * ldb, hid * ldb, 0, ldb, a2.data_ptr(), msg.data_ptr(), o_W2,
                                        mask.data_ptr(), S, N, B, hid, hid, ldp, ldb, lr, st)
            if lat:
                dzp, wp = self.lat_dzp_f[xkey], self.lat_wp_f[xkey]
                L.rcmarl_lattice_pack_dz_rowsum(dz1.data_ptr(), dzp.data_ptr(), self.w_grads.data_ptr(), 3 * hid + 1, 2 * hid + 1,
                                                S, N, B, hid, ldb, g.dzp[0], g.dzp[1], st)       # dz1 pieces + gb1 = its row sums
                L.rcmarl_layer1_backward_sgd_lattice(self.lat_ktp[xkey].data_ptr(), g.ktp[0], g.ktp[1], dzp.data_ptr(),
                                                     g.dzp[0], g.dzp[1], self.lat_alpha[xkey].data_ptr(), msg.data_ptr(),
                                                     mask.data_ptr(), S, N, B, in_dim, hid, ldp, lr,
                                                     None if step == self.cfg.local_fit_steps - 1 else wp.data_ptr(), g.wp[0],
                                                     g.wp[1], st)
                wp_fresh = True           # the epilogue left the split of the updated W1 in lat_wp
            else:
                L.rcmarl_dense_backward_sgd(ptr, stride, 0, 1, self.in_dim_x[xkey], dz1.data_ptr(), msg.data_ptr(), 0,
                                            mask.data_ptr(), S, N, B, in_dim, hid, ldp, ldb, lr, st)
            L.rcmarl_wide_small_sgd(self.w_grads.data_ptr(), self.w_losspart.data_ptr(), msg.data_ptr(), mask.data_ptr(),
                                    self.loss[net].data_ptr() if step == 0 else None, S, N, B, in_dim, hid, ldp, lr, st)
        self.a1_cached[net] = False

    def _consensus_wide(self, net, xkey, B, msg_all=None):
        L, S, N, c, hid = self.lib, self.S, self.N, self.cfg, self.hid[net]
        msg_all = self.msg[net] if msg_all is None else msg_all      # rows indexed by GLOBAL agent (in_nodes)
        self._k1(net, self.P[net] - (hid + 1))
        pk = self._pk_ok(net, xkey, B) and xkey in self.lat_ktp
        if pk:      # the packed layer-1 image stays behind for step 0 of the next local fit; phi = fp32 layer-2 activations
            self._wide_forward_pk(xkey, self.theta[net], net, B, "net", want_a2=True)
        else:
            self._wide_forward(xkey, self.theta[net], net, B, a1=self.a1net[net])
        if pk:      # |phi|^2 per row came out of the forward's epilogue
            L.rcmarl_wide_consensus_head_nrm(self.w_a2.data_ptr(), self.pk.npart.data_ptr(), L.rcmarl_pk_parts(hid), self.theta[net].data_ptr(),
                                             msg_all.data_ptr(), self.nbr.data_ptr(), self.coop.data_ptr(), self.w_hmat.data_ptr(),
                                             self.w_hb.data_ptr(), self.w_est.data_ptr(), self.w_ebuf.data_ptr(), self.w_grads.data_ptr(),
                                             None, S, N, B, self.in_dim[net], hid, self.ldp[net], self.ldb, c.d, c.H, self.stream)
        else:
            L.rcmarl_wide_consensus_head(self.w_a2.data_ptr(), self.theta[net].data_ptr(), msg_all.data_ptr(),
                                         self.nbr.data_ptr(), self.coop.data_ptr(), None, self.w_hmat.data_ptr(),
                                         self.w_hb.data_ptr(), self.w_est.data_ptr(), self.w_ebuf.data_ptr(),
                                         self.w_grads.data_ptr(), None, S, N, B, self.in_dim[net], hid, self.ldp[net], self.ldb,
                                         c.d, c.H, self.stream)
        self.a1_cached[net] = ("pk" if pk else True) if self.reuse_activations else False
        self.a2_cached = bool(pk and net == "critic" and xkey == "s" and self.reuse_activations)
        L.rcmarl_wide_head_apply(self.w_grads.data_ptr(), self.theta[net].data_ptr(), self.coop.data_ptr(), S, N, B,
                                 self.in_dim[net], hid, self.ldp[net], self.stream)

    def _value_wide(self, xkey, theta, net, out, B, row0=0, r_applied=None, x=None):
        if x is None and self._pk_ok(net, xkey, B, row0):
            self._wide_forward_pk(xkey, theta, net, B, "scratch")
            self.lib.rcmarl_pk_head(self.pk.vpart.data_ptr(), theta.data_ptr(), self._p(r_applied), self.cfg.gamma,
                                    0 if r_applied is None else 1, out.data_ptr(), None, None, self.S, self.N, B, self.in_dim[net],
                                    self.hid[net], self.ldp[net], self.ldb, self.stream)
            return
        self._wide_forward(xkey, theta, net, B, row0, x=x)
        self.lib.rcmarl_wide_head_value(self.w_a2.data_ptr(), theta.data_ptr(), self._p(r_applied), self.cfg.gamma,
                                        out.data_ptr(), self.S, self.N, B, self.in_dim[net], self.hid[net], self.ldp[net],
                                        self.ldb, self.stream)

    # ---- ONE instance over several GPUs (C2, SURVEY.md 8e / BASELINE configs[4]) ---------------------------------
    def shard_agents(self, rank=None, world=None, group=None, comm=None, force=False):
        """Shard the wide critic of this (single-seed) instance over the ranks of `group`: rank r owns a contiguous block of
        agents for everything that is independent per agent (TD targets, local fits, estimate consensus + projection,
        values) and a block of parameter COLUMNS for the hidden-layer consensus K1; two all-to-all transposes of the
        message matrix per epoch (parallel.ShardedConsensus) replace the reference's in-process gather
        `[critic_weights[i] for i in in_nodes[node]]` (training/train_agents.py:129-130), one small all-gather carries
        the neighbours' output layers (hid+1 floats per agent) and one the critic values the actors need.  The 20-unit
        nets (team reward, actors), the environment and the replay buffer stay replicated: they are a few per cent of a
        wide-critic block and every rank computes them bit-identically.  Results equal the unsharded engine bit for bit
        (tests/test_sharded_engine_gloo.py).  Call after torch.distributed.init_process_group; world 1 = no-op.
        comm: the collectives (parallel.TorchComm over `group` by default; parallel.ThreadComm in the one-GPU test).
        force: shard even at world size 1, so that EVERY collective of the sharded instance runs through the communicator
        (the one-rank RCCL run on a single-GPU box: tests/test_rccl_one_rank_gpu.py, `bench.py --workload cfg5_shard`)."""
        from .parallel import ShardedConsensus, TorchComm, agent_range
        if comm is None and world is None:
            comm = TorchComm(group)
        if comm is not None:
            rank, world = comm.rank, comm.world
        if world == 1 and not force:
            self.shard = None
            return self
        if self.S != 1 or not self.wide:
            raise ValueError("agent sharding is for ONE instance (n_seeds == 1) with a wide critic; independent seeds shard "
                             "over ranks without any exchange (parallel.shard_seeds)")
        if self.N % world:
            raise ValueError("n_agents must be a multiple of the number of ranks")
        if self.n_coop != self.N:
            raise ValueError("the agent-sharded instance is built for all-cooperative teams")
        a0, a1 = agent_range(self.N, rank, world)
        hid = self.hid["critic"]
        if self.lat_enabled and ((a1 - a0) * hid) % 128:
            raise ValueError("agents per rank x critic width must be a multiple of 128 (packed operand row tiles)")

        class _Shard:
            pass
        sh = _Shard()
        sh.rank, sh.world, sh.a0, sh.a1, sh.n_loc, sh.N = rank, world, a0, a1, a1 - a0, self.N
        sh.comm = TorchComm(group) if comm is None else comm
        assert (sh.comm.rank, sh.comm.world) == (rank, world), "rank/world do not match the communicator"
        # the 20-unit team-reward net is sharded the same way whenever its packed lattice operands split on 128-row tiles
        # (32 agents per rank); otherwise it stays replicated like the actors (a few per cent of a wide-critic block)
        sh.shard_tr = (not self.lat_enabled) or ((a1 - a0) * HID) % 128 == 0
        sh.sc = {}
        for net in ("critic", "tr") if sh.shard_tr else ("critic",):
            sh.sc[net] = ShardedConsensus(self.lib, 1, self.N, self.P[net] - (self.hid[net] + 1), self.cfg.d, self.cfg.H,
                                          self.cfg.in_nodes, self.coop_np, self.dev, comm=sh.comm, circulant=self.k1_circulant,
                                          force_collectives=force)
        self.shard = sh
        return self

    def _sharded(self, net):
        """is this network family's per-agent work split over the ranks (and are we not already inside a window)?"""
        sh = self.shard
        return sh is not None and not self._windowed and (net == "critic" or (net == "tr" and sh.shard_tr))

    def _wv(self, t):
        """this rank's agents of a per-agent tensor ([N] or [1][N][...])"""
        if t is None or self.shard is None:
            return t
        sh = self.shard
        return t[sh.a0:sh.a1] if t.dim() == 1 else t[:, sh.a0:sh.a1]

    @contextlib.contextmanager
    def _agent_window(self):
        """Inside, every per-agent buffer of the sharded network families IS its slice for this rank's agents and self.N the
        number of those agents: with one seed an agent range is a contiguous piece of every [S][N][...] tensor (and a
        range of 128-row tiles of the packed lattice operands), so the kernels run unchanged on (pointer, N_local).
        (Scratch that is only ever indexed from its base -- a1t, partials -- needs no slice.)"""
        sh = self.shard
        if sh is None or self._windowed:
            yield
            return
        a0, a1, hid = sh.a0, sh.a1, self.hid["critic"]
        rows = lambda t: t[:, a0:a1]
        units = lambda t: t[:, a0 * hid:a1 * hid]
        saved = []

        def swap_attr(name, fn):
            full = getattr(self, name)
            saved.append((self, name, full, False))
            setattr(self, name, fn(full))

        def swap_item(dname, key, value):
            d_ = getattr(self, dname)
            saved.append((d_, key, d_[key], True))
            d_[key] = value

        for name in ("w_a1", "w_a2", "w_dz1"):
            swap_attr(name, units)
        for name in ("w_dz3", "w_ebuf", "w_v", "w_grads", "w_losspart", "w_hmat", "w_hb", "w_est"):
            swap_attr(name, rows)
        swap_attr("coop", lambda t: t[a0:a1])
        swap_attr("nbr", lambda t: t[a0:a1])
        families = [("critic", hid, ("s", "ns"), self.in_c)] + ([("tr", HID, ("sa",), self.in_r)] if sh.shard_tr else [])
        for net, h_, xkeys, in_dim in families:
            for dname in ("theta", "msg", "loss"):
                swap_item(dname, net, rows(getattr(self, dname)[net]))
            swap_item("a1net", net, self.a1net[net][:, a0 * h_:a1 * h_])
            if self.lat_enabled:
                g_full = self.lat_geom[xkeys[0]]
                g_loc = LT.Geometry(sh.n_loc, in_dim, self.cap, h_)
                rt0 = a0 * h_ // 128
                for dname, rk_full, rk_loc in (("lat_wp_f", g_full.wp, g_loc.wp), ("lat_dzp_f", g_full.dzp, g_loc.dzp)):
                    per_rt = rk_full[1] * 3 * LT.PK_BLOCK
                    view = getattr(self, dname)[xkeys[0]][rt0 * per_rt:(rt0 + rk_loc[0]) * per_rt]
                    for xk in xkeys:
                        if xk in getattr(self, dname):
                            swap_item(dname, xk, view)
                for xk in xkeys:
                    swap_item("lat_geom", xk, g_loc)
        for key in ("y_c", "r_fit", "v_next", "v_cur", "v_tr"):
            swap_item("ybuf", key, rows(self.ybuf[key]))
        swap_attr("N", lambda n: sh.n_loc)
        self._windowed = True
        try:
            yield
        finally:
            self._windowed = False
            for obj, key, full, is_item in reversed(saved):
                if is_item:
                    obj[key] = full
                else:
                    setattr(obj, key, full)

    def _allgather_rows(self, full, c0, c1):
        """full [1][N][...]: every rank holds its own agents' rows of columns c0..c1; afterwards all rows everywhere"""
        sh = self.shard
        send = full[0, sh.a0:sh.a1, c0:c1].contiguous()
        recv = [torch.empty_like(send) for _ in range(sh.world)]
        sh.comm.all_gather(recv, send)
        for r, blk in enumerate(recv):
            if r != sh.rank:
                full[0, r * sh.n_loc:(r + 1) * sh.n_loc, c0:c1] = blk

    def sync_shards(self):
        """Every rank gets the other ranks' rows of the sharded networks (checkpoints, get_weights, the end of train()).
        A COLLECTIVE when the instance is sharded: every rank must call it (so also state_dict / save_checkpoint, which
        call it); no-op otherwise."""
        if self.shard is not None:
            for net in self.shard.sc:
                self._allgather_rows(self.theta[net], 0, self.ldp[net])
                self._allgather_rows(self.loss[net].unsqueeze(-1), 0, 1)
            self._shards_synced = True

    # ---- lattice (exact bf16x3) layer-1 path: csrc/lattice_gemm.hip, lattice.py ------------
    def _init_lattice(self):
        """Packed bf16 operands of the lattice GEMMs.  Policy "auto": on from 16 agents up (below that the
        128x256 tiles are mostly padding and the f32-MFMA kernels are launch-bound anyway);
        RCMARL_LATTICE=0/1 overrides.  The path is only USED for an update block whose replay rows pass the
        lattice check of rcmarl_lattice_encode (always true for rows produced by the grid-world)."""
        c = self.cfg
        want = c.lattice
        env = os.environ.get("RCMARL_LATTICE")
        if env is not None:
            want = env not in ("0", "false", "False", "")
        if want == "auto":
            want = self.N >= 16
        self.lat_enabled = bool(want)
        self.lat_active = False               # set per update block by _lattice_encode
        if not self.lat_enabled:
            return
        u8 = lambda rk, pieces: torch.zeros(self.S * LT.Geometry.nbytes(rk, pieces), dtype=torch.uint8, device=self.dev)
        # the state family's weight / dz operands belong to the critic (the actor only ever runs on the last rows of
        # the buffer, i.e. off this path), the state-action family's to the team-reward net
        self.lat_geom = {"s": LT.Geometry(self.N, self.in_c, self.cap, self.hid["critic"]),
                         "sa": LT.Geometry(self.N, self.in_r, self.cap, self.hid["tr"])}
        self.lat_geom["ns"] = self.lat_geom["s"]
        self.lat_kp = {k: u8(self.lat_geom[k].kp, 1) for k in ("s", "ns", "sa")}
        self.lat_ktp = {k: u8(self.lat_geom[k].ktp, 1) for k in ("s", "sa")}
        # scratch per input family (the TR and critic local fits may run concurrently on two streams)
        self.lat_wp_f = {"sa": u8(self.lat_geom["sa"].wp, 3), "s": u8(self.lat_geom["s"].wp, 3)}
        self.lat_dzp_f = {"sa": u8(self.lat_geom["sa"].dzp, 3), "s": u8(self.lat_geom["s"].dzp, 3)}
        self.lat_wp_f["ns"] = self.lat_wp_f["s"]
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.lat_alpha = {"s": torch.tensor(LT.column_alpha(self.N, 2, c.nrow, c.ncol, c.scaling), **f32),
                          "sa": torch.tensor(LT.column_alpha(self.N, 3, c.nrow, c.ncol, c.scaling), **f32)}
        self.lat_alpha["ns"] = self.lat_alpha["s"]
        self.lat_flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # the last next-state row of every episode (the TD target's own forward pass, _value_next_cached): a small image of its own
        self.lat_geom_term = LT.Geometry(self.N, self.in_c, max(1, -(-self.cap // max(c.max_ep_len, 1))), self.hid["critic"])
        self.lat_kp_term = u8(self.lat_geom_term.kp, 1)
        self.lat_flag_term = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def __del__(self):
        # the library remembers the operand form of every packed buffer by ADDRESS: tell it these addresses are free again, so a
        # later allocation that lands on one of them does not inherit a stale form (include/rcmarl.h: rcmarl_lattice_forget)
        try:
            forget = getattr(self.lib, "rcmarl_lattice_forget", None)
            if forget is None or not getattr(self, "lat_enabled", False):
                return
            for d in (self.lat_kp, self.lat_ktp, self.lat_wp_f, self.lat_dzp_f):
                for t in d.values():
                    forget(t.data_ptr())
            forget(self.lat_kp_term.data_ptr())
        except Exception:
            pass

    def _lattice_encode(self, B):
        """Once per update block: integer-lattice images of the replay tensors (rows 0..B)."""
        self.lat_active = False
        if not self.lat_enabled:
            return
        L = self.lib
        self.lat_flag.zero_()
        for k in ("s", "ns", "sa"):
            g = self.lat_geom[k]
            ptr, stride = self._x(k)
            ktp = self.lat_ktp.get(k)
            L.rcmarl_lattice_encode(ptr, stride, self.lat_alpha[k].data_ptr(), self.S, B, self.in_dim_x[k],
                                    self.lat_kp[k].data_ptr(), g.kp[0], g.kp[1], self._p8(ktp), g.ktp[0], g.ktp[1],
                                    self.lat_flag.data_ptr(), self.stream)
        self.lat_active = int(self.lat_flag.item()) == 0       # one host read per update block
        self.lat_B = B

    @staticmethod
    def _p8(t):
        return None if t is None else t.data_ptr()

    def _lattice_ok(self, xkey, B, row0):
        return self.lat_active and row0 == 0 and B == self.lat_B and xkey in self.lat_kp

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else None

    def sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    @staticmethod
    def _p(t, offset_elems=0):
        return None if t is None else t.data_ptr() + 4 * int(offset_elems)

    def set_weights(self, seed_idx, agent, net, params):
        vec = flatten_params(params)
        assert vec.size == self.P[net], (net, vec.size, self.P[net])
        self.theta[net][seed_idx, agent, :vec.size] = torch.from_numpy(vec).to(self.dev)
        if net == "critic" and "critic_local" in self.theta:     # Malicious ctor: critic_local = copy(critic) (:101)
            self.theta["critic_local"][seed_idx, agent, :vec.size] = self.theta["critic"][seed_idx, agent, :vec.size]

    def get_weights(self, seed_idx, agent, net):
        if self.shard is not None and net in self.shard.sc and not self._shards_synced:
            raise RuntimeError("agent-sharded instance: rows of other ranks are stale until sync_shards() (a collective: "
                               "every rank must call it) -- train() and state_dict() do")
        vec = self.theta[net][seed_idx, agent, :self.P[net]].detach().cpu().numpy()
        return unflatten_params(vec, self.in_dim[net], self.out_dim[net], self.hid[net])

    def set_all_weights(self, net, array):
        """array: [S][N][P] fp32."""
        arr = np.asarray(array, dtype=np.float32)
        assert arr.shape == (self.S, self.N, self.P[net])
        self.theta[net][:, :, :self.P[net]] = torch.from_numpy(arr).to(self.dev)
        if net == "critic" and "critic_local" in self.theta:
            self.theta["critic_local"].copy_(self.theta["critic"])

    def get_all_weights(self, net):
        return self.theta[net][:, :, :self.P[net]].detach().cpu().numpy()

    def load_adam(self, seed_idx, agent, m, v, t):
        """Adam slots of one actor (flat fp32 vectors in parameter-row order) and its step count.
        The batched engine keeps ONE step counter for all cooperative actors (they all take one step
        per block), so `t` must agree across cooperative agents."""
        n = self.P["actor"]
        self.adam_m[seed_idx, agent, :n] = torch.from_numpy(np.asarray(m, np.float32)).to(self.dev)
        self.adam_v[seed_idx, agent, :n] = torch.from_numpy(np.asarray(v, np.float32)).to(self.dev)
        if self.cfg.agent_label[agent] == COOP:
            loaded = getattr(self, "_adam_t_loaded", None)
            if loaded is not None and loaded != int(t):
                raise ValueError("cooperative actors disagree on their Adam step count (%d vs %d): the batched engine steps "
                                 "all of them together and keeps one bias-correction factor" % (loaded, int(t)))
            self._adam_t_loaded = self.adam_t = int(t)
        else:
            self._require_adversary_support()
            self.adv.adam_t = int(t)

    def dump_adam(self, seed_idx, agent):
        n = self.P["actor"]
        t = self.adam_t if self.cfg.agent_label[agent] == COOP else (self.adv.adam_t if hasattr(self, "adv") else 0)
        return (self.adam_m[seed_idx, agent, :n].detach().cpu().numpy().copy(),
                self.adam_v[seed_idx, agent, :n].detach().cpu().numpy().copy(), int(t))

    def init_glorot(self, base_seed=0):
        """Keras-default initialisation (Glorot-uniform kernels, zero biases; reference
        main.py:59-82) drawn from a NumPy generator per seed (TensorFlow's own init
        stream is not reproducible outside TensorFlow -- SURVEY.md section 7)."""
        for net in ("actor", "critic", "tr"):
            arr = np.zeros((self.S, self.N, self.P[net]), np.float32)
            for s in range(self.S):
                rng = np.random.default_rng([int(base_seed), int(self.seeds[s]) & 0x7FFFFFFF, {"actor": 0, "critic": 1, "tr": 2}[net]])
                for n in range(self.N):
                    o = 0
                    for sh in net_shapes(self.in_dim[net], self.out_dim[net], self.hid[net]):
                        cnt = int(np.prod(sh))
                        if len(sh) == 2:
                            lim = math.sqrt(6.0 / (sh[0] + sh[1]))
                            arr[s, n, o:o + cnt] = rng.uniform(-lim, lim, size=cnt).astype(np.float32)
                        o += cnt
            self.set_all_weights(net, arr)

    def set_goals(self, desired):
        """desired: [S][N][2] (or [N][2], broadcast over seeds) integer goal cells."""
        d = np.asarray(desired, dtype=np.int32)
        if d.ndim == 2:
            d = np.broadcast_to(d, (self.S, self.N, 2))
        self.goal.copy_(torch.from_numpy(np.array(d, dtype=np.int32)).to(self.dev))

    def load_replay(self, states, nstates, actions, rewards, seed_idx=0):
        """exp_buffer of the reference API (train_agents.py:36-40): lists of per-step arrays."""
        B = len(states)
        if B == 0:
            return
        self._ensure_cap(B + self.n_last)         # room for the rows of the next block (rows of other seeds are kept)
        s = np.asarray(states, np.float32).reshape(B, -1)
        ns = np.asarray(nstates, np.float32).reshape(B, -1)
        a = np.asarray(actions, np.float32).reshape(B, -1)
        r = np.asarray(rewards, np.float32).reshape(B, -1)
        sa = np.concatenate([s.reshape(B, self.N, 2), a.reshape(B, self.N, 1)], axis=-1).reshape(B, -1)
        for k, v in (("s", s), ("ns", ns), ("a", a), ("r", r), ("sa", sa)):
            self.rp[k][seed_idx, :B] = torch.from_numpy(np.ascontiguousarray(v)).to(self.dev)
        self.B = B
        self.rows_episode_aligned = False      # caller-supplied rows: no assumption about episode boundaries

    def dump_replay(self, seed_idx=0):
        B = self.B
        g = lambda k: self.rp[k][seed_idx, :B].detach().cpu().numpy()
        s, ns, a, r = g("s"), g("ns"), g("a"), g("r")
        N = self.N
        return ([s[b].reshape(N, 2).astype(np.float64) for b in range(B)], [ns[b].reshape(N, 2).astype(np.float64) for b in range(B)],
                [a[b].reshape(N, 1).astype(np.float64) for b in range(B)], [r[b].reshape(N, 1).astype(np.float64) for b in range(B)])

    # ---- mid-run checkpoint (the reference only saves final weights, main.py:119-121) ------
    _CKPT_TENSORS = ("goal", "adam_m", "adam_v", "ret_hist", "est_hist")

    def state_dict(self):
        """Everything a bit-identical resume needs: parameters, Adam slots and step counts, replay rows,
        agent positions, goals and the RNG position (device mode: the episode counter of the counter-based
        Philox stream; numpy mode: the per-seed RandomState states)."""
        c = self.cfg
        self.sync_shards()
        sd = {"format": 1, "S": self.S, "N": self.N, "seeds": list(self.seeds), "episode": self.episode, "B": self.B,
              "adam_t": self.adam_t, "cur": self.cur, "labels": list(c.agent_label), "in_nodes": c.in_nodes,
              "theta": {k: v.detach().cpu() for k, v in self.theta.items()},
              "replay": {k: v[:, :self.B].detach().cpu() for k, v in self.rp.items()},
              "pos": self.pos[self.cur].detach().cpu(), "xs": self.xs[self.cur].detach().cpu(),
              "rows_episode_aligned": self.rows_episode_aligned}
        for k in self._CKPT_TENSORS:
            sd[k] = getattr(self, k).detach().cpu()
        if hasattr(self, "adv"):
            sd["adv"] = self.adv.state_dict()
        sd["shape"] = {"H": c.H, "critic_hid": c.critic_hid, "buffer_size": c.buffer_size, "n_ep_fixed": c.n_ep_fixed,
                       "max_ep_len": c.max_ep_len, "nrow": c.nrow, "ncol": c.ncol, "rng_mode": c.rng_mode}
        if self.np_rngs is not None:          # plain tensors / numbers only, so the file loads with weights_only=True
            sd["np_rngs"] = []
            for r in self.np_rngs:
                name, keys, pos, has_gauss, cached = r.get_state()
                sd["np_rngs"].append({"name": str(name), "keys": torch.from_numpy(np.asarray(keys, np.uint32).astype(np.int64)),
                                      "pos": int(pos), "has_gauss": int(has_gauss), "cached": float(cached)})
        return sd

    def load_state_dict(self, sd):
        c = self.cfg
        if sd.get("format") != 1 or sd["S"] != self.S or sd["N"] != self.N or list(sd["labels"]) != list(c.agent_label):
            raise ValueError("checkpoint does not match this engine (format/S/N/agent labels)")
        if [list(map(int, r)) for r in sd["in_nodes"]] != c.in_nodes:
            raise ValueError("checkpoint was written for a different communication graph")
        mine = {"H": c.H, "critic_hid": c.critic_hid, "buffer_size": c.buffer_size, "n_ep_fixed": c.n_ep_fixed,
                "max_ep_len": c.max_ep_len, "nrow": c.nrow, "ncol": c.ncol, "rng_mode": c.rng_mode}
        theirs = sd.get("shape", mine)
        diff = {k: (theirs.get(k), v) for k, v in mine.items() if theirs.get(k) != v}
        if diff:
            raise ValueError("checkpoint does not match this engine: %r (checkpoint, engine)" % diff)
        for k, v in sd["theta"].items():
            if k not in self.theta or tuple(v.shape) != tuple(self.theta[k].shape):
                raise ValueError("checkpoint parameter matrix %r has shape %r, engine expects %r"
                                 % (k, tuple(v.shape), tuple(self.theta[k].shape) if k in self.theta else None))
        self.seeds = [int(x) for x in sd["seeds"]]
        self.seeds_dev.copy_(torch.tensor(np.asarray(self.seeds, dtype=np.uint64).view(np.int64), dtype=torch.int64))
        self._ensure_cap(int(sd["B"]))
        self.episode, self.B, self.adam_t, self.cur = int(sd["episode"]), int(sd["B"]), int(sd["adam_t"]), int(sd["cur"])
        for k, v in sd["theta"].items():
            self.theta[k].copy_(v)
        for k, v in sd["replay"].items():
            self.rp[k][:, :self.B].copy_(v)
        self.pos[self.cur].copy_(sd["pos"])
        self.xs[self.cur].copy_(sd["xs"])
        for k in self._CKPT_TENSORS:
            getattr(self, k).copy_(sd[k])
        if "adv" in sd:
            self._require_adversary_support()
            self.adv.load_state_dict(sd["adv"])
        if "np_rngs" in sd:
            self.np_rngs = []
            for st in sd["np_rngs"]:
                r = np.random.RandomState()
                r.set_state((st["name"], st["keys"].numpy().astype(np.uint32), int(st["pos"]), int(st["has_gauss"]),
                             float(st["cached"])))
                self.np_rngs.append(r)
        self.a1_cached["critic"] = self.a1_cached["tr"] = self.a2_cached = False
        self.lat_active = False
        self.rows_episode_aligned = bool(sd.get("rows_episode_aligned", False))

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path):
        # tensors, numbers, strings and containers of them only: nothing in the file is unpickled into code
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    # ---- rollout (train_agents.py:46-80) ---------------------------------------------------
    def _reset(self, host_positions=None):
        c, L = self.cfg, self.lib
        pin = None
        if host_positions is not None:
            pin = torch.from_numpy(np.ascontiguousarray(np.asarray(host_positions, dtype=np.int32))).to(self.dev)
        L.rcmarl_env_reset(self._p(pin), self.seeds_dev.data_ptr(), c.nrow, c.ncol, self.scale.data_ptr(), self.episode,
                           self.pos[self.cur].data_ptr(), self.xs[self.cur].data_ptr(), self.ret.data_ptr(), self.S,
                           self.N, self.stream)

    def _replay_ptrs(self):
        return [self.rp[k].data_ptr() for k in ("s", "ns", "sa", "a", "r")]

    def rollout_episode(self, ep_in_block):
        """One episode for all seeds; appends max_ep_len rows to the replay buffers."""
        c, L, S, N = self.cfg, self.lib, self.S, self.N
        self._reserve_rows(c.max_ep_len)
        if c.rng_mode == "numpy":
            if self.np_rngs is None:          # one legacy stream per seed, seeded like main.py:46 seeds the global one
                self.np_rngs = [np.random.RandomState(int(sd) & 0xFFFFFFFF) for sd in self.seeds]
            if c.randomize_state:                                     # env.reset(): grid_world.py:39-40
                host_pos = np.stack([rng.randint([0, 0], [c.nrow, c.ncol], size=(N, 2)) for rng in self.np_rngs])
            else:
                host_pos = np.broadcast_to(np.asarray(self.initial_state), (S, N, 2))
            self._reset(host_pos)
        else:
            self._reset(None if c.randomize_state else np.broadcast_to(np.asarray(self.initial_state), (S, N, 2)))
        # expected returns at the start state (train_agents.py:60-62)
        if self.wide:
            with self._agent_window():
                self._value_wide(None, self.theta["critic"], "critic", self.w_v, 1, x=(self.xs[self.cur].data_ptr(), 2 * N, 1, 2 * N))
            if self.shard is not None:
                self._allgather_rows(self.w_v, 0, 1)
            self.est.copy_(self.w_v[:, :, 0])
        else:
            L.rcmarl_value_rows(self.xs[self.cur].data_ptr(), self.theta["critic"].data_ptr(), self.est.data_ptr(), S, N,
                                self.in_c, HID, self.ldp["critic"], self.stream)
        self.est_hist[ep_in_block].copy_(self.est)
        rp = self._replay_ptrs()
        for j in range(c.max_ep_len):
            cur, nxt = self.cur, 1 - self.cur
            row = self.B + j
            if c.rng_mode == "device":
                L.rcmarl_rollout_step(self.xs[cur].data_ptr(), self.pos[cur].data_ptr(), self.goal.data_ptr(),
                                      self.theta["actor"].data_ptr(), self.seeds_dev.data_ptr(), c.nrow, c.ncol,
                                      self.scale.data_ptr(), rp[0], rp[1], rp[2], rp[3], rp[4], self.cap, row,
                                      self.pos[nxt].data_ptr(), self.xs[nxt].data_ptr(), self.ret.data_ptr(),
                                      self.gpow[j], self.episode, j, c.mu, S, N, HID, c.n_actions, self.ldp["actor"],
                                      None, self.stream)
            else:
                L.rcmarl_policy_probs(self.xs[cur].data_ptr(), self.theta["actor"].data_ptr(), self.probs.data_ptr(), S, N,
                                      self.in_c, HID, c.n_actions, self.ldp["actor"], self.stream)
                probs = self.probs.detach().cpu().numpy()
                acts = np.zeros((S, N), np.int32)
                for s in range(S):                                    # get_action(): agents/...:208-219
                    rng = self.np_rngs[s]
                    for i in range(N):
                        a_rand = rng.choice(c.n_actions)
                        a_pol = rng.choice(c.n_actions, p=probs[s, i])
                        acts[s, i] = rng.choice([a_pol, a_rand], p=[1 - c.mu, c.mu])
                self.act_i32.copy_(torch.from_numpy(acts).to(self.dev))
                L.rcmarl_env_apply(self.pos[cur].data_ptr(), self.goal.data_ptr(), self.act_i32.data_ptr(), c.nrow, c.ncol,
                                   self.scale.data_ptr(), rp[0], rp[1], rp[2], rp[3], rp[4], self.cap, row,
                                   self.pos[nxt].data_ptr(), self.xs[nxt].data_ptr(), self.ret.data_ptr(), self.gpow[j],
                                   S, N, self.stream)
            self.cur = nxt
        self.B += c.max_ep_len
        self.ret_hist[ep_in_block].copy_(self.ret)
        self.episode += 1

    def rollout_block(self, n_eps):
        """n_eps (<= n_ep_fixed) episodes for all seeds, stepped TOGETHER (rng_mode 'device'): every network
        is frozen between update blocks and the Philox draws depend only on (seed, episode, step, agent), so
        the episodes are independent; max_ep_len launches instead of n_eps*max_ep_len, and the replay rows
        land exactly where the sequential loop (train_agents.py:46-80) would have put them."""
        c, L, S, N, EP = self.cfg, self.lib, self.S, self.N, self.EP
        assert c.rng_mode == "device" and n_eps <= c.n_ep_fixed
        self._reserve_rows(c.max_ep_len * n_eps)
        pin = None
        if not c.randomize_state:
            pin = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(np.asarray(self.initial_state, dtype=np.int32),
                                                                        (S, N, 2)))).to(self.dev)
        L.rcmarl_env_reset_episodes(self._p(pin), self.seeds_dev.data_ptr(), c.nrow, c.ncol, self.scale.data_ptr(),
                                    self.episode, self.posT[0].data_ptr(), self.xsT[0].data_ptr(), self.retT.data_ptr(),
                                    S, N, n_eps, EP, self.stream)
        if self.wide:      # start states are episode-minor xsT[S][2N][EP]: a feature-major layer-1 input
            with self._agent_window():
                self._value_wide(None, self.theta["critic"], "critic", self.w_v, n_eps, x=(self.xsT[0].data_ptr(), 2 * N * EP, 0, EP))
            if self.shard is not None:
                self._allgather_rows(self.w_v, 0, n_eps)
            self.est_hist[:n_eps].copy_(self.w_v[:, :, :n_eps].permute(2, 0, 1))
        else:
            L.rcmarl_value_rows_episodes(self.xsT[0].data_ptr(), self.theta["critic"].data_ptr(), self.est_hist.data_ptr(), S, N,
                                         n_eps, EP, HID, self.ldp["critic"], self.stream)
        rp = self._replay_ptrs()
        cur = 0
        for j in range(c.max_ep_len):
            nxt = 1 - cur
            L.rcmarl_rollout_step_episodes(self.xsT[cur].data_ptr(), self.posT[cur].data_ptr(), self.goal.data_ptr(),
                                           self.theta["actor"].data_ptr(), self.seeds_dev.data_ptr(), c.nrow, c.ncol,
                                           self.scale.data_ptr(), rp[0], rp[1], rp[2], rp[3], rp[4], self.cap, self.B,
                                           c.max_ep_len, self.posT[nxt].data_ptr(), self.xsT[nxt].data_ptr(),
                                           self.retT.data_ptr(), self.gpow[j], self.episode, j, c.mu, S, N, n_eps, EP, HID,
                                           c.n_actions, self.ldp["actor"], self.stream)
            cur = nxt
        self.B += c.max_ep_len * n_eps
        self.ret_hist[:n_eps].copy_(self.retT[:, :, :n_eps].permute(2, 0, 1))
        self.pos[self.cur].copy_(self.posT[cur][:, :, :, n_eps - 1])      # where the last episode ended
        self.episode += n_eps

    # ---- update block (train_agents.py:86-163) ---------------------------------------------
    def _x(self, key, row0=0):
        """(pointer, seed_stride) of replay tensor `key` starting at row `row0`."""
        w = self.rp[key].shape[2]
        return self.rp[key].data_ptr() + 4 * row0 * w, self.cap * w

    def _layer1(self, xkey, theta, net, B, row0=0, buf=None, wp_fresh=False, lattice=True):
        ptr, stride = self._x(xkey, row0)
        buf = self.a1t if buf is None else buf
        if lattice and self._lattice_ok(xkey, B, row0):
            g, L = self.lat_geom[xkey], self.lib
            wp = self.lat_wp_f[xkey]
            if not wp_fresh:
                L.rcmarl_w1_split(theta.data_ptr(), self.lat_alpha[xkey].data_ptr(), wp.data_ptr(), self.S, self.N,
                                  self.in_dim[net], HID, self.ldp[net], g.wp[0], g.wp[1], self.stream)
            L.rcmarl_layer1_forward_lattice(self.lat_kp[xkey].data_ptr(), g.kp[0], g.kp[1], wp.data_ptr(), g.wp[0],
                                            g.wp[1], theta.data_ptr(), buf.data_ptr(), self.S, self.N, B, self.in_dim[net],
                                            HID, self.ldp[net], self.ldb, self.stream)
            return
        self.lib.rcmarl_layer1_forward(ptr, stride, theta.data_ptr(), buf.data_ptr(), self.S, self.N, B,
                                       self.in_dim[net], HID, self.ldp[net], self.ldb, self.stream)

    def _local_fit(self, net, xkey, y, B, mask, partials=None):
        """5 full-batch SGD steps on the message copy (agents/resilient_CAC_agents.py:118,136)."""
        if self._sharded(net):
            y, mask = self._wv(y), self._wv(mask)
            with self._agent_window():
                return self._local_fit(net, xkey, y, B, mask, partials)
        if self.hid[net] != HID:
            return self._local_fit_wide(net, xkey, y, B, mask)
        L, S, N = self.lib, self.S, self.N
        msg = self.msg[net]
        if self._fit_as_chains(net, mask):
            # Small networks, many instances: the whole 5-step full-batch fit of one (seed, agent) network as ONE matrix-core
            # wavefront (the adversaries' mini-batch kernel with batch_size = B and no shuffle: rcmarl_minibatch_fit accumulates
            # the B rows in 32-row tiles and takes one SGD step per epoch).  The activations never leave the registers: no a1t
            # round trip and one launch instead of 20 -- 3.7 ms per consensus epoch for both networks of 512 seeds x 5 agents
            # against 7.4.  A single instance prefers the row-parallel kernels below (a chain is 470 dependent tile steps).
            ptr, stride = self._x(xkey)
            L.rcmarl_minibatch_fit(ptr, stride, msg.data_ptr(), self.coop_idx.data_ptr(), self.n_coop, y.data_ptr(), None, S, N, B,
                                   self.in_dim[net], HID, self.ldp[net], self.ldb, B, self.cfg.local_fit_steps, self.cfg.fast_lr,
                                   self.loss[net].data_ptr(), self.ovf_flags(("chain", net), S * max(self.n_coop, 1)).data_ptr(),
                                   self.stream)
            self.a1_cached[net] = False
            return
        a1 = self.a1net[net]
        ptr, stride = self._x(xkey)
        lat = self._lattice_ok(xkey, B, 0) and xkey in self.lat_ktp
        g = self.lat_geom[xkey] if lat else None
        dzp, wp = (self.lat_dzp_f[xkey], self.lat_wp_f[xkey]) if lat else (None, None)
        partials = self.partials if partials is None else partials
        wp_fresh = False
        for step in range(self.cfg.local_fit_steps):
            if not (step == 0 and self.a1_cached[net]):       # msg == live net: activations left by _consensus
                self._layer1(xkey, msg, net, B, buf=a1, wp_fresh=wp_fresh)
            if lat:
                L.rcmarl_mid_fit_lattice(a1.data_ptr(), msg.data_ptr(), y.data_ptr(), partials.data_ptr(),
                                         dzp.data_ptr(), g.dzp[0], g.dzp[1], S, N, B, self.in_dim[net], HID,
                                         self.ldp[net], self.ldb, self.ovf_flags(("mid", net), S * N + 1).data_ptr(), self.stream)
            else:
                L.rcmarl_mid_fit(a1.data_ptr(), msg.data_ptr(), y.data_ptr(), partials.data_ptr(), S, N, B,
                                 self.in_dim[net], HID, self.ldp[net], self.ldb, self.stream)
            L.rcmarl_small_sgd(partials.data_ptr(), msg.data_ptr(), mask.data_ptr(),
                               self.loss[net].data_ptr() if step == 0 else None, S, N, B, self.in_dim[net], HID,
                               self.ldp[net], self.cfg.fast_lr, self.stream)
            if lat:
                # the epilogue also leaves the bf16x3 split of the updated W1 for the NEXT step's forward -- except after
                # the last step, whose weights only travel as a message (252 / 378 MB of writes per fit at cfg 4)
                last = step == self.cfg.local_fit_steps - 1
                L.rcmarl_layer1_backward_sgd_lattice(self.lat_ktp[xkey].data_ptr(), g.ktp[0], g.ktp[1],
                                                     dzp.data_ptr(), g.dzp[0], g.dzp[1],
                                                     self.lat_alpha[xkey].data_ptr(), msg.data_ptr(), mask.data_ptr(), S, N,
                                                     B, self.in_dim[net], HID, self.ldp[net], self.cfg.fast_lr,
                                                     None if last else wp.data_ptr(), g.wp[0], g.wp[1], self.stream)
                wp_fresh = True           # the epilogue left the split of the updated W1 in lat_wp
            else:
                L.rcmarl_layer1_backward_sgd(ptr, stride, a1.data_ptr(), msg.data_ptr(), mask.data_ptr(), S, N, B,
                                             self.in_dim[net], HID, self.ldp[net], self.ldb, self.cfg.fast_lr, self.stream)
        self.a1_cached[net] = False

    def ovf_flags(self, key, n):
        """Out-of-range flag buffer of one call site (int32, zero at first use; the library's f16 kernels flag an agent / network
        whose operands leave the f16 range and their fp32 fix-up consumes the flag).  Owned by THIS engine, one per call site and
        stream: nothing is shared between engines, host threads or launches in flight (include/rcmarl.h)."""
        d = self.__dict__.setdefault("_ovf_flags", {})
        t = d.get(key)
        if t is None or t.numel() < n:
            t = d[key] = torch.zeros(int(n), dtype=torch.int32, device=self.dev)
        return t

    def _fit_as_chains(self, net, mask):
        """Local fits through rcmarl_minibatch_fit (one wavefront per network)?  RCMARL_FIT_CHAINS=1 / 0 forces it on / off; default:
        20-unit networks of at most 20 inputs, off the lattice path, the cooperative mask, and enough networks to fill the GPU."""
        if self.hid[net] != HID or self.in_dim[net] > 20 or self.lat_active or mask is not self.coop or self.n_coop == 0:
            return False
        if self.dev.type != "cuda" and os.environ.get("RCMARL_FIT_CHAINS") != "1":
            return False
        e = os.environ.get("RCMARL_FIT_CHAINS")
        if e is not None:
            return e not in ("0", "false")
        return self.S * self.n_coop >= 1024

    def _value(self, xkey, theta, net, out, B, row0=0, r_applied=None, scratch=None, gather=False, value_f32=False):
        """scratch: private activation buffer of a caller that may run beside the main stream (the adversaries); such a
        caller also stays off the lattice path, whose packed-operand scratch belongs to the main stream.
        gather (agent-sharded instance only): the caller needs the values of ALL agents, not just this rank's.
        value_f32: layers 2-3 on the fp32 vector-ALU kernel (the adversaries' callers, see below)."""
        if self._sharded(net) and scratch is None:
            th, o, r = self._wv(theta), self._wv(out), self._wv(r_applied)
            with self._agent_window():
                self._value(xkey, th, net, o, B, row0, r, value_f32=value_f32)
            if gather:
                self._allgather_rows(out, 0, B)
            return
        if self.hid[net] != HID:
            return self._value_wide(xkey, theta, net, out, B, row0, r_applied)
        buf = self.a1t if scratch is None else scratch
        self._layer1(xkey, theta, net, B, row0, buf=buf, lattice=scratch is None)
        # (a caller with a private scratch buffer is an adversary: its values feed a chain of 32-row mini-batch steps that amplifies
        # last-bit differences -- they stay on the fp32 vector-ALU kernel, as they stay off the lattice path)
        fn = self.lib.rcmarl_mid_value if (scratch is None and not value_f32) else self.lib.rcmarl_mid_value_f32
        fn(buf.data_ptr(), theta.data_ptr(), self._p(r_applied), self.cfg.gamma, out.data_ptr(),
           self.S, self.N, B, self.in_dim[net], HID, self.ldp[net], self.ldb, self.stream)

    def _k1(self, net, g_hid):
        """hidden-layer consensus of one network family: msg -> theta (cooperative agents, columns < g_hid)"""
        L, c = self.lib, self.cfg
        if self._windowed:          # agent-sharded instance: all-to-all, K1 on this rank's parameter columns, all-to-all back
            sc = self.shard.sc[net]
            assert g_hid == sc.P_hid
            sc.stream = self.stream
            sc.exchange(self.msg[net])
            sc.consensus()
            sc.gather(self.theta[net])
            return
        if self.k1_circulant:
            L.rcmarl_consensus_params_circulant(self.msg[net].data_ptr(), self.theta[net].data_ptr(), self.coop.data_ptr(),
                                                self.S, self.N, self.ldp[net], g_hid, c.d, c.H, None, None, self.stream)
        else:
            L.rcmarl_consensus_params(self.msg[net].data_ptr(), self.theta[net].data_ptr(), self.nbr.data_ptr(),
                                      self.coop.data_ptr(), self.S, self.N, self.ldp[net], g_hid, c.d, c.H, None, None,
                                      self.stream)

    def _cached_rows_ok(self, net, row0, nrows):
        """May values of `net` on replay rows row0..row0+nrows come from the activations the consensus step left in
        a1net[net]?  (live net unchanged since, rows are whole episodes of ours, 20-unit net)"""
        ep = self.cfg.max_ep_len
        return (self.td_shortcut and self.a1_cached[net] and self.rows_episode_aligned and self.hid[net] == HID and ep >= 2
                and row0 % ep == 0 and nrows % ep == 0)

    def _value_cached(self, net, out, row0, nrows, r_applied=None):
        """out[:, :, 0:nrows] = head(a1net[net][:, :, row0:row0+nrows])  [r_applied + gamma * that]"""
        if self._sharded(net):
            o, r = self._wv(out), self._wv(r_applied)
            with self._agent_window():
                self._value_cached(net, o, row0, nrows, r)
            return self._allgather_rows(out, 0, nrows)
        self.lib.rcmarl_mid_value(self.a1net[net].data_ptr() + 4 * row0, self.theta[net].data_ptr(), self._p(r_applied),
                                  self.cfg.gamma, out.data_ptr(), self.S, self.N, nrows, self.in_dim[net], HID,
                                  self.ldp[net], self.ldb, self.stream)

    def _value_next_cached(self, out, row0, nrows, r_applied, scratch):
        """Critic values of the NEXT states of rows row0..row0+nrows (out = V, or r_applied + gamma V).  Inside an episode
        the next state of row b IS the state of row b+1 (training/train_agents.py:66-80 appends s, ns step by step), so 19
        of 20 values come from the cached activations of the s rows shifted by one row; only the last step of every
        episode needs a forward pass of its own (f32-MFMA kernel on the gathered ns rows)."""
        c, L, S, N = self.cfg, self.lib, self.S, self.N
        ep, th = c.max_ep_len, self.theta["critic"]
        L.rcmarl_mid_value(self.a1net["critic"].data_ptr() + 4 * (row0 + 1), th.data_ptr(), self._p(r_applied), c.gamma,
                           out.data_ptr(), S, N, nrows - 1, self.in_c, HID, self.ldp["critic"], self.ldb, self.stream)
        nt = nrows // ep
        ns_w = self.rp["ns"].shape[2]
        if self._ns_term is None or self._ns_term.numel() < S * nt * ns_w:
            self._ns_term = torch.empty(S * nt * ns_w, dtype=torch.float32, device=self.dev)
        ns_term = self._ns_term
        nptr, nstride = self._x("ns", row0)
        L.rcmarl_gather_rows(nptr, nstride, ep - 1, ep, nt, ns_w, ns_term.data_ptr(), S, self.stream)
        if self.lat_active and self.shard is None and "s" in self.lat_wp_f and self.a1_cached["critic"]:
            # on the lattice GEMM: the live critic's layer-1 pieces are what the consensus step's forward left in the weight operand
            # (its hidden layers have not moved since), the rows are a subset of the ns rows _lattice_encode verified -- 60 us
            # instead of 283 for the f32 kernel, which reads all of W1 for 150 rows
            gt, gs = self.lat_geom_term, self.lat_geom["s"]
            L.rcmarl_lattice_encode(ns_term.data_ptr(), nt * self.in_c, self.lat_alpha["s"].data_ptr(), S, nt, self.in_c,
                                    self.lat_kp_term.data_ptr(), gt.kp[0], gt.kp[1], None, 0, 0, self.lat_flag_term.data_ptr(),
                                    self.stream)
            L.rcmarl_layer1_forward_lattice(self.lat_kp_term.data_ptr(), gt.kp[0], gt.kp[1], self.lat_wp_f["s"].data_ptr(), gs.wp[0],
                                            gs.wp[1], th.data_ptr(), self.a1t.data_ptr(), S, N, nt, self.in_c, HID,
                                            self.ldp["critic"], self.ldb, self.stream)
        else:
            L.rcmarl_layer1_forward(ns_term.data_ptr(), nt * self.in_c, th.data_ptr(), self.a1t.data_ptr(), S, N, nt, self.in_c, HID,
                                    self.ldp["critic"], self.ldb, self.stream)
        L.rcmarl_mid_value(self.a1t.data_ptr(), th.data_ptr(), None, c.gamma, scratch.data_ptr(), S, N, nt, self.in_c, HID,
                           self.ldp["critic"], self.ldb, self.stream)
        # back into the agent-major vector (r + gamma*v in fp32: multiply, then add, like the kernel): one small launch, no
        # framework kernels inside an update epoch (they kept larger instances out of hipGraph capture)
        L.rcmarl_scatter_values(scratch.data_ptr(), self._p(r_applied), c.gamma, out.data_ptr(), ep - 1, ep, nt, S, N, self.ldb,
                                self.stream)

    def _a2_rows_ok(self, row0, nrows):
        """wide critic: may the values of the next states come from the fp32 layer-2 activations the consensus step's forward pass
        left in w_a2?  (live hidden layers unchanged since, rows are whole episodes of ours)"""
        ep = self.cfg.max_ep_len
        return (self.td_shortcut and self.a2_cached and self.wide and self.rows_episode_aligned and ep >= 2
                and row0 % ep == 0 and nrows % ep == 0)

    def _value_cached_wide(self, out, row0, nrows):
        """out[:, :, 0:nrows] = live head on the cached fp32 layer-2 activations of rows row0..row0+nrows (all agents)"""
        if self._sharded("critic"):
            o = self._wv(out)
            with self._agent_window():
                self._value_cached_wide(o, row0, nrows)
            return self._allgather_rows(out, 0, nrows)
        self.lib.rcmarl_wide_head_value(self.w_a2.data_ptr() + 4 * row0, self.theta["critic"].data_ptr(), None, self.cfg.gamma,
                                        out.data_ptr(), self.S, self.N, nrows, self.in_c, self.hid["critic"], self.ldp["critic"],
                                        self.ldb, self.stream)

    def _value_next_cached_wide(self, out, row0, nrows, r_applied, scratch, gather=False):
        """_value_next_cached for a wide critic: inside an episode V(ns[b]) = V(s[b+1]) is the live head on the cached layer-2
        activations of row b+1 (the head moved since they were computed, the hidden layers did not); the last step of every episode
        gets a forward pass of its own on the gathered ns rows."""
        if self._sharded("critic"):             # this rank's agents (gather: the actor phase needs everybody's)
            o, r, sc = self._wv(out), self._wv(r_applied), self._wv(scratch)
            with self._agent_window():
                self._value_next_cached_wide(o, row0, nrows, r, sc)
            if gather:
                self._allgather_rows(out, 0, nrows)
            return
        c, L, S, N = self.cfg, self.lib, self.S, self.N
        ep, th, hid = c.max_ep_len, self.theta["critic"], self.hid["critic"]
        L.rcmarl_wide_head_value(self.w_a2.data_ptr() + 4 * (row0 + 1), th.data_ptr(), self._p(r_applied), c.gamma, out.data_ptr(),
                                 S, N, nrows - 1, self.in_c, hid, self.ldp["critic"], self.ldb, self.stream)
        nt = nrows // ep
        ns_w = self.rp["ns"].shape[2]
        if self._ns_term is None or self._ns_term.numel() < S * nt * ns_w:
            self._ns_term = torch.empty(S * nt * ns_w, dtype=torch.float32, device=self.dev)
        ns_term = self._ns_term
        nptr, nstride = self._x("ns", row0)
        L.rcmarl_gather_rows(nptr, nstride, ep - 1, ep, nt, ns_w, ns_term.data_ptr(), S, self.stream)
        self._value_wide(None, th, "critic", scratch, nt, x=(ns_term.data_ptr(), nt * self.in_c, 1, self.in_c))     # (overwrites w_a2)
        L.rcmarl_scatter_values(scratch.data_ptr(), self._p(r_applied), c.gamma, out.data_ptr(), ep - 1, ep, nt, S, N, self.ldb,
                                self.stream)

    def _td_target(self, B):
        """y = r_applied + gamma * V_critic(ns)   (agents/resilient_CAC_agents.py:114-115), all agents, all rows: from the
        activations the previous epoch's consensus step left behind where possible (one forward GEMM and one W1 split
        less per epoch), else by a forward pass over the ns rows."""
        y, r = self.ybuf["y_c"], self.ybuf["r_fit"]
        if self._cached_rows_ok("critic", 0, B):
            self._value_next_cached(y, 0, B, r, self.ybuf["v_next"])
        elif self._a2_rows_ok(0, B):
            self._value_next_cached_wide(y, 0, B, r, self.ybuf["v_next"])
        else:
            self._value("ns", self.theta["critic"], "critic", y, B, r_applied=r)

    td_shortcut = os.environ.get("RCMARL_TD_SHORTCUT", "1") not in ("0", "false")

    def _consensus(self, net, xkey, B, msg_all=None):
        """Phase II for one network family: hidden-layer consensus (K1), then estimate
        consensus + projection step of the output layer (K2+K3).
        msg_all: the message matrix whose rows the in_nodes table indexes (GLOBAL agents; default: self.msg[net])."""
        if self._sharded(net):
            # the neighbours' output layers (W3, b3 of their messages) come from all ranks; everything else is per agent
            o = self.P[net] - (self.hid[net] + 1)
            self._allgather_rows(self.msg[net], o, self.P[net])
            msg_all = self.msg[net]
            with self._agent_window():
                return self._consensus(net, xkey, B, msg_all)
        if self.hid[net] != HID:
            return self._consensus_wide(net, xkey, B, msg_all)
        L, S, N, c = self.lib, self.S, self.N, self.cfg
        msg_all = self.msg[net] if msg_all is None else msg_all
        self._k1(net, self.P[net] - (HID * 1 + 1))
        a1 = self.a1net[net]
        self._layer1(xkey, self.theta[net], net, B, buf=a1)
        L.rcmarl_consensus_head(a1.data_ptr(), self.theta[net].data_ptr(), msg_all.data_ptr(),
                                self.nbr.data_ptr(), self.coop.data_ptr(), self.partials.data_ptr(), None, S, N, B,
                                self.in_dim[net], HID, self.ldp[net], self.ldb, c.d, c.H, self.stream)
        self.a1_cached[net] = self.reuse_activations       # hidden layers of the live net do not move until the next K1
        L.rcmarl_head_apply(self.partials.data_ptr(), self.theta[net].data_ptr(), self.coop.data_ptr(), S, N, B,
                            self.in_dim[net], HID, self.ldp[net], self.stream)

    def _timed(self, key, t0):
        if self.profile_phases:
            self.sync()
            t1 = time.perf_counter()
            self.timers[key] += t1 - t0
            return t1
        return t0

    profile_phases = False
    reuse_activations = True

    def _epoch_body(self, B, t0):
        """One update epoch: I) local fits of the TR and critic messages, II) resilient consensus."""
        # I) local fits of TR and critic on a copy (= the transmitted message); live nets untouched
        if self.shard is not None and not self.shard.shard_tr:
            self.msg["tr"].copy_(self.theta["tr"])
        with self._agent_window():
            if self.shard is None or self.shard.shard_tr:
                self.msg["tr"].copy_(self.theta["tr"])
            self.msg["critic"].copy_(self.theta["critic"])
        # TD target first (it depends on the live critic only), so the adversaries' message generators --
        # one latency-bound workgroup per (seed, adversary) -- can run on a side stream UNDER the cooperative
        # agents' local fits: they touch disjoint parameter rows and meet again at the consensus step
        self._td_target(B)
        join = self._adversary_messages_async(B)
        if self.dev.type == "cuda" and self._fit_as_chains("tr", self.coop) and self._fit_as_chains("critic", self.coop):
            # both fits are chains of one wavefront per network (disjoint buffers): side by side they fill the GPU's wavefront
            # slots in 3.3 rounds instead of 2 + 2
            if getattr(self, "chain_stream", None) is None:
                self.chain_stream = torch.cuda.Stream(device=self.dev)
            main = torch.cuda.current_stream(self.dev)
            fork = torch.cuda.Event()
            fork.record(main)
            self.chain_stream.wait_event(fork)
            with torch.cuda.stream(self.chain_stream):
                self._local_fit("tr", "sa", self.ybuf["r_fit"], B, self.coop)
                tr_done = torch.cuda.Event()
                tr_done.record(self.chain_stream)
            self._local_fit("critic", "s", self.ybuf["y_c"], B, self.coop)
            main.wait_event(tr_done)
        else:
            self._local_fit("tr", "sa", self.ybuf["r_fit"], B, self.coop)
            self._local_fit("critic", "s", self.ybuf["y_c"], B, self.coop)
        if join is not None:
            torch.cuda.current_stream(self.dev).wait_event(join)
        t0 = self._timed("phase1", t0)
        # II) resilient consensus (cooperative agents)
        self._consensus("critic", "s", B)
        self._consensus("tr", "sa", B)
        return self._timed("phase2", t0)

    # ---- one epoch as a hipGraph (single instances: an epoch is ~60 launches of a few microseconds each) --------------------
    # From the second epoch of a block on, every epoch issues the SAME launches with the same arguments (the first differs:
    # no cached activations yet), and its Python-side state (a1_cached) ends where it began.  That sequence is captured once per
    # (B, path flags) on torch's capture stream -- the library launches on torch's current stream and calls nothing that a
    # capture forbids (no allocation, no synchronisation) -- and replayed for the remaining epochs and the following blocks.
    # RCMARL_GRAPH=1 / 0 forces it on / off; default: on for small instances (S * N <= 256 agent-networks per launch), where
    # the launches, not the kernels, set the block time.  Never with an agent-sharded instance (collectives), a wide critic, phase
    # profiling or a timing wrapper around the library; with Greedy / Malicious agents only when their fits run inline (see below).
    def _graph_wanted(self):
        e = os.environ.get("RCMARL_GRAPH")
        if e is not None:
            want = e not in ("0", "false")
        else:
            want = self.S * self.N <= 256
        if not want or self.dev.type != "cuda" or self.shard is not None or self.wide or self.profile_phases:
            return False
        if hasattr(self, "adv") and self.adv.fit and not self.adv._multi_ok() and \
                os.environ.get("RCMARL_ADV_ASYNC", "1") not in ("0", "false"):
            # (round 5: with all fits of an epoch in ONE rcmarl_minibatch_fit_multi launch -- AdversaryPath._multi_ok -- there are no
            # side streams: the chains start together AND the epoch is captured.  What follows concerns the per-family launches.)
            # Greedy / Malicious agents: their three 940-step chains run side by side on three streams, and capturing that fork/join
            # SEGFAULTS inside the HIP runtime of this ROCm (7.2; reproduced with tools/diag_graph_adversaries.py, round 4).  Inline on
            # the capture stream (RCMARL_ADV_ASYNC=0) the capture works -- the shuffle-call counter lives on the device for that,
            # engine_adversaries._draw -- but then the chains run one after the other (3 x 2.7 ms per epoch instead of 2.7 ms) and that
            # costs more than the ~50 launches the graph saves.  A block of BASELINE configs[1] as one instance is 10 epochs x 2.7 ms
            # of chain latency + rollout: its 32 ms are at that floor, graph or not.
            return False
        return not getattr(self.lib, "enabled", False)            # bench.py's per-kernel timing wrapper records events per launch

    def _epoch(self, B, epoch, t0):
        if epoch == 0 or not self._graph_wanted() or not (self.a1_cached["critic"] and self.a1_cached["tr"]):
            return self._epoch_body(B, t0)
        key = (B, self.lat_active, bool(self.td_shortcut), bool(self.rows_episode_aligned), bool(self.k1_circulant),
               bool(self.reuse_activations), self.cap, self.lib.rcmarl_lattice_f16_mode(), os.environ.get("RCMARL_LAT_W8"),
               os.environ.get("RCMARL_MIDFIT"),
               # the adversaries' launch form is re-read every epoch (AdversaryPath._multi_ok, phase1): a captured epoch belongs to one form
               bool(hasattr(self, "adv") and self.adv._multi_ok()), os.environ.get("RCMARL_ADV_ASYNC", "1"))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 8:                             # growing replay buffer: B changes every block until steady state
                self._graphs.pop(next(iter(self._graphs)))
            g = torch.cuda.CUDAGraph()
            calls0 = list(self.adv.calls) if hasattr(self, "adv") else None
            # No cyclic garbage collection while the stream is capturing: a collected engine of an earlier run takes its hipGraphs
            # (and pooled tensors) with it, and destroying those during a capture aborts the process (seen in the GPU suite, round 4).
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()
            captured = False
            try:
                with torch.cuda.graph(g):
                    self._epoch_body(B, t0)
                captured = True
            finally:
                if gc_was_on:
                    gc.enable()
                if calls0 is not None:                             # the capture advanced the host's shuffle-call counter by one epoch
                    g.rcmarl_draws = self.adv.calls[0] - calls0[0]     # ... without running anything: take it back (also when the
                    self.adv.calls = calls0                        # capture raised), the replay below counts
                    if self.adv.base_dev is not None:              # (None: no adversary of this instance fits -- only Faulty ones)
                        try:                                       # the device-side counter: a stream whose capture just failed may
                            self.adv.base_dev.fill_(int(calls0[0]))    # refuse the launch -- that must not mask the capture's own error
                        except Exception:
                            if captured:
                                raise
            self._graphs[key] = g
            self.graph_captures += 1
        g.replay()
        if hasattr(self, "adv"):
            self.adv.replayed(getattr(g, "rcmarl_draws", 0))
        self.graph_replays += 1
        return t0

    def update_block(self):
        c, L, S, N, B = self.cfg, self.lib, self.S, self.N, self.B
        assert B >= self.n_last
        self._shards_synced = False               # (agent-sharded instance: other ranks' rows go stale from here on)
        for lab in c.agent_label:
            if lab in (GREEDY, MALICIOUS, FAULTY):
                self._require_adversary_support()
        t0 = time.perf_counter()
        if self.profile_phases:
            self.sync()
            t0 = time.perf_counter()
        self.a1_cached["critic"] = self.a1_cached["tr"] = self.a2_cached = False       # new replay rows
        self._lattice_encode(B)
        rptr, rstride = self._x("r")
        L.rcmarl_team_reward(rptr, rstride, self.coop.data_ptr(), max(self.n_coop, 1), self.rcoop.data_ptr(), S, N, B,
                             self.ldb, self.stream)
        L.rcmarl_gather_agent_major(rptr, rstride, self.rcoop.data_ptr(), self.fit_mode.data_ptr(),
                                    self.ybuf["r_fit"].data_ptr(), S, N, B, self.ldb, self.stream)
        for epoch in range(c.n_epochs):
            t0 = self._epoch(B, epoch, t0)
        # III) actor update on the last n_ep_fixed episodes
        self._actor_update(B)
        self._adversary_actor_updates(B)
        t0 = self._timed("phase3", t0)
        # IV) trim the replay buffer (after the update, train_agents.py:158-163)
        if B > c.buffer_size:
            q = B - c.buffer_size
            if q % c.max_ep_len:                   # the buffer no longer starts at an episode boundary
                self.rows_episode_aligned = False
            for k in self.rp:
                self.rp[k][:, :c.buffer_size] = self.rp[k][:, q:B].clone()
            self.B = c.buffer_size
        self.timers["blocks"] += 1

    def _actor_update(self, B):
        c, L, S, N = self.cfg, self.lib, self.S, self.N
        nl = self.n_last
        row0 = B - nl
        # team-average TD error (agents/resilient_CAC_agents.py:95-98): the last consensus step left the layer-1
        # activations of the live TR net and critic on every row, so no forward GEMM is needed here
        if self._cached_rows_ok("tr", row0, nl):
            self._value_cached("tr", self.ybuf["v_tr"], row0, nl)
        else:
            self._value("sa", self.theta["tr"], "tr", self.ybuf["v_tr"], nl, row0, gather=True)
        if self._cached_rows_ok("critic", row0, nl):
            self._value_next_cached(self.ybuf["v_next"], row0, nl, None, self.ybuf["delta"])
            self._value_cached("critic", self.ybuf["v_cur"], row0, nl)
        elif self._a2_rows_ok(row0, nl):
            # wide critic: both values from the fp32 layer-2 activations the last consensus step left behind (V(s) first: the forward
            # pass over the episodes' last next-state rows inside _value_next_cached_wide re-uses that buffer)
            self._value_cached_wide(self.ybuf["v_cur"], row0, nl)
            self._value_next_cached_wide(self.ybuf["v_next"], row0, nl, None, self.ybuf["delta"], gather=True)
        else:
            self._value("ns", self.theta["critic"], "critic", self.ybuf["v_next"], nl, row0, gather=True)
            self._value("s", self.theta["critic"], "critic", self.ybuf["v_cur"], nl, row0, gather=True)
        L.rcmarl_td_error(self.ybuf["v_tr"].data_ptr(), self.ybuf["v_next"].data_ptr(), self.ybuf["v_cur"].data_ptr(),
                          c.gamma, self.ybuf["delta"].data_ptr(), S * N * self.ldb, self.stream)
        aptr, astride = self._x("a", row0)
        L.rcmarl_gather_agent_major(aptr, astride, None, None, self.ybuf["act_t"].data_ptr(), S, N, nl, self.ldb, self.stream)
        self.adam_t += 1
        b1, b2, eps = 0.9, 0.999, 1e-7
        alpha = float(np.float32(c.slow_lr * math.sqrt(1.0 - b2 ** self.adam_t) / (1.0 - b1 ** self.adam_t)))
        omb1, omb2, epsf = float(np.float32(1 - b1)), float(np.float32(1 - b2)), float(np.float32(eps))
        self._layer1("s", self.theta["actor"], "actor", nl, row0)
        L.rcmarl_mid_actor(self.a1t.data_ptr(), self.theta["actor"].data_ptr(), self.ybuf["act_t"].data_ptr(),
                           self.ybuf["delta"].data_ptr(), self.partials.data_ptr(), S, N, nl, self.in_c, HID, c.n_actions,
                           self.ldp["actor"], self.ldb, self.stream)
        L.rcmarl_small_adam(self.partials.data_ptr(), self.theta["actor"].data_ptr(), self.adam_m.data_ptr(),
                            self.adam_v.data_ptr(), self.coop.data_ptr(), self.loss["actor"].data_ptr(), S, N, nl, self.in_c,
                            HID, c.n_actions, self.ldp["actor"], alpha, omb1, omb2, epsf, self.stream)
        sptr, sstride = self._x("s", row0)
        L.rcmarl_layer1_backward_adam(sptr, sstride, self.a1t.data_ptr(), self.theta["actor"].data_ptr(),
                                      self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.coop.data_ptr(), S, N, nl,
                                      self.in_c, HID, self.ldp["actor"], self.ldb, alpha, omb1, omb2, epsf, self.stream)

    # adversaries (agents/adversarial_CAC_agents.py) are handled by engine_adversaries.py
    def _require_adversary_support(self):
        if not hasattr(self, "adv"):
            from . import engine_adversaries
            engine_adversaries.attach(self)

    def _adversary_messages(self, B):
        if hasattr(self, "adv"):
            self.adv.phase1(B)

    def _adversary_messages_async(self, B):
        """adv.phase1 on the side stream (GPU) -> the event to wait for before the consensus step; inline otherwise."""
        if not hasattr(self, "adv"):
            return None
        if self.dev.type != "cuda" or self.wide or os.environ.get("RCMARL_ADV_ASYNC", "1") in ("0", "false"):
            self.adv.phase1(B)                # (a wide critic's value / fit scratch is shared with the main stream: inline)
            return None
        if getattr(self, "adv_stream", None) is None:
            self.adv_stream = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(self.adv_stream):
            self.adv_stream.wait_event(fork)
            self.adv.phase1(B)
            join = torch.cuda.Event()
            join.record(self.adv_stream)
        return join

    def _adversary_actor_updates(self, B):
        if hasattr(self, "adv"):
            self.adv.actor_updates(B)

    # ---- whole training loop ----------------------------------------------------------------
    def run_block(self):
        """n_ep_fixed episodes of rollout followed by one update block.  Returns the
        per-episode logs of the block: (team_returns[E][S], adv_returns[E][S], est_returns[E][S])."""
        c = self.cfg
        t0 = time.perf_counter()
        if self.profile_phases:
            self.sync()
            t0 = time.perf_counter()
        if c.rng_mode == "device":
            self.rollout_block(c.n_ep_fixed)
        else:
            for e in range(c.n_ep_fixed):
                self.rollout_episode(e)
        self._timed("rollout", t0)
        self.update_block()
        return self.episode_logs(c.n_ep_fixed)

    def episode_logs(self, n_eps):
        """Episode summaries exactly as train_agents.py:168-180 computes them (float64
        sequential sums over agents; np.mean of the float32 start-state values)."""
        ret = self.ret_hist[:n_eps].detach().cpu().numpy()
        est = self.est_hist[:n_eps].detach().cpu().numpy()
        coop = self.coop_np.astype(bool)
        E, S, N = ret.shape
        n_coop = self.n_coop
        # The reference accumulates  t += ret_i / n_coop  agent by agent in float64 (train_agents.py:168-172): np.cumsum is
        # that strictly sequential sum (np.sum would be pairwise), so the values are bit-identical to the Python loop --
        # which cost 25 ms per block at 16 seeds x 256 agents with the GPU idle behind it.
        team = np.cumsum(ret[:, :, coop] / n_coop, axis=-1)[:, :, -1] if n_coop else np.zeros((E, S))
        adv = np.cumsum(ret[:, :, ~coop] / (N - n_coop), axis=-1)[:, :, -1] if n_coop < N else np.zeros((E, S))
        # np.mean of the cooperative agents' float32 start-state values (:173), row by row
        estm = np.ascontiguousarray(est[:, :, coop]).mean(axis=-1).astype(np.float64) if n_coop else np.full((E, S), np.nan)
        return team, adv, estm

    _diverged_warned = False
    _shards_synced = False

    def _warn_if_diverged(self):
        """The reference's plain-SGD local fits diverge to NaN when fast_lr is too large for the input width
        (e.g. 0.01 at N = 256); it would carry on silently.  One warning per engine, one reduction per block.  In an
        agent-sharded instance a rank only sees its own agents' rows: the verdict is all-reduced (MIN) over the shard's
        communicator, so every rank warns (or none) -- a COLLECTIVE there."""
        if self.pk is not None and int(self.pk.ovf.item()) != 0:
            # a critic operand left the f16 range of the packed-operand path (|a1| > 1015, |W2| or |W2 W3| > 63, |dz1| > 254: a fit that
            # is blowing up): its pieces were carried clipped.  From here on this engine takes the rcmarl_dense_* path, whose kernels
            # recompute such tiles in fp32 -- what the reference's arithmetic would do with the same (diverging) numbers.
            import warnings
            warnings.warn("rcmarl_amd: a wide-critic operand left the f16 range of the packed-operand path (training is diverging: "
                          "lower fast_lr); falling back to the dense path that recomputes out-of-range tiles in fp32", RuntimeWarning)
            self.pk = None
            self.a1_cached["critic"] = self.a2_cached = False
        if self._diverged_warned:
            return
        finite = all(bool(torch.isfinite(self.theta[k]).all().item()) for k in ("critic", "tr", "actor"))
        if self.shard is not None:
            flag = torch.tensor([1.0 if finite else 0.0], dtype=torch.float32, device=self.dev)
            parts = [torch.empty_like(flag) for _ in range(self.shard.world)]
            self.shard.comm.all_gather(parts, flag)
            finite = all(bool(p.item() > 0.5) for p in parts)
        if not finite:
            import warnings
            warnings.warn("rcmarl_amd: non-finite network weights after an update block (training diverged; "
                          "lower fast_lr -- the reference's 0.01 is unstable beyond ~64 agents)", RuntimeWarning)
            self._diverged_warned = True

    def train(self, n_episodes):
        """Full loop; returns dict of per-episode arrays [n_episodes][S]."""
        c = self.cfg
        logs = {"True_team_returns": [], "True_adv_returns": [], "Estimated_team_returns": []}
        done = 0
        while done < n_episodes:
            n = min(c.n_ep_fixed, n_episodes - done)
            if n == c.n_ep_fixed:
                team, adv, est = self.run_block()
                self._warn_if_diverged()
            else:                               # trailing episodes without an update (t % n_ep_fixed never hits)
                if c.rng_mode == "device":
                    self.rollout_block(n)
                else:
                    for e in range(n):
                        self.rollout_episode(e)
                team, adv, est = self.episode_logs(n)
            logs["True_team_returns"].append(team)
            logs["True_adv_returns"].append(adv)
            logs["Estimated_team_returns"].append(est)
            done += n
        self.sync_shards()
        if not done:                            # n_episodes = 0: the reference's loop body never runs (train_agents.py:46)
            return {k: np.zeros((0, self.S)) for k in logs}
        return {k: np.concatenate(v, axis=0) for k, v in logs.items()}
