"""Multi-GPU layer: independent seeds are sharded over the ranks (one process per
GPU); the ONLY collective is an all-reduce (sum) of the return-curve
accumulators -- RCCL over xGMI on GPUs (backend "nccl"), gloo in CPU tests.

The reference has no counterpart: it launches one SGE job per seed and averages
the curves offline (plot_results.py:39).  Seeds share nothing, so there is no
data-path collective; the all-reduce is ~tens of KB and latency-bound.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_seeds(all_seeds, rank, world):
    """Round-robin seed -> rank assignment."""
    return [s for k, s in enumerate(all_seeds) if k % world == rank]


def allreduce_curves(local_sums, n_local_seeds, device=None, sq_sums=None):
    """local_sums: [E, C] per-episode sums over this rank's seeds (float64).
    Returns the across-all-seeds mean curve [E, C] (and std if sq_sums given),
    identical on every rank.  Works without an initialised process group (world=1)."""
    local_sums = np.asarray(local_sums, dtype=np.float64)
    parts = [local_sums.ravel(), np.asarray([float(n_local_seeds)])]
    if sq_sums is not None:
        parts.insert(1, np.asarray(sq_sums, dtype=np.float64).ravel())
    buf = torch.from_numpy(np.concatenate(parts))
    if device is not None:
        buf = buf.to(device)
    if dist.is_available() and dist.is_initialized():            # also with ONE rank: the same RCCL call path
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    n = out[-1]
    k = local_sums.size
    mean = (out[:k] / n).reshape(local_sums.shape)
    if sq_sums is None:
        return mean
    var = np.maximum(out[k:2 * k].reshape(local_sums.shape) / n - mean ** 2, 0.0)
    return mean, np.sqrt(var)


# ---------------------------------------------------------------------------------------------------------------
# C2: ONE instance sharded over the ranks (BASELINE configs[4]: 1024 agents x 512-unit critic)
#
# Phase I (local fits) is independent per AGENT, phase II-b (hidden-layer consensus, K1) is independent per parameter
# COLUMN: every output column needs all d neighbour rows of that column and nothing else.  So an agent-sharded phase I
# followed by a column-sharded K1 needs exactly one exchange each way -- an all-to-all transpose of the message matrix
# [N, P_hid] (each rank sends the (its agents) x (peer's columns) block to every peer) and the reverse transpose of the
# aggregated hidden parameters.  This replaces the reference's in-process gather
# `[critic_weights[i] for i in in_nodes[node]]` (training/train_agents.py:129-130).  On xGMI the all-to-all is direct
# (7 links per GPU used in parallel), not a ring: cfg 5 sends 7 x 84 MB per GPU per net per epoch.
# K1 per column is the SAME kernel launch on a narrower matrix, so the sharded result equals the unsharded one bit for
# bit (tests/test_sharded_consensus_gloo.py).
def agent_range(n_agents, rank, world):
    """Contiguous block of agents owned by `rank` (phase I / III shard)."""
    base, rem = divmod(int(n_agents), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def column_ranges(n_cols, world, align=64):
    """`world` contiguous column ranges covering [0, n_cols), boundaries on multiples of `align` (K1 walks 64-column
    tiles and 16-byte loads want aligned rows); trailing ranks may get an empty range when n_cols is small."""
    tiles = (int(n_cols) + align - 1) // align
    out = []
    for r in range(world):
        base, rem = divmod(tiles, world)
        t0 = r * base + min(r, rem)
        t1 = t0 + base + (1 if r < rem else 0)
        out.append((min(t0 * align, n_cols), min(t1 * align, n_cols)))
    return out


class TorchComm:
    """The two collectives of the sharded instance over a torch.distributed group (RCCL on GPUs, gloo in CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_gather(self, recv_list, send):
        dist.all_gather(recv_list, send, group=self.group)

    def all_to_all_single(self, recv, send, out_sizes, in_sizes):
        dist.all_to_all_single(recv, send, output_split_sizes=out_sizes, input_split_sizes=in_sizes, group=self.group)


class ThreadComm:
    """The same two collectives among THREADS of one process, each driving its own engine on the same device (same
    stream, so plain copies are ordered): lets a single-GPU box run the sharded instance's kernels on real hardware
    (tests/test_sharded_engine_gpu.py).  make(world) returns one endpoint per rank."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.group, self._sh = rank, world, None, shared

    @staticmethod
    def make(world):
        import threading
        shared = {"slot": [None] * world, "barrier": threading.Barrier(world, timeout=300)}   # a failed peer breaks it
        return [ThreadComm(r, world, shared) for r in range(world)]

    def all_gather(self, recv_list, send):
        sh = self._sh
        sh["slot"][self.rank] = send
        sh["barrier"].wait()
        for r in range(self.world):
            recv_list[r].copy_(sh["slot"][r])
        sh["barrier"].wait()

    def all_to_all_single(self, recv, send, out_sizes, in_sizes):
        sh = self._sh
        sh["slot"][self.rank] = (send, list(in_sizes))
        sh["barrier"].wait()
        o = 0
        for r in range(self.world):
            peer, sizes = sh["slot"][r]
            start = sum(sizes[:self.rank])
            recv[o:o + out_sizes[r]].copy_(peer[start:start + sizes[self.rank]])
            o += out_sizes[r]
        sh["barrier"].wait()


class ShardedConsensus:
    """Column-sharded K1 for one network family of ONE instance whose agents are sharded over the ranks.

        sc = ShardedConsensus(lib, S, N, P_hid, d, H, in_nodes, coop, device)        # after init_process_group
        msg_cols = sc.exchange(msg_local)          # [S][N_loc][ldp]  ->  [S][N][ldc]   (all-to-all #1)
        sc.consensus(msg_cols)                     # K1 on this rank's columns -> sc.theta_cols
        sc.gather(theta_local)                     # [S][N][ldc] -> columns < P_hid of [S][N_loc][ldp]   (all-to-all #2)

    Data movement: ONE pack pass (rcmarl_copy3d, csrc/shard_pack.hip) per direction and nothing else.  The sender of
    all-to-all #1 packs its (agents x peer's columns) boxes with the RECEIVER's row stride, so with one seed (the
    sharded instance) the blocks land in place in msg_cols; all-to-all #2 sends row ranges of theta_cols as they lie
    (contiguous with one seed) and the receiver scatters them into its parameter rows in one masked pass.  (Round 2 did
    .contiguous() + torch.cat + slice-assign / torch.where: three passes over a 5.4 GB matrix per net and epoch at
    BASELINE configs[4].)  Several seeds take one extra staging pass on the strided side.

    Rows of non-cooperative agents are never aggregated (agents/resilient_CAC_agents.py is the cooperative agent's
    class); gather() leaves them as they are.  circulant: None = decide from the graph (RCMARL_K1_CIRC=0 forces the
    general kernel), True/False = the caller's decision (the engine passes its own, so that a sharded run takes the same
    K1 kernel as the unsharded one it is compared with).  force_collectives: run the all-to-all even at world size 1
    (the one-rank RCCL test on a single-GPU box)."""

    def __init__(self, lib, S, N, P_hid, d, H, in_nodes, coop, device, stream=None, group=None, comm=None, circulant=None,
                 force_collectives=False):
        import os
        self.lib, self.S, self.N, self.P_hid, self.d, self.H = lib, int(S), int(N), int(P_hid), int(d), int(H)
        self.dev, self.stream, self.group = torch.device(device), stream, group
        self.comm = TorchComm(group) if comm is None else comm
        self.world, self.rank = self.comm.world, self.comm.rank
        self.force = bool(force_collectives)
        self.a_lo, self.a_hi = agent_range(N, self.rank, self.world)
        self.agents = [agent_range(N, r, self.world) for r in range(self.world)]
        self.cols = column_ranges(P_hid, self.world)
        self.c_lo, self.c_hi = self.cols[self.rank]
        self.width = self.c_hi - self.c_lo
        self.ldcs = [max(64, (c1 - c0 + 63) // 64 * 64) for (c0, c1) in self.cols]       # every rank's row stride
        self.ldc = self.ldcs[self.rank]
        nodes = np.asarray(in_nodes, dtype=np.int32)
        is_circ = bool(all(list(nodes[i]) == [(i + k) % N for k in range(d)] for i in range(N)) and
                       lib.rcmarl_consensus_params_circulant_supported(N, d, H) == 1)
        if circulant is None:
            circulant = os.environ.get("RCMARL_K1_CIRC", "1") not in ("0", "false")
        self.circulant = bool(circulant) and is_circ
        self.nbr = torch.tensor(nodes, dtype=torch.int32, device=self.dev)
        self.coop = torch.tensor(np.asarray(coop, dtype=np.int32), dtype=torch.int32, device=self.dev)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.msg_cols = torch.zeros(self.S, self.N, self.ldc, **f32)
        self.theta_cols = torch.zeros(self.S, self.N, self.ldc, **f32)
        n_loc = self.a_hi - self.a_lo
        # boxes this rank sends in #1 / receives in #2: [S][n_loc][ldc_r] per peer r (pad columns stay zero)
        self._box_sizes = [self.S * n_loc * l for l in self.ldcs]
        self._boxes = torch.zeros(sum(self._box_sizes), **f32)
        # row blocks this rank receives in #1 / sends in #2: [S][n_r][ldc] per peer r; with one seed they ARE msg_cols /
        # theta_cols (peer r's agents are rows a0_r..a1_r), otherwise a staging buffer
        self._blk_sizes = [self.S * (a1 - a0) * self.ldc for (a0, a1) in self.agents]
        self._stage = None if self.S == 1 else torch.zeros(sum(self._blk_sizes), **f32)
        self.passes = 0                           # pack / unpack launches so far (tests count them)

    # -- the two copy shapes ------------------------------------------------------------------------------------
    def _copy(self, src, src_batch, ld_src, dst, dst_batch, ld_dst, rows, cols, mask=None):
        if rows > 0 and cols > 0:
            self.lib.rcmarl_copy3d(src, src_batch, ld_src, dst, dst_batch, ld_dst, self.S, rows, cols,
                                   None if mask is None else mask.data_ptr(), self.stream)
            self.passes += 1

    def _box_ptrs(self):
        off, out = 0, []
        for n in self._box_sizes:
            out.append(self._boxes.data_ptr() + 4 * off)
            off += n
        return out

    def _a2a(self, recv, send, out_sizes, in_sizes):
        if self.world == 1 and not self.force:
            recv.copy_(send)                      # one rank, no process group: the self-exchange is a copy
        else:
            self.comm.all_to_all_single(recv, send, out_sizes, in_sizes)

    def exchange(self, msg_local):
        """msg_local [S][N_loc][ldp] (this rank's agents, all columns) -> self.msg_cols [S][N][ldc] (all agents, this
        rank's columns)."""
        n_loc = self.a_hi - self.a_lo
        assert msg_local.shape[0] == self.S and msg_local.shape[1] == n_loc and msg_local.is_contiguous()
        ldp = msg_local.shape[2]
        for r, ((c0, c1), dst) in enumerate(zip(self.cols, self._box_ptrs())):       # pack: strided column box -> contiguous
            self._copy(msg_local.data_ptr() + 4 * c0, n_loc * ldp, ldp, dst, n_loc * self.ldcs[r], self.ldcs[r], n_loc, c1 - c0)
        recv = self.msg_cols.view(-1) if self.S == 1 else self._stage
        self._a2a(recv, self._boxes, self._blk_sizes, self._box_sizes)
        if self.S > 1:                                                                # [S][n_r][ldc] blocks -> [S][N][ldc]
            off = 0
            for (a0, a1), n in zip(self.agents, self._blk_sizes):
                self._copy(self._stage.data_ptr() + 4 * off, (a1 - a0) * self.ldc, self.ldc,
                           self.msg_cols.data_ptr() + 4 * a0 * self.ldc, self.N * self.ldc, self.ldc, a1 - a0, self.ldc)
                off += n
        return self.msg_cols

    def consensus(self, msg_cols=None):
        """K1 (agents/resilient_CAC_agents.py:142-166) on this rank's columns; same launch as the unsharded step."""
        from .capi import RcmarlError
        msg = self.msg_cols if msg_cols is None else msg_cols
        if self.width == 0:
            return self.theta_cols
        if self.circulant:
            try:
                self.lib.rcmarl_consensus_params_circulant(msg.data_ptr(), self.theta_cols.data_ptr(), self.coop.data_ptr(),
                                                           self.S, self.N, self.ldc, self.width, self.d, self.H, None, None,
                                                           self.stream)
                return self.theta_cols
            except RcmarlError as e:              # e.g. the 64-bit cooperation-mask guard: the general kernel serves any shape
                if "UNSUPPORTED" not in str(e):
                    raise
                self.circulant = False
        self.lib.rcmarl_consensus_params(msg.data_ptr(), self.theta_cols.data_ptr(), self.nbr.data_ptr(), self.coop.data_ptr(),
                                         self.S, self.N, self.ldc, self.width, self.d, self.H, None, None, self.stream)
        return self.theta_cols

    def gather(self, theta_local):
        """self.theta_cols [S][N][ldc] -> columns < P_hid of theta_local [S][N_loc][ldp], cooperative agents only."""
        n_loc = self.a_hi - self.a_lo
        assert theta_local.shape[0] == self.S and theta_local.shape[1] == n_loc and theta_local.is_contiguous()
        ldp = theta_local.shape[2]
        if self.S == 1:
            send = self.theta_cols.view(-1)
        else:                                                                         # [S][N][ldc] -> [S][n_r][ldc] blocks
            off = 0
            for (a0, a1), n in zip(self.agents, self._blk_sizes):
                self._copy(self.theta_cols.data_ptr() + 4 * a0 * self.ldc, self.N * self.ldc, self.ldc,
                           self._stage.data_ptr() + 4 * off, (a1 - a0) * self.ldc, self.ldc, a1 - a0, self.ldc)
                off += n
            send = self._stage
        self._a2a(self._boxes, send, self._box_sizes, self._blk_sizes)
        mine = self.coop[self.a_lo:self.a_hi]
        for r, ((c0, c1), src) in enumerate(zip(self.cols, self._box_ptrs())):       # unpack: scatter into the parameter rows
            self._copy(src, n_loc * self.ldcs[r], self.ldcs[r], theta_local.data_ptr() + 4 * c0, n_loc * ldp, ldp, n_loc,
                       c1 - c0, mask=mine)
        return theta_local
