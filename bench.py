#!/usr/bin/env python
"""bench.py -- RPBCAC training throughput on MI355X (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (it re-executes
under torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1); a mismatch between --gpus and the
ranks that actually joined aborts -- the line never reports a GPU count it did not run on.

One "step" = one training block of the reference loop (training/train_agents.py:
46-163): n_ep_fixed=50 episodes x max_ep_len=20 environment steps of rollout for
every seed and agent, followed by the full update block (10 epochs of local
fits + resilient consensus, the actor step, the replay trim).  Nothing is
skipped inside the timed region.  Data: synthetic (random goals, Glorot-init
networks, on-device grid-world); arithmetic fp32 like the reference.

Prints ONE JSON line (rank 0): metric = agent-steps/s (whole loop), plus
consensus-updates/s, the per-kernel breakdown, `roofline` for the dominant
kernel, `roofline_consensus` for the consensus kernel, `cpu_baseline`, and (N=1) `extra`: short runs of the
other workloads in the same process, among them the north-star target shape whose consensus-kernel roofline is
`roofline_consensus_target`.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X dense fp32 (vector == fp32-input MFMA), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0         # spec; ~6290 GB/s measured copy ceiling

# Hyper-parameters are the reference's logged run values (slow_lr 0.002, fast_lr 0.01, gamma 0.9),
# except fast_lr at N = 256: the reference's plain full-batch SGD local fit DIVERGES to NaN there (768 unscaled
# inputs; the oracle reproduces it: oracle fit at N=256, lr=0.01 -> NaN within 10 epochs, lr <= 0.005 converges).
# A benchmark on NaN weights would be meaningless (and data-dependent clocks would flatter it), so the N=256
# workloads use fast_lr = 0.001 (and cfg3, N = 64 with a bootstrapped critic, 0.005), and main() asserts that every weight
# on every rank is finite at the end.  (Rounds 1-3 used 0.0025 = 0.01 * 64/N: there one to three of a shard's 4096 team-reward
# fits still blow up -- to a finite 1e8 on the seeds of rank 0, to NaN on the seed shards of ranks 1 and 2, in exact arithmetic
# too (profiles/r03ah_*) -- which only shows once all ranks are checked; at 0.001 no fit leaves |w| < 4 on any of the 8 shards,
# profiles/r03ai_*.  The step count, and therefore the timing, does not depend on the learning rate.)
WORKLOADS = {
    # BASELINE.json configs[3] sharded over the node: 128 seeds / 8 GPUs = 16 seeds per GPU (weak scaling)
    "cfg4_shard": dict(N=256, nrow=32, ncol=32, H=8, d=18, S=16, graph="circulant", fast_lr=0.001,
                       desc="BASELINE configs[3] per-GPU shard: 256 agents, 32x32 grid, H=8, circulant in-degree d=18 "
                            "(=2H+2), 16 independent seeds per GPU, all cooperative"),
    # the configuration the north-star targets are quoted on
    "target_N256_H1": dict(N=256, nrow=5, ncol=5, H=1, d=4, S=16, graph="circulant", fast_lr=0.001,
                           desc="north_star target: 256 agents, 5x5 grid, H=1, circulant d=4, 16 seeds per GPU"),
    "cfg3": dict(N=64, nrow=16, ncol=16, H=4, d=10, S=32, graph="regular", fast_lr=0.005,
                 desc="BASELINE configs[2]: 64 agents, 16x16 grid, random 9-regular in-graph + self (d=10), H=4, 32 seeds per GPU"),
    # BASELINE configs[1] (the reference's own adversarial scenario, .../malicious/H=1), many seeds per GPU
    "cfg2_batched": dict(N=5, nrow=5, ncol=5, H=1, d=4, S=512, graph="circulant",
                         labels=["Cooperative"] * 4 + ["Malicious"],
                         desc="BASELINE configs[1]: 4 cooperative + 1 Malicious agent, 5x5 grid, H=1, circulant d=4, "
                              "512 independent seeds per GPU"),
    "cfg1_batched": dict(N=5, nrow=5, ncol=5, H=1, d=4, S=512, graph="circulant",
                         desc="5 cooperative agents, 5x5 grid, H=1, 512 seeds per GPU"),
    # BASELINE configs[0]: the reference's own CPU-runnable case (main.py defaults with --H 0): plain-mean consensus
    "cfg0_H0_batched": dict(N=5, nrow=5, ncol=5, H=0, d=4, S=512, graph="circulant",
                            desc="BASELINE configs[0]: 5 cooperative agents, 5x5 grid, H=0 (plain mean over the in-neighbourhood, "
                                 "agents/resilient_CAC_agents.py:50-56 with H=0), circulant d=4, 512 seeds per GPU"),
    # single-instance (S = 1) forms: what `python -m rcmarl_amd.main` / train_RPBCAC run (reference main.py:117 trains ONE seed);
    # SURVEY.md section 8's size table has S per GPU = 1 for configs 1-3.  Latency-bound: ~600 launches per block.
    "cfg0_H0_single": dict(N=5, nrow=5, ncol=5, H=0, d=4, S=1, graph="circulant",
                           desc="BASELINE configs[0] as ONE instance (the drop-in train_RPBCAC path): 5 cooperative agents, H=0"),
    "cfg2_single": dict(N=5, nrow=5, ncol=5, H=1, d=4, S=1, graph="circulant", labels=["Cooperative"] * 4 + ["Malicious"],
                        desc="BASELINE configs[1] as ONE instance (the drop-in train_RPBCAC path): 4 cooperative + 1 Malicious "
                             "agent, 5x5 grid, H=1"),
    "cfg3_single": dict(N=64, nrow=16, ncol=16, H=4, d=10, S=1, graph="regular", fast_lr=0.005,
                        desc="BASELINE configs[2] as ONE instance: 64 agents, 16x16 grid, random 9-regular in-graph + self, H=4"),
    # BASELINE configs[4] as ONE instance on ONE GPU (the 8-GPU agent/column sharding of SURVEY.md 8e is not built):
    # only the critic is widened to 512 units (BASELINE: "wide (512-unit) critic"), team-reward net and actor keep 20
    "cfg5_1gpu": dict(N=1024, nrow=32, ncol=32, H=32, d=66, S=1, graph="circulant", fast_lr=0.0005, critic_hid=512,
                      desc="BASELINE configs[4] on one GPU: 1024 agents, 512-unit critic (dense per-agent GEMMs on the 16-bit matrix core from "
                           "PRE-SPLIT packed operands written by the producing kernels' epilogues, csrc/dense_pk.hip; layer 1 on the lattice "
                           "GEMMs), 20-unit team-reward net and actor, 32x32 grid, H=32, circulant in-degree d=66 (=2H+2), one instance"),
    # the same instance sharded over ALL ranks of the job (strong scaling): agents for the per-agent phases, parameter
    # columns for the hidden-layer consensus, two all-to-all transposes per epoch (RPBCACEngine.shard_agents)
    "cfg5_shard": dict(N=1024, nrow=32, ncol=32, H=32, d=66, S=1, graph="circulant", fast_lr=0.0005, critic_hid=512,
                       shard_instance=True,
                       desc="BASELINE configs[4] as ONE instance over all ranks: 1024 agents, 512-unit critic sharded by agent "
                            "(fits, estimate consensus) and by parameter column (hidden-layer consensus), the team-reward net "
                            "alike; actors, environment and replay replicated"),
}


def build_graph(kind, N, d, seed=0):
    if kind == "circulant":
        return [[(i + k) % N for k in range(d)] for i in range(N)]
    rng = np.random.default_rng(seed)
    out = []
    for i in range(N):
        others = [j for j in range(N) if j != i]
        out.append([i] + [int(x) for x in rng.permutation(others)[:d - 1]])
    return out


def make_engine(w, S, seeds, lib):
    from rcmarl_amd.engine import EngineConfig, RPBCACEngine
    N = w["N"]
    cfg = EngineConfig(N, w.get("labels", ["Cooperative"] * N), build_graph(w["graph"], N, w["d"]), H=w["H"], gamma=0.9, slow_lr=0.002,
                       fast_lr=w.get("fast_lr", 0.01), max_ep_len=20, n_ep_fixed=50, n_epochs=10, buffer_size=2000, nrow=w["nrow"],
                       ncol=w["ncol"], n_seeds=S, rng_mode="device", critic_hid=w.get("critic_hid", 20))
    eng = RPBCACEngine(cfg, seeds=seeds, device="cuda", lib=lib)
    eng.init_glorot(base_seed=1)
    goals = np.stack([np.random.RandomState(int(s)).randint(0, 5, size=(N, 2)) for s in seeds])   # main.py:48 draws goals in [0,5)
    eng.set_goals(goals)
    # Steady state of the reference loop: the replay lists hold buffer_size rows when a block starts and
    # grow to buffer_size + n_ep_fixed*max_ep_len = 3000 before the update (train_agents.py:158-163).
    # Pre-fill with rollouts only, so that EVERY update block of this process (warm-up included) runs at the
    # steady-state batch B = 3000 and rocprof's per-kernel averages are over identical launches.
    while eng.B + eng.n_last <= cfg.buffer_size:
        eng.rollout_block(cfg.n_ep_fixed)
    return eng


# --------------------------------------------------------------------------------------------
def _blas_threads():
    """Threads the NumPy port actually uses: its Python loops are serial, the matmuls inside run on
    OpenBLAS' pool."""
    try:
        import threadpoolctl
        n = [d.get("num_threads", 1) for d in threadpoolctl.threadpool_info() if d.get("user_api") == "blas"]
        return int(max(n)) if n else 1
    except Exception:
        return 1


def cpu_baseline(w, budget_s=25.0):
    """The reference's loop structure (per agent -> per neighbour -> per layer; oracle/
    rpbcac_oracle.py) timed on this host on a bounded sample of the same workload and
    extrapolated linearly to one training block of one seed."""
    from oracle import rpbcac_oracle as O
    from oracle import mlp_np as M
    N, d, H = w["N"], w["d"], w["H"]
    B, n_last, n_epochs, steps_per_block = 3000, 1000, 10, 1000
    rng = np.random.default_rng(0)
    in_nodes = build_graph(w["graph"], N, d)
    chid = w.get("critic_hid", 20)
    k = (8 if chid == 20 else 2) if N >= 64 else min(N, 5)           # agents whose update phases are timed
    # only the k sampled agents need a critic of their own (a 512-unit critic at N = 1024 is 4 MB per agent)
    critics = [M.init_mlp(rng, 2 * N, chid, 1) for _ in range(k)]
    agents = [O.CoopAgent(M.init_mlp(rng, 2 * N, 20, 5), critics[i] if i < k else critics[0], M.init_mlp(rng, 3 * N, 20, 1),
                          0.002, 0.01, 0.9, H) for i in range(N)]
    goals = rng.integers(0, 5, size=(N, 2))
    env = O.GridWorldOracle(w["nrow"], w["ncol"], N, goals, None, True, True, rng_mode="device", seed=1)
    # (a) rollout: per-agent batch-of-one policy forward + numpy RNG draws + python env step
    env.reset(episode=0)
    state, _ = env.get_data()
    n_steps = 20 if N <= 256 else 6
    reps = 3

    def timed(fn):
        """min and median of `reps` repeats (the host is shared with nobody, but BLAS thread start-up and page faults land on
        the first repeat)"""
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), sorted(ts)[len(ts) // 2]

    def roll():
        nonlocal state
        for j in range(n_steps):
            action = np.zeros(N)
            for i in range(N):
                action[i] = agents[i].act_numpy(state[None])
            env.step(action)
            state, _ = env.get_data()
    t_step, t_step_med = (t / n_steps for t in timed(roll))
    # (b) phases I-III on a sample of agents
    s = rng.normal(size=(B, N, 2)).astype(np.float32)
    ns = rng.normal(size=(B, N, 2)).astype(np.float32)
    a = rng.integers(0, 5, size=(B, N, 1)).astype(np.float32)
    r = -rng.integers(0, 9, size=(B, N, 1)).astype(np.float32) / 5
    sa = np.concatenate([s, a], axis=-1)
    sample = list(range(k))
    msgs_c, msgs_t = [], []

    def fits():
        msgs_c.clear()
        msgs_t.clear()
        for i in sample:
            x, _ = agents[i].local_fit_tr(sa, r[:, i])
            y, _ = agents[i].local_fit_critic(s, ns, r[:, i])
            msgs_t.append(x)
            msgs_c.append(y)

    def cons():
        for i in sample:
            c_in = [msgs_c[j % k] for j in range(d)]
            t_in = [msgs_t[j % k] for j in range(d)]
            ag = agents[i]
            ag.consensus_hidden_critic(c_in)
            ag.consensus_hidden_tr(t_in)
            c_agg = ag.consensus_estimates_critic(s, c_in)
            t_agg = ag.consensus_estimates_tr(sa, t_in)
            ag.projection_step_critic(s, c_agg)
            ag.projection_step_tr(sa, t_agg)

    def actor():
        for i in sample:
            agents[i].actor_step(s[-n_last:], ns[-n_last:], sa[-n_last:], a[-n_last:, i])
    (t_fit, t_fit_med), (t_cons, t_cons_med), (t_actor, t_actor_med) = (tuple(t / k for t in timed(f)) for f in (fits, cons, actor))

    def block_of(ts, tf, tc, ta):
        return steps_per_block * ts + n_epochs * N * (tf + tc) + N * ta
    block = block_of(t_step_med, t_fit_med, t_cons_med, t_actor_med)              # the reported value: medians
    block_fast = block_of(t_step, t_fit, t_cons, t_actor)                         # ... and the spread: minima
    return {
        "value": N * steps_per_block / block, "unit": "agent-steps/s", "cores": _blas_threads(),
        "host_cpus": os.cpu_count(), "kind": "port",
        "value_from_minima": N * steps_per_block / block_fast, "repeats": reps,
        "consensus_updates_per_s": 1.0 / t_cons_med,
        "sample": "oracle/rpbcac_oracle.py (reference loop structure, numpy fp32), one seed: %d env steps with all %d agents; "
                  "local fits, consensus b/c/d and actor step of %d sample agents at B=%d; each part %d times, median (value) and "
                  "minimum (value_from_minima); extrapolated linearly to one block (1000 env steps + 10 epochs x %d agents + "
                  "actor step)" % (n_steps, N, k, B, reps, N),
        "seconds": {"env_step_all_agents": t_step_med, "local_fit_per_agent_epoch": t_fit_med,
                    "consensus_per_agent_epoch": t_cons_med, "actor_per_agent": t_actor_med, "block_extrapolated": block},
    }


# --------------------------------------------------------------------------------------------
class StubEngine:
    """Test double for the CONTROL path of this script (tests/test_bench_launcher.py): same surface as
    RPBCACEngine as far as main() touches it, no GPU, no HIP library.  Selected with --stub-engine only; a
    bench line produced with it says so in `data` and is not a measurement."""

    class _Cfg:
        n_ep_fixed, max_ep_len, n_epochs, fast_lr, slow_lr, critic_hid = 50, 20, 10, 0.01, 0.002, 20

    def __init__(self, w, S, seeds):
        self.cfg, self.S, self.N, self.seeds = self._Cfg(), S, w["N"], list(seeds)
        self.n_coop, self.cap, self.blocks = w["N"], 3000, 0
        self.timers = {"rollout": 0.0, "phase1": 0.0, "phase2": 0.0, "phase3": 0.0, "blocks": 0}
        self.theta = {k: torch.zeros(1) for k in ("actor", "critic", "tr")}
        self.profile_phases = False

    def run_block(self):
        time.sleep(0.01)
        self.blocks += 1
        if self.profile_phases:
            for k in ("rollout", "phase1", "phase2", "phase3"):
                self.timers[k] += 0.0025
        E = self.cfg.n_ep_fixed
        team = np.tile(-np.asarray(self.seeds, np.float64)[None, :], (E, 1))      # return of seed s == -s
        return team, np.zeros_like(team), team + 1.0


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves
    (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1) and become that launcher.
    The reference has no counterpart -- it runs one SGE job per seed (.../seed=100/job.sh:4)."""
    if not args.stub_engine:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench: --gpus %d requested but only %d GPU(s) visible; refusing to report a %d-GPU number "
                             "from fewer devices" % (args.gpus, have, args.gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg4_shard", choices=sorted(WORKLOADS))
    ap.add_argument("--seeds-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing-steps", type=int, default=2,
                    help="timed steps whose launches carry HIP events (evenly spaced; 0 = all of them).  Two events per launch cost "
                         "~10 us of queue time each: on all ~5000 launches of a 10-step run that is 5 %% of the step "
                         "(204.4 against 194.4 ms per block at cfg4_shard, profiles/r04x_event_overhead.txt)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other workloads (`extra`)")
    ap.add_argument("--stub-engine", action="store_true", help="TEST ONLY: CPU stub engine + gloo, exercises the launcher/"
                    "barrier/all-reduce control path without a GPU")
    return ap.parse_args(argv)


def timed_steps(steps, k):
    """the `k` evenly spaced steps of `steps` whose launches carry events (all of them for k <= 0 or k >= steps)"""
    if k <= 0 or k >= steps:
        return list(range(steps))
    return sorted({(2 * i + 1) * steps // (2 * k) for i in range(k)})


def time_blocks(eng, steps, warmup, barrier, S, dev, tlib=None, want_kernels=True, event_steps=0):
    """W untimed blocks, then exactly K timed blocks bracketed by barrier + synchronize; returns
    (seconds on this rank, mean return curve over ALL ranks' seeds).  Per-kernel HIP events ride on the launches of
    timed_steps(steps, event_steps) -- inside the timed region, on the launch stream."""
    from rcmarl_amd.parallel import allreduce_curves
    for _ in range(warmup):
        eng.run_block()
    barrier()
    ev = set(timed_steps(steps, event_steps)) if want_kernels else set()
    if tlib is not None:
        tlib.enabled = False
        tlib.reset()
    curves = []
    t0 = time.perf_counter()
    for k in range(steps):
        if tlib is not None:
            tlib.enabled = k in ev
        team, adv, est = eng.run_block()
        curves.append(np.stack([team.sum(1), adv.sum(1), est.sum(1)], axis=1))   # per-episode sums over local seeds
    curve = allreduce_curves(np.concatenate(curves, 0), S, device=dev)            # C1: the path's only collective
    barrier()
    dt = time.perf_counter() - t0
    if tlib is not None:
        tlib.enabled = False
    return dt, curve


def _self_hash():
    import hashlib
    with open(os.path.abspath(__file__), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def _speedup(out):
    """GPU line / CPU port, to the two digits the bounded CPU sample supports, with the spread of the CPU measurement"""
    cb = out["cpu_baseline"]
    return {"x": float("%.2g" % (out["value"] / cb["value"])),
            "x_against_the_fastest_cpu_repeat": float("%.2g" % (out["value"] / cb.get("value_from_minima", cb["value"])))}


def _lat_mode(tlib):
    try:
        return int(tlib.rcmarl_lattice_f16_mode()) if tlib is not None else 0
    except Exception:
        return 0


def phase_split(eng):
    """one extra, untimed block with a synchronisation at every phase boundary"""
    eng.profile_phases = True
    for k in eng.timers:
        eng.timers[k] = 0.0
    eng.run_block()
    eng.profile_phases = False
    return dict(eng.timers)


def extra_workloads(main_name, tlib, barrier, dev, main_steps=2, main_warmup=1, cpu_target=True):
    """Short driver-timed runs of the OTHER workloads in the same process (1 warm-up + 2 timed blocks each), so the one
    bench line also carries BASELINE configs[1], [2], [4] and the north-star target shape; K1's roofline on the target
    shape is measured here with HIP events (`roofline_consensus_target`).  Two of them get more: the headline workload in its EXACT
    operand form is timed with the headline's own --steps / --warmup (it is the strict-fp32 figure of the same line), and the
    north-star target shape gets its own CPU baseline (the >= 100x target is quoted on that shape)."""
    out, k1_target = {}, None
    exact = main_name + "_exact"           # the headline workload again in the EXACT operand form (three bf16 pieces, fp32 mid kernels)
    for name in (exact, "target_N256_H1", "cfg3", "cfg2_batched", "cfg1_batched", "cfg0_H0_batched", "cfg0_H0_single", "cfg2_single",
                 "cfg3_single", "cfg5_1gpu"):
        if name == main_name:
            continue
        w = WORKLOADS[main_name if name == exact else name]
        saved_env = None
        try:
            if name == exact:
                # every fp32 operand of the matrix-core products as three bf16 pieces whose sum IS the fp32 value, layers 2-3 and
                # the adversaries' chains on the fp32-arithmetic kernels: no operand narrower than the reference's fp32
                saved_env = {k: os.environ.get(k) for k in ("RCMARL_LAT_F16", "RCMARL_MIDFIT", "RCMARL_MB_MX")}
                os.environ.update(RCMARL_LAT_F16="0", RCMARL_MIDFIT="5", RCMARL_MB_MX="0")
                tlib.rcmarl_lattice_set_f16_mode(0)
                tlib.rcmarl_wide_set_f16_mode(0)                 # a wide critic's dense layers on the f32-input matrix-core kernel
            S = w["S"]
            t_setup = time.perf_counter()
            eng = make_engine(w, S, [1000 + k for k in range(S)], tlib)
            t_setup = time.perf_counter() - t_setup
            steps, warm = (max(int(main_steps), 1), max(int(main_warmup), 0)) if name == exact else (2, 1)
            # single instances are launch-bound: timed WITHOUT the per-launch events, so the engine's captured epochs
            # (hipGraph, engine._epoch) run as they do for a user of the drop-in path
            single = name.endswith("_single")
            dt, _ = time_blocks(eng, steps, warm, barrier, S, dev, tlib, want_kernels=not single, event_steps=1)
            ksum = {} if single else tlib.summary()
            finite = all(bool(torch.isfinite(eng.theta[k]).all().item()) for k in ("actor", "critic", "tr"))
            c = eng.cfg
            env_steps = c.n_ep_fixed * c.max_ep_len
            rec = {"description": w["desc"], "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warm,
                   "agent_steps_per_s": S * w["N"] * env_steps * steps / dt,
                   "consensus_updates_per_s": S * eng.n_coop * c.n_epochs * steps / dt,
                   "weights_finite": finite, "fast_lr": c.fast_lr, "setup_s": round(t_setup, 2)}
            if single:
                rec["epochs_replayed_from_hipgraph"] = eng.graph_replays
            if ksum:
                tot_ms = sum(v[1] for v in ksum.values())
                top = sorted(ksum.items(), key=lambda kv: -kv[1][1])[:4]
                rec["top_kernels"] = {k.replace("rcmarl_", ""): {"avg_us": round(v[2], 2), "frac": round(v[1] / tot_ms, 4)}
                                      for k, v in top}
                _, k1, _, _ = rooflines(tlib, ksum, name)
                if k1:
                    rec["roofline_consensus"] = {k: k1[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_us",
                                                                    "launches", "algorithmic_bytes_per_launch",
                                                                    "traffic", "traffic_source")}
                    if name == "target_N256_H1":
                        k1_target = k1
            if name == "target_N256_H1" and cpu_target:
                # the north-star's ">= 100x agent-steps/s over the reference CPU path at N=256, H=1, 5x5 grid" is quoted on THIS shape
                try:
                    del eng
                    torch.cuda.empty_cache()
                    eng = None
                    cb = cpu_baseline(w, budget_s=20.0)
                    rec["cpu_baseline"] = _brief(cb)
                    rec["speedup_vs_cpu_port"] = {"x": float("%.2g" % (rec["agent_steps_per_s"] / cb["value"])),
                                                  "x_against_the_fastest_cpu_repeat":
                                                      float("%.2g" % (rec["agent_steps_per_s"] / cb.get("value_from_minima", cb["value"]))),
                                                  "target": ">= 100 (north_star)"}
                except Exception as e:
                    rec["cpu_baseline"] = {"error": repr(e)}
            if name == exact:
                rec["operand_form"] = ("exact: RCMARL_LAT_F16=0 (three bf16 pieces whose sum is the fp32 operand, bit for bit), "
                                       "RCMARL_MIDFIT=5 and RCMARL_MB_MX=0 (fp32-arithmetic mid / mini-batch kernels), "
                                       "rcmarl_wide_set_f16_mode(0) (a wide critic's dense layers on the f32-input MFMA kernel)")
            out[name] = rec
            eng = None
            torch.cuda.empty_cache()
        except Exception as e:                                  # an extra must never kill the bench line
            out[name] = {"error": repr(e)}
        finally:
            if saved_env is not None:
                for k, v in saved_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                tlib.rcmarl_lattice_set_f16_mode(-1)
                tlib.rcmarl_wide_set_f16_mode(-1)
    return out, k1_target


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)                                       # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    import torch.distributed as dist
    stub = args.stub_engine
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    if not stub:
        torch.cuda.set_device(local_rank)
    comm = None
    # RCMARL_BENCH_FORCE_PG=1: create the process group even for ONE rank (exercises the RCCL plumbing on a 1-GPU box)
    use_pg = world > 1 or (os.environ.get("RCMARL_BENCH_FORCE_PG") == "1" and "RANK" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)           # "nccl" IS RCCL on ROCm
        joined = dist.get_world_size()
        if joined != args.gpus:
            raise SystemExit("bench: %d ranks joined, --gpus %d" % (joined, args.gpus))
        comm = {"backend": dist.get_backend(), "world_size": joined,
                "rccl_version": None if stub else ".".join(str(v) for v in torch.cuda.nccl.version())}

    w = WORKLOADS[args.workload]
    S = args.seeds_per_gpu or w["S"]
    one_instance = bool(w.get("shard_instance"))                   # every rank works on the SAME instance (strong scaling)
    seeds = [1000 + (0 if one_instance else rank * S) + k for k in range(S)]      # else: disjoint seed shards per rank
    N = w["N"]
    if stub:
        tlib, eng = None, StubEngine(w, S, seeds)
    else:
        from rcmarl_amd import capi
        from rcmarl_amd.timing import TimedLib
        tlib = TimedLib(capi.load())
        eng = make_engine(w, S, seeds, tlib)
        if one_instance and use_pg:
            eng.shard_agents(force=True)                           # over the default process group (also ONE rank: the
            #                                                        RCCL path of the sharded instance on a 1-GPU box)
    c = eng.cfg
    jobs = 1 if one_instance else world                            # independent instances of the workload in the job

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    dt, curve = time_blocks(eng, args.steps, args.warmup, barrier, S, dev, tlib, want_kernels=not args.no_kernel_timing,
                            event_steps=args.kernel_timing_steps)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                # the slowest rank's clock
    dt = float(tmax.item())
    B_steady = eng.cap

    ksum = tlib.summary() if tlib is not None else {}              # per-kernel HIP-event times of the timed region
    roof = rooflines(tlib, ksum, args.workload) if ksum else None  # (before the extras reset the counters)
    ph = phase_split(eng)

    finite = all(bool(torch.isfinite(eng.theta[k]).all().item()) for k in ("actor", "critic", "tr"))
    if use_pg:
        # every rank must reach the same verdict BEFORE anyone leaves: a rank that exits alone leaves the others in the
        # barrier below (and in a sharded instance a remote shard's divergence is invisible in the local rows)
        fl = torch.tensor([1.0 if finite else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(fl, op=dist.ReduceOp.MIN)
        finite = bool(fl.item() > 0.5)
    if not finite:
        if use_pg:
            dist.destroy_process_group()
        raise SystemExit("bench invalid: non-finite network weights after the timed region (diverged training)")
    if rank == 0:
        env_steps = c.n_ep_fixed * c.max_ep_len
        agent_steps = jobs * S * N * env_steps * args.steps
        cons_updates = jobs * S * eng.n_coop * c.n_epochs * args.steps
        ph_total = ph["rollout"] + ph["phase1"] + ph["phase2"] + ph["phase3"]
        out = {
            "metric": "agent-steps/sec (whole RPBCAC training loop; + consensus-updates/sec)",
            "value": agent_steps / dt, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if one_instance else "weak",
            "vs_baseline": None,
            "dtype": "f32 (matrix-core operands as 2 x f16 pieces: each fp32 operand to <= 1 ulp; the 20-unit layers run 3 of the 4 piece products)" if _lat_mode(tlib) else "f32",
            "dtype_note": "fp32 parameters, activations and accumulation throughout; the matrix-core products of Phase I take their fp32 "
                          "operands as two f16 pieces (the value to one unit in its last place: 22-23 significand bits where fp32 has 24; "
                          "RCMARL_LAT_F16=3, the default; a value below 2^-3 of its fixed scale -- |alpha W1| < 1.2e-4, |dz1| < 4.9e-4 -- is carried to an "
                          "absolute 3e-11 / 1.2e-10 instead) or, =0, as three bf16 pieces whose sum is the fp32 value bit for bit -- that form "
                          "is timed in extra.<workload>_exact and reported as ms_per_step_exact / value_exact of this same line",
            "data": "synthetic" if not stub else "STUB ENGINE (control-path test, not a measurement)",
            "config": {"workload": args.workload, "description": w["desc"], "n_agents": N, "seeds_per_gpu": S,
                       "grid": [w["nrow"], w["ncol"]], "H": w["H"], "d": w["d"], "replay_rows_B": B_steady, "fast_lr": c.fast_lr, "slow_lr": c.slow_lr, "weights_finite": finite,
                       "env_steps_per_block": env_steps, "n_epochs": c.n_epochs, "hidden": 20, "critic_hidden": c.critic_hid,
                       "parallelism": ("one instance over %d GPU: agent-sharded critic phases, column-sharded K1, two all-to-all "
                                       "per epoch" % world) if one_instance else
                                      "seed-sharded, %d seeds/GPU x %d GPU, one all-reduce of return curves" % (S, world)},
            "comm": comm,
            "consensus_updates_per_s": cons_updates / dt,
            "consensus_updates_per_s_phase2_only": (S * eng.n_coop * c.n_epochs) / ph["phase2"] if ph["phase2"] > 0 else None,
            "phase_seconds_per_block": {k: ph[k] for k in ("rollout", "phase1", "phase2", "phase3")},
            "phase_fraction": {k: ph[k] / ph_total for k in ("rollout", "phase1", "phase2", "phase3")} if ph_total > 0 else None,
            "mean_team_return_last_block": float(curve[-c.n_ep_fixed:, 0].mean()),
        }
        if abs(c.fast_lr - 0.01) > 1e-12:
            out["deviation"] = {"fast_lr": c.fast_lr, "reference": 0.01,
                                "why": "the reference's plain full-batch SGD local fit diverges to NaN at this input width "
                                       "with its logged fast_lr (the oracle reproduces it); work per step is unchanged"}
        if ksum:
            tot_ms = sum(v[1] for v in ksum.values())
            out["kernels"] = {k.replace("rcmarl_", ""): {"launches": v[0], "total_ms": round(v[1], 3), "avg_us": round(v[2], 2),
                                                        "frac": round(v[1] / tot_ms, 4)} for k, v in
                              sorted(ksum.items(), key=lambda kv: -kv[1][1])}
            out["roofline"], out["roofline_consensus"], out["roofline_gemm"], out["roofline_mid"] = roof
            out["kernel_timing"] = {"steps_with_events": timed_steps(args.steps, args.kernel_timing_steps), "of_timed_steps": args.steps,
                                    "what": "HIP events around every C-ABI launch of these timed steps, on the launch stream; "
                                            "launches / total_ms in `kernels` count those steps only"}
        if world == 1 and not stub and not args.no_extra:
            del eng
            torch.cuda.empty_cache()
            out["extra"], k1t = extra_workloads(args.workload, tlib, barrier, dev, args.steps, args.warmup, cpu_target=not args.no_cpu_baseline)
            if k1t is not None:
                out["roofline_consensus_target"] = k1t
        cache = os.path.join(ROOT, ".bench_cpu_baseline_%s.json" % args.workload.replace("cfg5_shard", "cfg5_1gpu"))
        if not args.no_cpu_baseline and world == 1 and not stub:
            try:
                out["cpu_baseline"] = cpu_baseline(w)
                out["speedup_vs_cpu_port"] = _speedup(out)
                with open(cache, "w") as f:                            # the N > 1 lines of the same box sit beside it
                    json.dump(dict(out["cpu_baseline"], bench_py_sha=_self_hash(), workload=args.workload), f)
            except Exception as e:                                     # the baseline must never kill the bench line
                out["cpu_baseline"] = {"error": repr(e)}
        elif not args.no_cpu_baseline and not stub and os.path.exists(cache):
            # timed on rank 0 at N = 1 only (a few tens of CPU-seconds); the N > 1 lines of the same box carry that record
            try:
                with open(cache) as f:
                    rec = json.load(f)
                if rec.get("bench_py_sha") != _self_hash():            # written by another bench.py: not this measurement's baseline
                    raise ValueError("stale CPU-baseline cache (another bench.py wrote it)")
                out["cpu_baseline"] = dict(rec, carried_from="the N=1 run of this workload on this box")
                out["speedup_vs_cpu_port"] = _speedup(out)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        ex = (out.get("extra") or {}).get(args.workload + "_exact") or {}
        if "ms_per_step" in ex:
            # the headline workload with NO operand narrower than the reference's fp32 (three bf16 pieces whose sum is the fp32
            # value, fp32-arithmetic mid kernels): the strict-fp32 form of `value` / `ms_per_step`, same process, same box
            out["ms_per_step_exact"] = ex["ms_per_step"]
            out["value_exact"] = ex["agent_steps_per_s"]
            out["exact_steps"] = ex.get("steps")
            out["config"] = dict(out["config"], ms_per_step_exact=ex["ms_per_step"], value_exact=ex["agent_steps_per_s"],
                                 exact_form="three bf16 pieces whose sum is the fp32 operand + fp32-arithmetic mid kernels "
                                            "(extra.%s_exact)" % args.workload)
        # big objects first, the scalars a reader wants LAST (a truncated tail of this line still holds them)
        big = ("dtype_note", "config", "kernels", "extra", "roofline_gemm", "roofline_mid", "roofline_consensus",
               "roofline_consensus_target", "kernel_timing", "deviation", "phase_seconds_per_block", "phase_fraction", "comm")
        line = {k: out[k] for k in big if k in out}
        line.update({k: v for k, v in out.items() if k not in big and k not in ("roofline", "cpu_baseline")})
        line["summary_ms_per_step"] = dict({args.workload: round(out["ms_per_step"], 2)},
                                           **{k: (round(v["ms_per_step"], 2) if "ms_per_step" in v else "error")
                                              for k, v in (out.get("extra") or {}).items()})
        line["summary_hipgraph_epochs"] = {k: v.get("epochs_replayed_from_hipgraph") for k, v in (out.get("extra") or {}).items()
                                           if "epochs_replayed_from_hipgraph" in v}
        for k in ("roofline", "cpu_baseline"):
            if k in out:
                line[k] = _brief(out[k])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_exact", "value_exact"):
            if k in line:
                line[k] = line.pop(k)
        print(json.dumps(line))
        sys.stdout.flush()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def _brief(o):
    """the object with its prose shortened (the full notes are in DESIGN.md section 3 / 5)"""
    if not isinstance(o, dict):
        return o
    return {k: (_brief(v) if isinstance(v, dict) else
                (v[:160] + " ..." if k in ("note", "what", "why") and isinstance(v, str) and len(v) > 164 else v)) for k, v in o.items()}


BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 MFMA, MI355X_MICROARCH.md
BF16_SUSTAINED_TFLOPS = 1780.0  # measured on this pool: pure v_mfma_f32_32x32x16_bf16 loops on all SIMDs with operands that
#                                 change every instruction (tools/micro/mfma_peak.hip; 2100 with constant operands).  It is a POWER
#                                 limit: that stream runs at an effective 1.79 GHz with the matrix pipe 97 % busy (2.09 GHz with
#                                 constant operands; GRBM_GUI_ACTIVE / duration, profiles/r05_mfma_peak_clock.txt); the lattice GEMMs
#                                 themselves clock at 1.45-1.6 GHz (roofline*.counters)

# kernel -> (bound, note).  "mfma_pieces": fp32-equivalent flops 2MNK against the 16-bit dense peak (f16 = bf16 rate); the kernel
# EXECUTES as many times those flops as the fp32 operand has 16-bit pieces (2 f16, or 3 bf16), so its ceiling is peak / pieces.
ROOFLINE_KIND = {
    "rcmarl_consensus_params": ("hbm", "algorithmic bytes 8*P_hid per (seed, cooperative agent) (SURVEY 8d). HBM-bound for "
                                "small d (d=4: ~54% of 8 TB/s); at d=18 the 128-op min/max selection network makes it "
                                "VALU-issue-bound (~83% of the v_min/v_max issue rate, DESIGN.md section 3)"),
    "rcmarl_consensus_params_circulant": ("hbm", "algorithmic bytes 8*P_hid per (seed, cooperative agent) (SURVEY 8d); circulant "
                                          "graph: one selection network per G consecutive agents (96/4 + 16 min/max ops per "
                                          "agent at (18,8) instead of 128) + 18 clamps + 9 packed adds + a 3-instruction "
                                          "division: VALU-bound at d=18 (343 instructions per 4 agents, VALU busy 77 % of the "
                                          "kernel: profiles/r02h_sq_counters_k1_circ_d18.json), toward HBM at d=4"),
    "rcmarl_mid_fit_lattice": ("hbm", "algorithmic bytes per (seed, agent, replay row): 20 fp32 activations read + 20 x 2 f16 "
                               "dz1 pieces written = 160 B (RCMARL_LAT_F16=0: 20 x 3 bf16 pieces, 200 B); one API call = k_mid_fit_v8 (layers 2-3 "
                               "and the row reduction as v_mfma_f32_32x32x16_f16 on two-piece f16 operands, four exact products per fp32 "
                               "product) + a fix-up launch of the fp32 kernel k_mid_fit_v5 for agents whose operands left the f16 range "
                               "(returns at once otherwise) + a one-thread generation bump.  Round 4: ~220 vector + 24 MFMA + ~45 LDS instructions "
                               "per 32 rows and wavefront (was ~480 + 24 + ~75), the next block's loads issued a block ahead, one record "
                               "reduction per workgroup: 600 -> 455-470 us; at 4.2 TB/s of its own bytes it is within ~20 % of what a "
                               "half-read half-write stream sustains on this part (DESIGN.md section 5, Round 4; SQ counters of the "
                               "intermediate build: profiles/r04r_sq_k_mid_fit_v8_prefetch_build.json).  The f32-input MFMA form it "
                               "replaced (v5, RCMARL_MIDFIT=5) runs 720-770 us"),
    "rcmarl_minibatch_fit": ("mfma_f32", "the adversaries' fit(batch_size=32, epochs=10): 940 sequentially DEPENDENT SGD steps per "
                             "network, one wavefront per network (6 us per step): bound by the latency of one step, not by a pipe; "
                             "flops = 6 per weight per row"),
    "rcmarl_mid_fit": ("hbm", "algorithmic bytes per (seed, agent, replay row): 20 fp32 read + 20 fp32 written"),
    "rcmarl_w1_split": ("hbm", "reads W1 (4 B/weight), writes two f16 pieces (4 B/weight; three bf16 pieces with RCMARL_LAT_F16=0)"),
    "rcmarl_minibatch_fit_multi": ("mfma_f32", "all mini-batch fits of a consensus epoch (the adversaries' fit(batch_size=32, epochs=10), e.g. a "
                                   "Malicious agent's three 940-step chains) in ONE launch: sequentially DEPENDENT SGD steps, one wavefront "
                                   "per network -- bound by the latency of one step, not by a pipe; flops = 6 per weight per row"),
    "rcmarl_layer1_forward_lattice_pk": ("mfma_pieces", ""),
    "rcmarl_pk_forward2": ("mfma_passes:3", "layer 2 forward of a wide critic from packed operands (weights and activations as two f16 pieces "
                           "each, the l*l product dropped)"),
    "rcmarl_pk_backward_data": ("mfma_passes:2", "dz1 of a wide critic: the LeakyReLU mask of layer 2 is ONE exact f16 piece, W2 W3 two"),
    "rcmarl_pk_backward_w2": ("mfma_passes:3", "W2 gradient + SGD of a wide critic: activations as two f16 pieces, the second operand selected "
                              "in the k-loop between the pieces of dz3 and 0.1 dz3 under the layer-2 mask"),
    "rcmarl_layer1_forward_lattice": ("mfma_pieces", ""),
    "rcmarl_layer1_backward_sgd_lattice": ("mfma_pieces", ""),
}


def _pmc_traffic(workload):
    """Per-launch HBM bytes from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_<workload>.json; FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_%s.json" % workload)))
    if not files:
        return {}
    try:
        with open(files[-1]) as f:
            t = dict(json.load(f).get("traffic_bytes_per_launch", {}))
        t["__source__"] = "profiles/" + os.path.basename(files[-1])
        return t
    except Exception:
        return {}


def _pmc_latbench(kernel_substr):
    """Counters of the lattice GEMM kernels from the committed rocprofv3 passes of tools/micro/lat_bench on the same shapes
    (tools/gpu_pmc_latbench.sh -> profiles/*_pmc_latbench.json): effective shader clock, matrix-pipe busy fraction, L2 hit rate."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_latbench.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        for k, v in d.items():
            if k.startswith(kernel_substr + "<"):
                clk = v.get("eff_clock_GHz")
                return {"source": "profiles/" + os.path.basename(files[-1]), "kernel": k, "eff_clock_GHz": clk,
                        "mfma_busy_frac_of_cycles": v.get("mfma_busy_frac"), "l2_hit_rate": v.get("l2_hit_rate"),
                        "measured_where": "NOT this process and NOT this box: replayed from the committed rocprofv3 passes of "
                                          "tools/micro/lat_bench on the same shapes (an earlier GPU visit).  Clock and pipe duty vary "
                                          "by a few per cent between boxes: do not combine them with this run's `achieved` / `executed`",
                        "what": "GRBM_GUI_ACTIVE / 8 XCDs / duration; SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles; "
                                "TCC_HIT / (TCC_HIT + TCC_MISS)"}
    except Exception:
        return None
    return None


def rooflines(tlib, ksum, workload=None):
    """roofline objects for the time-dominant kernel, the consensus kernel (K1), the layer-1 GEMM and the mid (layers 2-3) step."""
    work = tlib.work
    dom = max(ksum.items(), key=lambda kv: kv[1][1])[0]
    pmc = _pmc_traffic(workload) if workload else {}

    def obj(name):
        n, tot_ms, avg_us = ksum[name]
        flops, byts = work.get(name, (0.0, 0.0))
        kind, note = ROOFLINE_KIND.get(name, ("mfma_f32", ""))
        traffic = pmc.get(name.replace("rcmarl_", ""))
        # `traffic` is NOT measured by this process: it is replayed from the committed rocprofv3 PMC passes of the same
        # command (tools/gpu_pmc_bench.sh); traffic_source names the file (null: no committed pass for this workload)
        src = pmc.get("__source__") if traffic is not None else None
        if kind == "hbm":
            ach = byts / (tot_ms * 1e-3) / 1e9
            out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "launches": n, "avg_us": avg_us,
                   "algorithmic_bytes_per_launch": byts / n, "note": note}
            return out
        ach = flops / (tot_ms * 1e-3) / 1e12
        if kind.startswith("mfma_passes:"):
            npc = int(kind.split(":")[1])
            return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src, "launches": n, "avg_us": avg_us,
                    "algorithmic_flops_per_launch": flops / n,
                    "executed": {"achieved": npc * ach, "frac": npc * ach / BF16_PEAK_TFLOPS,
                                 "what": "16-bit MFMA flops actually issued = %d x the fp32-equivalent 2*B*hid*hid per (seed, agent)" % npc},
                    "note": note}
        if kind == "mfma_pieces":
            from rcmarl_amd.timing import lattice_pieces
            npc = lattice_pieces(1 if "forward" in name else 2)            # matrix passes = 16-bit pieces of the fp32 operand
            form = "two f16 pieces of the scaled fp32 operand (the value to one unit in its last place; RCMARL_LAT_F16)" if npc == 2 \
                else "three exact bf16 pieces of the fp32 operand"
            return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src, "launches": n, "avg_us": avg_us,
                    "algorithmic_flops_per_launch": flops / n,
                    "counters": _pmc_latbench("k_lat_forward" if "forward" in name else "k_lat_backward_sgd"),
                    "executed": {"achieved": npc * ach, "frac": npc * ach / BF16_PEAK_TFLOPS,
                                 "frac_of_measured_sustained_peak": npc * ach / BF16_SUSTAINED_TFLOPS,
                                 "what": "16-bit MFMA flops actually issued = %d x algorithmic; sustained peak = %.0f TFLOP/s "
                                         "measured with tools/micro/mfma_peak.hip at an effective 1.79 GHz (a power limit: "
                                         "profiles/r05_mfma_peak_clock.txt); `counters` has this kernel's own clock and pipe-busy fraction"
                                         % (npc, BF16_SUSTAINED_TFLOPS)},
                    "note": "fp32 GEMM as %d passes of v_mfma_f32_32x32x16_%s (integer-lattice operand x %s, fp32 accumulate): "
                            "achieved = fp32-equivalent 2MNK flops; ceiling of the method = 16-bit dense peak / %d = %.0f TFLOP/s; "
                            "the f32-input MFMA it replaces peaks at 157.3"
                            % (npc, "f16" if npc == 2 else "bf16", form, npc, BF16_PEAK_TFLOPS / npc)}
        return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src, "launches": n, "avg_us": avg_us,
                "algorithmic_flops_per_launch": flops / n,
                "note": note or "fp32-input MFMA (v_mfma_f32_32x32x2_f32), dense fp32 peak 157.3 TFLOP/s"}
    gemm = next((k for k in ("rcmarl_layer1_forward_lattice", "rcmarl_layer1_forward")
                 if k in ksum), None)
    k1 = next((k for k in ("rcmarl_consensus_params_circulant", "rcmarl_consensus_params") if k in ksum), None)
    mid = next((k for k in ("rcmarl_mid_fit_lattice", "rcmarl_mid_fit") if k in ksum), None)
    return (obj(dom), obj(k1) if k1 else None,
            obj(gemm) if gemm else None, obj(mid) if mid else None)


if __name__ == "__main__":
    main()
