"""The agent-sharded wide-critic instance (RPBCACEngine.shard_agents) on REAL kernels: two "ranks" as
two threads of this process, each with its own engine on cuda:0, the two collectives as plain device copies
(parallel.ThreadComm).  Checks what the gloo/hipemu test cannot: the gfx950 kernels (LDS-DMA, packed lattice operands,
MFMA GEMMs) running on agent-range views of every buffer.  Result: bit-identical to the unsharded engine."""
import threading

import numpy as np
import pytest
import torch

import engine_checks as EC
from rcmarl_amd import capi
from rcmarl_amd.parallel import ThreadComm

pytestmark = pytest.mark.gpu

# (agents, d, H, graph, critic width, lattice path)
CASES = [(8, 4, 1, "circ", 64, True),          # packed bf16x3 layer 1: 4 agents x 64 units = two 128-row tiles per rank
         (8, 3, 1, "rand", 128, False),        # dense f32-MFMA path, general K1 kernel
         (16, 6, 2, "circ", 512, True),        # the cfg-5 critic width
         (64, 6, 2, "circ", 64, True),         # 32 agents per rank: the 20-unit team-reward net is sharded too (packed operands)
         (128, 66, 32, "circ", 512, True)]     # BASELINE configs[4] at an eighth of the agents: d = 66, H = 32, 512-unit critic


def _setup(case):
    n, d, H, graph, hid, lattice = case
    rng = np.random.default_rng(n * 7 + d)
    if graph == "circ":
        nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    else:
        nodes = [[i] + [int(x) for x in rng.permutation([j for j in range(n) if j != i])[:d - 1]] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=H, n_episodes=9, max_ep_len=5, n_ep_fixed=4, n_epochs=2, buffer_size=30, seed=23,
                        in_nodes=nodes, fast_lr=0.002)
    W, goals = EC.make_inputs(args, 6, (23,), critic_hid=hid)
    return args, W, goals, hid, lattice


def _snapshot(eng, logs):
    out = {"theta_" + k: v.cpu().numpy() for k, v in eng.theta.items()}
    out.update({"adam_m": eng.adam_m.cpu().numpy(), "loss_critic": eng.loss["critic"].cpu().numpy(),
                "loss_tr": eng.loss["tr"].cpu().numpy()})
    out.update({"rp_" + k: v[:, :eng.B].cpu().numpy() for k, v in eng.rp.items()})
    out.update({"log_" + k: np.asarray(v) for k, v in logs.items()})
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d-d%d-%s-hid%d-%s" % (c[0], c[1], c[3], c[4], "lattice" if c[5] else "dense"))
def test_agent_sharded_wide_critic_two_ranks_on_one_gpu(case):
    lib = capi.load()
    args, W, goals, hid, lattice = _setup(case)
    ref_eng, ref_logs = EC.run_engine(args, 6, 6, "device", "cuda", lib, (23,), W, goals, lattice=lattice, critic_hid=hid)
    assert ref_eng.wide and ref_eng.lat_active == lattice
    assert all(bool(torch.isfinite(v).all()) for v in ref_eng.theta.values())
    ref = _snapshot(ref_eng, ref_logs)
    world = 2
    comms, results, errors = ThreadComm.make(world), [None] * world, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            eng, logs = EC.run_engine(args, 6, 6, "device", "cuda", lib, (23,), W, goals, lattice=lattice, critic_hid=hid,
                                      tweak=lambda e: e.shard_agents(comm=comms[r]))
            assert eng.shard is not None and eng.shard.n_loc == case[0] // world and not eng._windowed
            # the team-reward net joins whenever its packed operands split on 128-row tiles (or are not used)
            assert eng.shard.shard_tr == ((not lattice) or (eng.shard.n_loc * 20) % 128 == 0)
            assert sorted(eng.shard.sc) == (["critic", "tr"] if eng.shard.shard_tr else ["critic"])
            results[r] = _snapshot(eng, logs)
        except BaseException as e:            # noqa: BLE001 -- reported below; the peer's barrier breaks by timeout/abort
            errors.append((r, repr(e)))
            comms[r]._sh["barrier"].abort()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for r in range(world):
        assert sorted(results[r]) == sorted(ref)
        for k in ref:
            np.testing.assert_array_equal(results[r][k], ref[k], err_msg="rank %d %s" % (r, k))
