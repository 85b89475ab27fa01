"""C2 end to end (BASELINE configs[4] as ONE instance over several GPUs): RPBCACEngine.shard_agents
shards the wide critic of a single-seed instance by AGENTS (TD targets, local fits, estimate consensus, values) and by
parameter COLUMNS (hidden-layer consensus), with the exchanges of parallel.ShardedConsensus in between.  world_size 2
under gloo, kernels from the hipemu build: after two update blocks (and a trailing episode) every parameter of every network, the Adam slots,
the replay rows and the three logged curves equal the UNSHARDED engine's bit for bit, on both ranks.  CPU-only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

_HERE = os.path.dirname(os.path.abspath(__file__))

# (agents, d, H, graph, critic width, lattice path, rng mode)
CASES = [(4, 4, 1, "circ", 64, True, "device"),        # packed bf16x3 layer 1: 2 agents x 64 units = one 128-row tile per rank
         (6, 3, 1, "rand", 24, False, "numpy"),        # dense f32 path, general K1 kernel
         (4, 4, 1, "circ", 128, True, "device")]       # dense layers on pre-split packed operands (csrc/dense_pk.hip)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(case, lib, shard):
    import engine_checks as EC
    n, d, H, graph, hid, lattice, rng_mode = case
    rng = np.random.default_rng(n * 7 + d)
    if graph == "circ":
        nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    else:
        nodes = [[i] + [int(x) for x in rng.permutation([j for j in range(n) if j != i])[:d - 1]] for i in range(n)]
    n_epochs = 2 if hid % 128 == 0 else 1       # (two epochs: the second takes its TD target from the cached layer-2 activations)
    args = EC.make_args(["Cooperative"] * n, H=H, n_episodes=5, max_ep_len=3, n_ep_fixed=2, n_epochs=n_epochs, buffer_size=9, seed=17,
                        in_nodes=nodes)
    W, goals = EC.make_inputs(args, 5, (17,), critic_hid=hid)
    calls = {"exchange": {}, "rows": []}

    def tweak(eng):
        eng.shard_agents()
        assert eng.shard.shard_tr == (not lattice)           # 2 agents x 20 units do not fill a 128-row tile of the packed operands
        fit = eng._local_fit_wide

        def count(net, exchange):
            def counted_exchange(msg_local):
                calls["exchange"][net] = calls["exchange"].get(net, 0) + 1
                assert msg_local.shape[1] == n // 2
                return exchange(msg_local)
            return counted_exchange
        for net, sc in eng.shard.sc.items():
            sc.exchange = count(net, sc.exchange)

        def counted_fit(net, xkey, y, B, mask):
            calls["rows"].append((eng.N, y.shape[1], eng.msg[net].shape[1]))      # inside the window: this rank's agents only
            return fit(net, xkey, y, B, mask)
        eng._local_fit_wide = counted_fit
    eng, logs = EC.run_engine(args, 5, 5, rng_mode, "cpu", lib, (17,), W, goals, lattice=lattice, critic_hid=hid,
                              tweak=tweak if shard else None)
    assert eng.wide and eng.lat_active == lattice and (eng.shard is not None) == shard and not eng._windowed
    if shard:           # 2 update blocks x n_epochs: one transpose each way per epoch, fits on half of the agents
        ne = 2 * n_epochs
        assert calls["exchange"] == ({"critic": ne} if lattice else {"critic": ne, "tr": ne}), calls
        assert calls["rows"] == [(n // 2, n // 2, n // 2)] * ne, calls
    out = {"theta_" + k: v.numpy().copy() for k, v in eng.theta.items()}
    out.update({"adam_m": eng.adam_m.numpy().copy(), "adam_v": eng.adam_v.numpy().copy(),
                "loss_critic": eng.loss["critic"].numpy().copy(), "loss_tr": eng.loss["tr"].numpy().copy()})
    out.update({"rp_" + k: v[:, :eng.B].numpy().copy() for k, v in eng.rp.items()})
    out.update({"log_" + k: np.asarray(v) for k, v in logs.items()})
    return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(_HERE))
    sys.path.insert(0, _HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import emu_lib
    lib = emu_lib()
    for ci, case in enumerate(CASES):
        got = _run(case, lib, shard=True)
        np.savez(os.path.join(out_dir, "c%d_r%d.npz" % (ci, rank)), **got)
        if rank == 0:
            np.savez(os.path.join(out_dir, "c%d_ref.npz" % ci), **_run(case, lib, shard=False))
    dist.destroy_process_group()


def test_agent_sharded_wide_critic_equals_unsharded_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for ci in range(len(CASES)):
        ref = np.load(os.path.join(str(tmp_path), "c%d_ref.npz" % ci))
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "c%d_r%d.npz" % (ci, rank)))
            assert sorted(got.files) == sorted(ref.files)
            for k in ref.files:
                np.testing.assert_array_equal(got[k], ref[k], err_msg="case %d rank %d %s" % (ci, rank, k))


def test_shard_agents_refuses_what_it_cannot_shard():
    sys.path.insert(0, _HERE)
    import engine_checks as EC
    from emu_util import emu_lib
    from rcmarl_amd.engine import EngineConfig, RPBCACEngine
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=2, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=1)

    def engine(S, hid):
        cfg = EngineConfig(5, args["agent_label"], args["in_nodes"], H=1, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9,
                           n_seeds=S, rng_mode="device", lattice=False, critic_hid=hid)
        return RPBCACEngine(cfg, seeds=list(range(S)), device="cpu", lib=emu_lib())
    assert engine(1, 24).shard_agents(rank=0, world=1).shard is None           # one rank: nothing to shard
    with pytest.raises(ValueError, match="ONE instance"):
        engine(2, 24).shard_agents(rank=0, world=2)                            # several seeds shard by seed instead
    with pytest.raises(ValueError, match="ONE instance"):
        engine(1, 20).shard_agents(rank=0, world=2)                            # the reference's 20-unit critic: nothing to gain
    with pytest.raises(ValueError, match="multiple of the number of ranks"):
        engine(1, 24).shard_agents(rank=0, world=2)                            # 5 agents over 2 ranks
