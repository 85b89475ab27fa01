"""Engine-vs-oracle end-to-end checks shared by the hipemu (CPU) and GPU tests."""
import numpy as np

from oracle import mlp_np as M
from oracle import rpbcac_oracle as O
from rcmarl_amd.engine import EngineConfig, RPBCACEngine

CIRC5 = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]]


def make_args(labels, H, n_episodes, max_ep_len, n_ep_fixed, n_epochs, buffer_size, seed, in_nodes=None,
              common_reward=False, slow_lr=0.002, fast_lr=0.01, gamma=0.9):
    n = len(labels)
    return {"n_agents": n, "agent_label": list(labels), "in_nodes": in_nodes or CIRC5, "n_actions": 5, "n_states": 2,
            "n_episodes": n_episodes, "max_ep_len": max_ep_len, "n_ep_fixed": n_ep_fixed, "n_epochs": n_epochs,
            "slow_lr": slow_lr, "fast_lr": fast_lr, "batch_size": 200, "buffer_size": buffer_size, "gamma": gamma, "H": H,
            "common_reward": common_reward, "summary_dir": "./", "pretrained_agents": False, "random_seed": seed}


def init_weights(rng, n_agents, critic_hid=20):
    out = []
    for _ in range(n_agents):
        out.append({"actor": M.init_mlp(rng, 2 * n_agents, 20, 5), "critic": M.init_mlp(rng, 2 * n_agents, critic_hid, 1),
                    "tr": M.init_mlp(rng, 3 * n_agents, 20, 1)})
    return out


def make_inputs(args, nrow, seeds, weight_seed=3, critic_hid=20):
    """Initial weights and goals of every seed (shared by the oracle run and the engine run)."""
    n = args["n_agents"]
    wrng = np.random.default_rng(weight_seed)
    W = [init_weights(wrng, n, critic_hid) for _ in seeds]
    goals = [np.random.default_rng(100 + s).integers(0, min(5, nrow), size=(n, 2)) for s in range(len(seeds))]
    return W, goals


def run_oracle(args, nrow, ncol, rng_mode, seeds, W, goals):
    """oracle.train, one run per seed -> (per-seed DataFrames, per-seed weight lists)."""
    n = args["n_agents"]
    o_logs, o_weights = [], []
    for s in range(len(seeds)):
        a = dict(args)
        a["random_seed"] = int(seeds[s])
        agents = [O.make_agent(lab, W[s][i]["actor"], W[s][i]["critic"], W[s][i]["tr"], a["slow_lr"], a["fast_lr"], a["gamma"], a["H"])
                  for i, lab in enumerate(a["agent_label"])]
        if rng_mode == "numpy":
            np.random.seed(int(seeds[s]))
            env = O.GridWorldOracle(nrow, ncol, n, goals[s], None, True, True)
            w, df = O.train(env, agents, a, rng_mode="numpy")
        else:
            env = O.GridWorldOracle(nrow, ncol, n, goals[s], None, True, True, rng_mode="device", seed=int(seeds[s]))
            w, df = O.train(env, agents, a, rng_mode="device")
        o_logs.append(df)
        o_weights.append(w)
    return o_logs, o_weights


def run_engine(args, nrow, ncol, rng_mode, device, lib, seeds, W, goals, lattice="auto", critic_hid=20, tweak=None):
    """The batched engine on all seeds from the same weights/goals.  tweak(eng): instance-level switches
    (e.g. eng.td_shortcut = False) applied before training."""
    n, S = args["n_agents"], len(seeds)
    cfg = EngineConfig(n, args["agent_label"], args["in_nodes"], H=args["H"], gamma=args["gamma"], slow_lr=args["slow_lr"],
                       fast_lr=args["fast_lr"], max_ep_len=args["max_ep_len"], n_ep_fixed=args["n_ep_fixed"],
                       n_epochs=args["n_epochs"], buffer_size=args["buffer_size"], common_reward=args["common_reward"],
                       nrow=nrow, ncol=ncol, n_seeds=S, rng_mode=rng_mode, lattice=lattice, critic_hid=critic_hid)
    eng = RPBCACEngine(cfg, seeds=list(seeds), device=device, lib=lib)
    for s in range(S):
        for i in range(n):
            for net in ("actor", "critic", "tr"):
                eng.set_weights(s, i, net, W[s][i][net])
    eng.set_goals(np.stack(goals))
    if tweak is not None:
        tweak(eng)
    if rng_mode == "numpy":
        eng.np_rngs = []
        for s in range(S):
            r = np.random.RandomState(int(seeds[s]))
            r.randint([0, 0], [nrow, ncol], size=(n, 2))       # the env constructor's reset() draw (grid_world.py:28)
            eng.np_rngs.append(r)
    logs = eng.train(args["n_episodes"])
    return eng, logs


def run_pair(args, nrow, ncol, rng_mode, device, lib, seeds=(11,), weight_seed=3, lattice="auto", critic_hid=20):
    """Run the oracle (one run per seed) and the engine (all seeds batched); return both results."""
    W, goals = make_inputs(args, nrow, seeds, weight_seed, critic_hid)
    o_logs, o_weights = run_oracle(args, nrow, ncol, rng_mode, seeds, W, goals)
    eng, logs = run_engine(args, nrow, ncol, rng_mode, device, lib, seeds, W, goals, lattice, critic_hid)
    return eng, logs, o_logs, o_weights


def compare(eng, logs, o_logs, o_weights, rtol_w=2e-4, actor="strict"):
    """actor="strict": every actor parameter within 5 % of an Adam step per update.  actor="stat" (hundreds of agents):
    Adam turns a gradient of magnitude ~eps into anything in [-lr, lr] and a pre-activation within rounding of 0 flips its
    LeakyReLU slope, so among millions of parameters a few legitimately differ by more between any two fp32 summation
    orders: bulk within 5 % of a step, at most 1e-4 of the entries beyond, none beyond two full steps (the bar of
    kernel_checks.check_actor_step)."""
    S, n = eng.S, eng.N
    steps = max(1, eng.adam_t)
    for s in range(S):
        df = o_logs[s]
        # identical action streams -> bit-identical float64 returns
        np.testing.assert_array_equal(logs["True_team_returns"][:, s], df["True_team_returns"].to_numpy(dtype=np.float64))
        np.testing.assert_array_equal(logs["True_adv_returns"][:, s], df["True_adv_returns"].to_numpy(dtype=np.float64))
        np.testing.assert_allclose(logs["Estimated_team_returns"][:, s], df["Estimated_team_returns"].to_numpy(dtype=np.float64),
                                   rtol=1e-4, atol=1e-5)
        actor_err = []
        for i in range(n):
            for k, net in enumerate(("actor", "critic", "tr")):
                got = eng.get_weights(s, i, net)
                for a, b in zip(got, o_weights[s][i][k]):
                    scale = max(1.0, float(np.abs(b).max()))
                    err = float(np.abs(a - b).max())
                    if net == "actor" and actor == "stat":
                        actor_err.append(np.abs(a - b).ravel())
                        continue
                    tol = rtol_w * scale if net != "actor" else 0.05 * eng.cfg.slow_lr * steps + 1e-5
                    assert err <= tol, (s, i, net, err, tol)
            if len(o_weights[s][i]) == 4:                       # Malicious: private critic (adversarial:180-182)
                got = eng.get_weights(s, i, "critic_local")
                for a, b in zip(got, o_weights[s][i][3]):
                    scale = max(1.0, float(np.abs(b).max()))
                    assert float(np.abs(a - b).max()) <= rtol_w * scale, (s, i, "critic_local")
        if actor_err:
            e = np.concatenate(actor_err)
            lr = eng.cfg.slow_lr
            assert e.max() <= 2.0 * lr * steps + 1e-6, ("actor", float(e.max()))
            frac = float(np.mean(e > 0.05 * lr * steps + 1e-5))
            assert frac <= 1e-4, ("actor outliers", frac, float(e.max()))


def check_checkpoint_resume(labels, rng_mode, device, lib, path, n=5, nrow=5, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9,
                            S=2, blocks=(1, 2), lattice=True, d=4):
    """Train blocks[0]+blocks[1] blocks straight vs blocks[0] -> save -> FRESH engine -> load -> blocks[1]: same logs, same
    bits (weights, Adam slots and step counts, replay rows, agent positions, RNG position all travel in the file)."""
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]

    def make():
        cfg = EngineConfig(n, labels, in_nodes, H=1, max_ep_len=max_ep_len, n_ep_fixed=n_ep_fixed, n_epochs=n_epochs,
                           buffer_size=buffer_size, nrow=nrow, ncol=nrow, n_seeds=S, rng_mode=rng_mode, lattice=lattice)
        eng = RPBCACEngine(cfg, seeds=[7 + s for s in range(S)], device=device, lib=lib)
        eng.init_glorot(base_seed=3)
        eng.set_goals(np.random.default_rng(9).integers(0, min(5, nrow), size=(n, 2)))
        if rng_mode == "numpy":
            eng.np_rngs = [np.random.RandomState(70 + s) for s in range(S)]
        return eng
    e1, e2 = blocks[0] * n_ep_fixed, blocks[1] * n_ep_fixed
    a = make()
    la = a.train(e1 + e2)
    b = make()
    lb1 = b.train(e1)
    b.save_checkpoint(path)
    c = make()
    c.init_glorot(base_seed=99)                       # different weights: everything must come from the file
    c.load_checkpoint(path)
    lc = c.train(e2)
    for k in la:
        np.testing.assert_array_equal(la[k], np.concatenate([lb1[k], lc[k]], axis=0))
    for net in a.theta:
        np.testing.assert_array_equal(a.get_all_weights(net), c.get_all_weights(net))
    np.testing.assert_array_equal(a.adam_m.cpu().numpy(), c.adam_m.cpu().numpy())
    np.testing.assert_array_equal(a.adam_v.cpu().numpy(), c.adam_v.cpu().numpy())
    for k in a.rp:
        np.testing.assert_array_equal(a.rp[k][:, :a.B].cpu().numpy(), c.rp[k][:, :c.B].cpu().numpy())
    # a checkpoint of a different scenario is refused, not silently loaded
    other = EngineConfig(n, labels, in_nodes, H=0, max_ep_len=max_ep_len, n_ep_fixed=n_ep_fixed, n_epochs=n_epochs,
                         buffer_size=buffer_size, nrow=nrow, ncol=nrow, n_seeds=S, rng_mode=rng_mode, lattice=lattice)
    wrong = RPBCACEngine(other, seeds=[7 + s for s in range(S)], device=device, lib=lib)
    try:
        wrong.load_checkpoint(path)
    except ValueError:
        pass
    else:
        raise AssertionError("a checkpoint written with H=1 loaded into an H=0 engine")
