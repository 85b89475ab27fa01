#!/usr/bin/env python
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN SOURCES.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference ships no tests and no golden vectors, so these
fixtures are produced by importing its modules verbatim under the numpy
`tensorflow`/`gym` stubs (tests/ref_shims).  What this pins: the aggregation
rule, the consensus bookkeeping, the training-loop orchestration, the RNG call
order and the grid-world.  What it cannot pin: TensorFlow's kernel numerics
(the stub's Keras arithmetic is oracle/mlp_np.py).

Fixtures are small on purpose; they travel to the GPU box, the reference does not.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root
import ref_harness  # noqa: E402

REF = ref_harness.load_reference()
keras = REF.keras


def flat_net(weights):
    return np.concatenate([np.asarray(w, dtype=np.float32).ravel() for w in weights])


# ----------------------------------------------------------------------------
def gen_aggregation(out):
    """reference RPBCAC_agent._resilient_aggregation on hand KATs + random inputs."""
    agg = REF.resilient.RPBCAC_agent._resilient_aggregation
    rng = np.random.default_rng(7)
    cases = {}
    kats = [(1, [0, 10, -10, 1]), (1, [10, 0, 1, 2]), (0, [10, 0, 1, 2]), (1, [5, 5, 5, 5])]
    for n, (H, v) in enumerate(kats):
        x = np.asarray(v, np.float32)[:, None]
        cases[f"kat{n}"] = (H, x)
    for d, H in [(2, 0), (3, 1), (4, 0), (4, 1), (5, 2), (6, 2), (10, 4), (18, 8), (18, 1), (7, 3), (34, 16), (66, 32), (9, 0)]:
        x = rng.normal(size=(d, 97)).astype(np.float32)
        # inject ties, outliers and signed zeros
        x[:, 3] = x[0, 3]
        x[1:, 5] = np.float32(1e3)
        x[d // 2, 7] = np.float32(-1e30)
        x[:, 11] = np.where(np.arange(d) % 2 == 0, np.float32(0.0), np.float32(-0.0))
        cases[f"rand_d{d}_H{H}"] = (H, x)
    for name, (H, x) in cases.items():
        y = np.asarray(agg(types.SimpleNamespace(H=H), REF.tf.convert_to_tensor(x)))
        out[f"agg/{name}/H"] = np.int64(H)
        out[f"agg/{name}/x"] = x
        out[f"agg/{name}/y"] = y.astype(np.float32)


def gen_hidden_consensus_fixture(out):
    """resilient_consensus_critic_hidden / _TR_hidden on the shipped *trained*
    weights of a malicious run (real adversarial messages)."""
    path = os.path.join(ref_harness.REF_ROOT, "simulation_results/raw_data/malicious/H=1/seed=300/pretrained_weights2.npy")
    W = np.load(path, allow_pickle=True)
    in_nodes = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]]
    n = 5
    critic_msgs = [[np.asarray(a, np.float32) for a in W[i][1]] for i in range(n)]
    tr_msgs = [[np.asarray(a, np.float32) for a in W[i][2]] for i in range(n)]
    out["hid/critic_msgs"] = np.stack([flat_net(m) for m in critic_msgs])
    out["hid/tr_msgs"] = np.stack([flat_net(m) for m in tr_msgs])
    out["hid/in_nodes"] = np.asarray(in_nodes, np.int64)
    for H in (0, 1):
        res_c, res_t = [], []
        for i in range(4):                                   # cooperative agents 0..3
            models = build_models(n, seed=1)[i]
            ag = REF.resilient.RPBCAC_agent(models[0], models[1], models[2], slow_lr=0.002, fast_lr=0.01, gamma=0.9, H=H)
            ag.critic.set_weights(critic_msgs[i])
            ag.TR.set_weights(tr_msgs[i])
            ag.resilient_consensus_critic_hidden([critic_msgs[j] for j in in_nodes[i]])
            ag.resilient_consensus_TR_hidden([tr_msgs[j] for j in in_nodes[i]])
            res_c.append(flat_net(ag.critic.get_weights()))
            res_t.append(flat_net(ag.TR.get_weights()))
        out[f"hid/H{H}/critic_after"] = np.stack(res_c)
        out[f"hid/H{H}/tr_after"] = np.stack(res_t)


def gen_shipped_artifacts(out):
    """The reference's shipped artefacts of one run in full (main.py:119-121 wrote them): every agent's actor / critic /
    team-reward weights (+ the Malicious agent's private critic) and the desired state, so that tests on the GPU box can
    rebuild `pretrained_weights.npy` / `desired_state.npy` in the reference's own on-disk format and warm-start from them
    (main.py:52-54)."""
    d = os.path.join(ref_harness.REF_ROOT, "simulation_results/raw_data/malicious/H=1/seed=300")
    W = np.load(os.path.join(d, "pretrained_weights2.npy"), allow_pickle=True)
    names = ("actor", "critic", "tr", "critic_local")
    for i in range(len(W)):
        for k, net in enumerate(W[i]):
            out[f"art/mal_H1_s300/weights/{i}/{names[k]}"] = flat_net(net)
    out["art/mal_H1_s300/desired_state"] = np.asarray(np.load(os.path.join(d, "desired_state.npy"), allow_pickle=True), np.int64)


def gen_env(out):
    """Grid_World stepped with a random action stream."""
    for name, (nrow, ncol, n) in {"g5": (5, 5, 5), "g16": (16, 16, 12)}.items():
        np.random.seed(11)
        desired = np.random.randint(0, 5, size=(n, 2))
        env = REF.grid_world.Grid_World(nrow=nrow, ncol=ncol, n_agents=n, desired_state=desired,
                                        initial_state=None, randomize_state=True, scaling=True)
        acts, raw_states, states, rewards = [], [], [], []
        for ep in range(3):
            env.reset()
            raw_states.append(env.state.copy())
            for t in range(40):
                a = np.random.randint(0, 5, size=n).astype(np.float64)
                env.step(a)
                s, r = env.get_data()
                acts.append(a.copy()); raw_states.append(env.state.copy()); states.append(s.copy()); rewards.append(r.copy())
        out[f"env/{name}/dims"] = np.asarray([nrow, ncol, n])
        out[f"env/{name}/desired"] = desired
        out[f"env/{name}/actions"] = np.asarray(acts)
        out[f"env/{name}/raw_states"] = np.asarray(raw_states)
        out[f"env/{name}/states"] = np.asarray(states)
        out[f"env/{name}/rewards"] = np.asarray(rewards)


# ----------------------------------------------------------------------------
def build_models(n_agents, seed, n_states=2, n_actions=5, hidden=20):
    """Three small MLPs per agent with the architecture of reference main.py:59-82."""
    keras.set_init_seed(seed)
    L = keras.layers
    nets = []
    for _ in range(n_agents):
        trio = []
        for in_cols, out_units, act in ((n_states, n_actions, "softmax"), (n_states, 1, None), (n_states + 1, 1, None)):
            trio.append(keras.Sequential([
                keras.Input(shape=(n_agents, in_cols)), L.Flatten(),
                L.Dense(hidden, activation=L.LeakyReLU(alpha=0.1)),
                L.Dense(hidden, activation=L.LeakyReLU(alpha=0.1)),
                L.Dense(out_units, activation=act)]))
        nets.append(trio)
    return nets


SCENARIOS = {
    "coop_H0": dict(labels=["Cooperative"] * 5, H=0, seed=100, common_reward=False),
    "malicious_H1": dict(labels=["Cooperative"] * 4 + ["Malicious"], H=1, seed=300, common_reward=False),
    "mixed_H1": dict(labels=["Cooperative", "Cooperative", "Greedy", "Cooperative", "Faulty"], H=1, seed=200,
                     common_reward=True),
}


def scenario_args(sc):
    return {
        "n_agents": 5, "agent_label": sc["labels"],
        "in_nodes": [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]],
        "n_actions": 5, "n_states": 2, "n_episodes": 50, "max_ep_len": 10, "n_ep_fixed": 25, "n_epochs": 2,
        "slow_lr": 0.002, "fast_lr": 0.01, "batch_size": 200, "buffer_size": 400, "gamma": 0.9, "H": sc["H"],
        "common_reward": sc["common_reward"], "summary_dir": "./", "pretrained_agents": False,
        "random_seed": sc["seed"],
    }


def gen_training(out):
    """Whole reference training loop (train_RPBCAC) on three tiny scenarios."""
    from oracle.rpbcac_oracle import ShuffleStream
    for name, sc in SCENARIOS.items():
        args = scenario_args(sc)
        np.random.seed(args["random_seed"])                        # main.py:46-49
        s_desired = np.random.randint(0, 5, size=(5, 2))
        s_initial = np.random.randint(0, 5, size=(5, 2))
        nets = build_models(5, seed=args["random_seed"])
        init = [[flat_net(m.get_weights()) for m in trio] for trio in nets]
        agents = []
        for i, lab in enumerate(args["agent_label"]):              # main.py:88-104
            actor, critic, tr = nets[i]
            kw = dict(slow_lr=args["slow_lr"], gamma=args["gamma"])
            if lab == "Malicious":
                agents.append(REF.adversarial.Malicious_CAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], **kw))
            elif lab == "Faulty":
                agents.append(REF.adversarial.Faulty_CAC_agent(actor, critic, tr, **kw))
            elif lab == "Greedy":
                agents.append(REF.adversarial.Greedy_CAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], **kw))
            else:
                agents.append(REF.resilient.RPBCAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], H=args["H"], **kw))
        env = REF.grid_world.Grid_World(nrow=5, ncol=5, n_agents=5, desired_state=s_desired, initial_state=s_initial,
                                        randomize_state=True, scaling=True)
        keras.set_shuffle_stream(ShuffleStream(args["random_seed"]))
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            weights, sim = REF.train_agents.train_RPBCAC(env, agents, args)
        out[f"train/{name}/args"] = np.asarray(json.dumps(args))
        out[f"train/{name}/desired"] = s_desired
        for i in range(5):
            for k, netname in enumerate(["actor", "critic", "tr"]):
                out[f"train/{name}/init/{i}/{netname}"] = init[i][k]
                out[f"train/{name}/final/{i}/{netname}"] = flat_net(weights[i][k])
            if len(weights[i]) == 4:
                out[f"train/{name}/final/{i}/critic_local"] = flat_net(weights[i][3])
        for col in sim.columns:
            out[f"train/{name}/sim/{col}"] = sim[col].to_numpy(dtype=np.float64)
        out[f"train/{name}/final_env_state"] = env.state.copy()
        print(name, "last returns", sim["True_team_returns"].to_numpy()[-3:])


if __name__ == "__main__":
    out = {}
    gen_aggregation(out)
    gen_hidden_consensus_fixture(out)
    gen_shipped_artifacts(out)
    gen_env(out)
    gen_training(out)
    path = os.path.join(HERE, "reference_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
