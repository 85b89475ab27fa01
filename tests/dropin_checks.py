"""Drop-in boundary checks shared by the hipemu (CPU) and GPU tests: the reference-shaped
Python surface (rcmarl_amd.agents / training / environments / main) against the oracle and
against the golden vectors produced by the reference's own train_RPBCAC (tests/golden)."""
import numpy as np

import helpers
from oracle import mlp_np as M
from oracle import rpbcac_oracle as O
from rcmarl_amd import keras_compat as K
from rcmarl_amd import single
from rcmarl_amd.agents import adversarial_CAC_agents as ADV
from rcmarl_amd.agents.resilient_CAC_agents import RPBCAC_agent
from rcmarl_amd.environments.grid_world import Grid_World
from rcmarl_amd.training.train_agents import train_RPBCAC

IN_NODES = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]]


def make_models(n_agents, weights=None):
    def mlp(width, out, act):
        return K.Sequential([K.Input(shape=(n_agents, width)), K.layers.Flatten(),
                             K.layers.Dense(20, activation=K.layers.LeakyReLU(alpha=0.1)),
                             K.layers.Dense(20, activation=K.layers.LeakyReLU(alpha=0.1)),
                             K.layers.Dense(out, activation=act)])
    trio = [mlp(2, 5, 'softmax'), mlp(2, 1, None), mlp(3, 1, None)]
    if weights is not None:
        for m, w in zip(trio, weights):
            m.set_weights(w)
    return trio


WORST = {}           # label -> worst |a - b| / max(1, |b|max) seen by close(): printed by the callers as "[parity]" lines


def close(a, b, rtol, what=""):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    key = what.split(" agent")[0].split(" after")[-1].strip() or "?"
    WORST[what[:24]] = max(WORST.get(what[:24], 0.0), err / scale)
    assert err <= rtol * scale, "%s: %.3e > %.1e*%.3g" % (what, err, rtol, scale)


def check_agent_methods(H=1, n=5, B=300, seed=3):
    """Every public method of RPBCAC_agent against the oracle's CoopAgent, method by method."""
    rng = np.random.default_rng(seed)
    K.set_seed(seed)
    actor, critic, tr = make_models(n)
    for m in (critic, tr):                      # non-zero biases
        w = m.get_weights()
        for k in (1, 3, 5):
            w[k] = (0.1 * rng.normal(size=w[k].shape)).astype(np.float32)
        m.set_weights(w)
    ag = RPBCAC_agent(actor, critic, tr, slow_lr=0.002, fast_lr=0.01, gamma=0.9, H=H)
    ref = O.CoopAgent(actor.get_weights(), critic.get_weights(), tr.get_weights(), 0.002, 0.01, 0.9, H)
    assert ag.n_actions == 5
    s = rng.normal(size=(B, n, 2)).astype(np.float32)
    ns = rng.normal(size=(B, n, 2)).astype(np.float32)
    a = rng.integers(0, 5, size=(B, n, 1)).astype(np.float32)
    sa = np.concatenate([s, a], axis=-1)
    r = rng.normal(size=(B, 1)).astype(np.float32)
    # forward passes
    close(critic(s), M.forward(ref.critic, s.reshape(B, -1)), 1e-5, "critic(s)")
    close(actor.predict(s[:3]), M.softmax(M.forward(ref.actor, s[:3].reshape(3, -1))), 1e-5, "actor.predict")
    # local fits: message returned, live net untouched
    before = [w.copy() for w in critic.get_weights()]
    msg_c, loss_c = ag.critic_update_local(s, ns, r)
    want_c, wl = ref.local_fit_critic(s, ns, r)
    for x, y in zip(msg_c, want_c):
        close(x, y, 1e-5, "critic message")
    assert abs(loss_c - wl) <= 1e-5 * max(1, abs(wl))
    for x, y in zip(critic.get_weights(), before):
        np.testing.assert_array_equal(x, y)
    msg_t, loss_t = ag.TR_update_local(sa, r)
    want_t, wl = ref.local_fit_tr(sa, r)
    for x, y in zip(msg_t, want_t):
        close(x, y, 1e-5, "TR message")
    # neighbour messages: perturbed copies + one outlier
    def neighbours(own, in_dim):
        out = [own]
        for k in range(3):
            p = [(w + 0.05 * rng.normal(size=w.shape)).astype(np.float32) for w in own]
            out.append(p)
        out[2] = [(50 * w).astype(np.float32) for w in out[2]]
        return out
    cin, tin = neighbours(want_c, 2 * n), neighbours(want_t, 3 * n)
    ag.resilient_consensus_critic_hidden(cin)
    ref.consensus_hidden_critic(cin)
    ag.resilient_consensus_TR_hidden(tin)
    ref.consensus_hidden_tr(tin)
    for x, y in zip(critic.get_weights(), ref.critic):
        close(x, y, 2e-6, "critic after hidden consensus")
    for x, y in zip(tr.get_weights(), ref.tr):
        close(x, y, 2e-6, "TR after hidden consensus")
    c_agg, t_agg = ag.resilient_consensus_critic(s, cin), ag.resilient_consensus_TR(sa, tin)
    close(c_agg, ref.consensus_estimates_critic(s, cin), 1e-5, "critic_agg")
    close(t_agg, ref.consensus_estimates_tr(sa, tin), 1e-5, "TR_agg")
    ag.critic_update_team(s, c_agg)
    ref.projection_step_critic(s, c_agg)
    ag.TR_update_team(sa, t_agg)
    ref.projection_step_tr(sa, t_agg)
    for x, y in zip(critic.get_weights(), ref.critic):
        close(x, y, 2e-5, "critic after team update")
    for x, y in zip(tr.get_weights(), ref.tr):
        close(x, y, 2e-5, "TR after team update")
    for _ in range(2):
        la = ag.actor_update(s, ns, sa, a[:, 0])
        lw = ref.actor_step(s, ns, sa, a[:, 0])
        assert abs(la - lw) <= 2e-5 * max(1, abs(lw)), (la, lw)
    for x, y in zip(actor.get_weights(), ref.actor):
        assert np.abs(x - y).max() <= 0.05 * 0.002 * 2 + 1e-6
    # get_action consumes the global NumPy stream exactly like the oracle
    np.random.seed(5)
    got = [ag.get_action(s[k:k + 1]) for k in range(10)]
    np.random.seed(5)
    ref.actor = [w.copy() for w in actor.get_weights()]
    want = [ref.act_numpy(s[k:k + 1]) for k in range(10)]
    assert got == want
    assert len(ag.get_parameters()) == 3
    # hand KATs of the aggregation rule
    for Hk, v, want_v in [(1, [0, 10, -10, 1], 0.5), (1, [10, 0, 1, 2], 3.5), (0, [10, 0, 1, 2], 3.25), (1, [5, 5, 5, 5], 5.0)]:
        ag.H = Hk
        out = ag._resilient_aggregation(np.asarray(v, np.float32)[:, None])
        assert out.shape == (1,) and abs(float(out[0]) - want_v) < 1e-6, (v, out)


def check_adversary_methods(n=5, B=100, seed=9):
    rng = np.random.default_rng(seed)
    K.set_seed(seed)
    s = rng.normal(size=(B, n, 2)).astype(np.float32)
    ns = rng.normal(size=(B, n, 2)).astype(np.float32)
    a = rng.integers(0, 5, size=(B, n, 1)).astype(np.float32)
    sa = np.concatenate([s, a], axis=-1)
    r = rng.normal(size=(B, 1)).astype(np.float32)
    # Greedy
    actor, critic, tr = make_models(n)
    g = ADV.Greedy_CAC_agent(actor, critic, tr, slow_lr=0.002, fast_lr=0.01, gamma=0.9)
    ref = O.GreedyAgent(actor.get_weights(), critic.get_weights(), tr.get_weights(), 0.002, 0.01, 0.9)
    ADV.set_shuffle_seed(77)
    sh = O.ShuffleStream(77)
    x, lt = g.TR_update_local(sa, r)
    y, lc = g.critic_update_local(s, ns, r)
    wx, wlt = ref.local_fit_tr(sa, r, sh)
    wy, wlc = ref.local_fit_critic(s, ns, r, sh)
    for p, q in zip(x + y, wx + wy):
        close(p, q, 5e-5, "greedy messages")
    assert abs(lt - wlt) <= 5e-5 * max(1, abs(wlt)) and abs(lc - wlc) <= 5e-5 * max(1, abs(wlc))
    for p, q in zip(critic.get_weights(), wy):              # no rollback
        close(p, q, 5e-5, "greedy keeps its fit")
    la = g.actor_update(s, ns, r, a[:, 0])
    lw = ref.actor_step(s, ns, r, a[:, 0], sh)
    assert abs(la - lw) <= 5e-5 * max(1, abs(lw))
    # Malicious
    actor, critic, tr = make_models(n)
    m = ADV.Malicious_CAC_agent(actor, critic, tr, slow_lr=0.002, fast_lr=0.01, gamma=0.9)
    ref = O.MaliciousAgent(actor.get_weights(), critic.get_weights(), tr.get_weights(), 0.002, 0.01, 0.9)
    m.critic_update_local(s, ns, r)
    ref.local_fit_private_critic(s, ns, r, sh)
    x, _ = m.TR_update_compromised(sa, -r)
    y, _ = m.critic_update_compromised(s, ns, -r)
    wx, _ = ref.fit_compromised_tr(sa, -r, sh)
    wy, _ = ref.fit_compromised_critic(s, ns, -r, sh)
    for p, q in zip(x + y + m.critic_local_weights, wx + wy + ref.critic_local):
        close(p, q, 5e-5, "malicious nets")
    assert len(m.get_parameters()) == 4
    # Faulty
    actor, critic, tr = make_models(n)
    f = ADV.Faulty_CAC_agent(actor, critic, tr, slow_lr=0.002, gamma=0.9)
    for p, q in zip(f.get_critic_weights() + f.get_TR_weights(), critic.get_weights() + tr.get_weights()):
        np.testing.assert_array_equal(p, q)


def check_env_golden(golden):
    """Grid_World drop-in against the reference env's recorded trajectories."""
    for name in ("g5", "g16"):
        nrow, ncol, n = [int(v) for v in golden[f"env/{name}/dims"]]
        np.random.seed(11)
        desired = np.random.randint(0, 5, size=(n, 2))
        np.testing.assert_array_equal(desired, golden[f"env/{name}/desired"])
        env = Grid_World(nrow=nrow, ncol=ncol, n_agents=n, desired_state=desired, initial_state=None,
                         randomize_state=True, scaling=True)
        acts, raw, st, rw = (golden[f"env/{name}/{k}"] for k in ("actions", "raw_states", "states", "rewards"))
        t = ri = 0
        for ep in range(3):
            env.reset()
            np.testing.assert_array_equal(env.state, raw[ri]); ri += 1
            for _ in range(40):
                a = np.random.randint(0, 5, size=n).astype(np.float64)
                np.testing.assert_array_equal(a, acts[t])
                env.step(a)
                s, r = env.get_data()
                np.testing.assert_array_equal(env.state, raw[ri]); ri += 1
                np.testing.assert_array_equal(s, st[t])
                np.testing.assert_array_equal(r, rw[t])
                t += 1


def check_train_golden(golden, name, engine_hook, rtol_w=1e-4):
    """train_RPBCAC drop-in vs the golden run of the REFERENCE's own train_RPBCAC source
    (tests/golden/make_golden.py): same seeds, same NumPy stream, same initial weights."""
    args, desired, init, final, sim = helpers.golden_scenario(golden, name)
    np.random.seed(args["random_seed"])                          # main.py:46-49
    s_desired = np.random.randint(0, 5, size=(5, 2))
    s_initial = np.random.randint(0, 5, size=(5, 2))
    np.testing.assert_array_equal(s_desired, desired)
    agents = []
    for i, lab in enumerate(args["agent_label"]):
        actor, critic, tr = make_models(5, [init[i]["actor"], init[i]["critic"], init[i]["tr"]])
        kw = dict(slow_lr=args["slow_lr"], gamma=args["gamma"])
        if lab == "Malicious":
            agents.append(ADV.Malicious_CAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], **kw))
        elif lab == "Faulty":
            agents.append(ADV.Faulty_CAC_agent(actor, critic, tr, **kw))
        elif lab == "Greedy":
            agents.append(ADV.Greedy_CAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], **kw))
        else:
            agents.append(RPBCAC_agent(actor, critic, tr, fast_lr=args["fast_lr"], H=args["H"], **kw))
    env = Grid_World(nrow=5, ncol=5, n_agents=5, desired_state=s_desired, initial_state=s_initial,
                     randomize_state=True, scaling=True)
    weights, df = train_RPBCAC(env, agents, args, engine_hook=engine_hook)
    assert list(df.columns) == ["True_team_returns", "True_adv_returns", "Estimated_team_returns"]
    assert len(df) == args["n_episodes"]
    # identical action streams => bit-identical float64 returns
    np.testing.assert_array_equal(df["True_team_returns"].to_numpy(), sim["True_team_returns"])
    np.testing.assert_array_equal(df["True_adv_returns"].to_numpy(), sim["True_adv_returns"])
    np.testing.assert_allclose(df["Estimated_team_returns"].to_numpy(), sim["Estimated_team_returns"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(env.state, golden[f"train/{name}/final_env_state"])
    worst, fails = {}, []
    for i in range(5):
        names = ["actor", "critic", "tr"] + (["critic_local"] if len(weights[i]) == 4 else [])
        assert len(weights[i]) == (4 if args["agent_label"][i] == "Malicious" else 3)
        for k, net in enumerate(names):
            got = helpers.flatten(weights[i][k])
            want = final[i][net]
            scale = max(1.0, float(np.abs(want).max()))
            tol = rtol_w * scale if net != "actor" else 0.05 * args["slow_lr"] * 12 + 1e-5
            err = float(np.abs(got - want).max())
            if net != "actor":
                worst[net] = max(worst.get(net, 0.0), err / scale)
            if err > tol:
                fails.append((name, i, args["agent_label"][i], net, err, tol))
    print("[parity] drop-in train_RPBCAC vs the reference's golden run '%s': worst |w - w_ref| / max(1,|w|max) %s  (bar %.0e)"
          % (name, "  ".join("%s %.2e" % kv for kv in sorted(worst.items())), rtol_w))
    assert not fails, fails


def check_train_wide_critic(engine_hook, critic_hid=32, seed=5):
    """train_RPBCAC with critic models wider than the reference's 20 units (BASELINE configs[4] in miniature): the
    width is read off the model objects; results vs the oracle's train() on the same NumPy stream."""
    import engine_checks as EC
    n = 5
    K.set_seed(seed)

    def mlp(width, out, act, hid):
        return K.Sequential([K.Input(shape=(n, width)), K.layers.Flatten(),
                             K.layers.Dense(hid, activation=K.layers.LeakyReLU(alpha=0.1)),
                             K.layers.Dense(hid, activation=K.layers.LeakyReLU(alpha=0.1)),
                             K.layers.Dense(out, activation=act)])
    agents, W = [], []
    for i in range(n):
        actor, critic, tr = mlp(2, 5, 'softmax', 20), mlp(2, 1, None, critic_hid), mlp(3, 1, None, 20)
        W.append([actor.get_weights(), critic.get_weights(), tr.get_weights()])
        agents.append(RPBCAC_agent(actor, critic, tr, slow_lr=0.002, fast_lr=0.01, gamma=0.9, H=1))
    args = EC.make_args(["Cooperative"] * n, H=1, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=seed)
    goals = np.array([[1, 2], [0, 0], [4, 4], [2, 3], [3, 1]])
    np.random.seed(seed)
    env = Grid_World(nrow=5, ncol=5, n_agents=n, desired_state=goals, initial_state=goals, randomize_state=True, scaling=True)
    weights, df = train_RPBCAC(env, agents, args, engine_hook=engine_hook)
    o_agents = [O.make_agent("Cooperative", [a.copy() for a in W[i][0]], [a.copy() for a in W[i][1]], [a.copy() for a in W[i][2]],
                             0.002, 0.01, 0.9, 1) for i in range(n)]
    np.random.seed(seed)
    oenv = O.GridWorldOracle(5, 5, n, goals, None, True, True)
    ow, odf = O.train(oenv, o_agents, args, rng_mode="numpy")
    np.testing.assert_array_equal(df["True_team_returns"].to_numpy(), odf["True_team_returns"].to_numpy(dtype=np.float64))
    np.testing.assert_allclose(df["Estimated_team_returns"].to_numpy(), odf["Estimated_team_returns"].to_numpy(dtype=np.float64),
                               rtol=1e-4, atol=1e-5)
    for i in range(n):
        assert weights[i][1][0].shape == (2 * n, critic_hid)
        for k in (1, 2):                                  # critic, team-reward net
            for a, b in zip(weights[i][k], ow[i][k]):
                close(a, b, 2e-4, "agent %d net %d" % (i, k))


def write_reference_artifacts(golden, directory, tag="mal_H1_s300"):
    """`pretrained_weights.npy` + `desired_state.npy` in the reference's OWN on-disk format (main.py:119-121: object array
    [agent][net][six Keras arrays], 4 nets for a Malicious agent), rebuilt from the shipped run
    simulation_results/raw_data/malicious/H=1/seed=300 carried in tests/golden (art/*)."""
    import os
    n = 5
    obj = np.empty(n, dtype=object)
    for i in range(n):
        nets = []
        for net in ("actor", "critic", "tr", "critic_local"):
            key = "art/%s/weights/%d/%s" % (tag, i, net)
            if key in golden.files:
                nets.append(helpers.unflatten(golden[key], *helpers.net_dims(n, "critic" if net == "critic_local" else net)))
        obj[i] = nets
    np.save(os.path.join(directory, "pretrained_weights.npy"), obj, allow_pickle=True)
    np.save(os.path.join(directory, "desired_state.npy"), golden["art/%s/desired_state" % tag], allow_pickle=True)
    return obj


def check_main_roundtrip(golden, tmp_path, engine_hook, n_episodes=100, n_ep_fixed=50, max_ep_len=20, n_epochs=3, buffer_size=2000):
    """rcmarl_amd.main against the reference's artefact contract (main.py:52-54 load, :119-121 save):
    1. warm start from the reference's SHIPPED weights of its malicious run (--pretrained_agents True), train, and
       compare with the oracle's train() warm-started from the same file on the same NumPy stream;
    2. the artefacts it writes have the reference's format;
    3. save -> --pretrained_agents load -> save (zero episodes) reproduces the weight file bit for bit."""
    import json
    import os
    import pandas as pd
    from rcmarl_amd import main as RM
    labels = ["Cooperative"] * 4 + ["Malicious"]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        shipped = write_reference_artifacts(golden, ".")
        argv = ["--pretrained_agents", "True", "--agent_label", json.dumps(labels), "--H", "1", "--slow_lr", "0.002",
                "--fast_lr", "0.01", "--random_seed", "300", "--n_episodes", str(n_episodes), "--n_ep_fixed", str(n_ep_fixed),
                "--max_ep_len", str(max_ep_len), "--n_epochs", str(n_epochs), "--buffer_size", str(buffer_size),
                "--rng_mode", "numpy"]
        weights, df = RM.main(argv, engine_hook=engine_hook)
        # ---- 2. artefact formats
        saved = np.load("pretrained_weights.npy", allow_pickle=True)
        assert saved.dtype == object and saved.shape == (5,)
        assert [len(saved[i]) for i in range(5)] == [3, 3, 3, 3, 4]
        for i in range(5):
            for k in range(len(saved[i])):
                assert [np.shape(a) for a in saved[i][k]] == [np.shape(a) for a in shipped[i][k]]
                for a, b in zip(saved[i][k], weights[i][k]):
                    np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.load("desired_state.npy", allow_pickle=True), golden["art/mal_H1_s300/desired_state"])
        sim = pd.read_pickle("sim_data.pkl")
        assert list(sim.columns) == ["True_team_returns", "True_adv_returns", "Estimated_team_returns"] and len(sim) == n_episodes
        # ---- 1. the same warm start in the oracle (main.py:46-49 draws s_desired, s_initial before the file replaces s_desired)
        np.random.seed(300)
        np.random.randint(0, 5, size=(5, 2))
        np.random.randint(0, 5, size=(5, 2))
        desired = golden["art/mal_H1_s300/desired_state"]
        o_agents = []
        for i, lab in enumerate(labels):
            w = [[np.array(a, np.float32) for a in net] for net in shipped[i]]
            ag = O.make_agent(lab, w[0], w[1], w[2], 0.002, 0.01, 0.9, 1)
            if lab == "Malicious":
                ag.critic_local = w[3]
            o_agents.append(ag)
        args = {"n_agents": 5, "agent_label": labels, "in_nodes": IN_NODES, "n_actions": 5, "n_states": 2, "n_episodes": n_episodes,
                "max_ep_len": max_ep_len, "n_ep_fixed": n_ep_fixed, "n_epochs": n_epochs, "slow_lr": 0.002, "fast_lr": 0.01,
                "batch_size": 200, "buffer_size": buffer_size, "gamma": 0.9, "H": 1, "common_reward": False, "random_seed": 300}
        oenv = O.GridWorldOracle(5, 5, 5, desired, None, True, True)
        ow, odf = O.train(oenv, o_agents, args, rng_mode="numpy")
        np.testing.assert_array_equal(df["True_team_returns"].to_numpy(), odf["True_team_returns"].to_numpy(dtype=np.float64))
        np.testing.assert_array_equal(df["True_adv_returns"].to_numpy(), odf["True_adv_returns"].to_numpy(dtype=np.float64))
        np.testing.assert_allclose(df["Estimated_team_returns"].to_numpy(), odf["Estimated_team_returns"].to_numpy(dtype=np.float64),
                                   rtol=1e-4, atol=1e-5)
        for i in range(5):
            for k in (1, 2) + ((3,) if len(ow[i]) == 4 else ()):
                for a, b in zip(weights[i][k], ow[i][k]):
                    close(a, b, 1e-4, "agent %d net %d after warm start" % (i, k))
        # a policy trained for 8000 episodes by the reference: its first block (before any update of ours) must already be good
        first = float(df["True_team_returns"].to_numpy()[:n_ep_fixed].mean())
        assert first > -6.5, "shipped policy's return on the new engine: %.3f (reference phase-2 band: -5.3 .. -5.6)" % first
        # ---- 3. save -> load -> save identity
        before = np.load("pretrained_weights.npy", allow_pickle=True)
        argv0 = list(argv)
        argv0[argv0.index("--n_episodes") + 1] = "0"
        w0, df0 = RM.main(argv0, engine_hook=engine_hook)
        assert len(df0) == 0
        after = np.load("pretrained_weights.npy", allow_pickle=True)
        for i in range(5):
            assert len(before[i]) == len(after[i])
            for k in range(len(before[i])):
                for a, b in zip(before[i][k], after[i][k]):
                    np.testing.assert_array_equal(a, b)
    finally:
        os.chdir(cwd)
