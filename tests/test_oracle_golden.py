"""Pin the oracle (oracle/) to fixtures produced by executing the reference's
own sources (tests/golden/make_golden.py).  CPU-only."""
import numpy as np
import pytest

from oracle import rpbcac_oracle as O
from helpers import flatten, golden_scenario, unflatten, net_dims


def _agg_cases(golden):
    return sorted({k.split("/")[1] for k in golden.files if k.startswith("agg/")})


def test_hand_kats():
    # hand-derived known answers of the aggregation rule
    for H, v, want in [(1, [0, 10, -10, 1], 0.5), (1, [10, 0, 1, 2], 3.5), (0, [10, 0, 1, 2], 3.25), (1, [5, 5, 5, 5], 5.0)]:
        got = O.resilient_aggregate(np.asarray(v, np.float32)[:, None], H)
        assert got.shape == (1,) and got[0] == np.float32(want)


def test_aggregation_matches_reference_fixtures(golden):
    names = _agg_cases(golden)
    assert len(names) >= 15
    for n in names:
        H, x, y = int(golden[f"agg/{n}/H"]), golden[f"agg/{n}/x"], golden[f"agg/{n}/y"]
        got = O.resilient_aggregate(x, H)
        np.testing.assert_array_equal(got, y, err_msg=n)


def test_hidden_consensus_on_shipped_weights(golden):
    in_nodes = golden["hid/in_nodes"]
    cm, tm = golden["hid/critic_msgs"], golden["hid/tr_msgs"]
    for H in (0, 1):
        for i in range(4):
            dc, dt = net_dims(5, "critic"), net_dims(5, "tr")
            ag = O.CoopAgent(unflatten(golden["train/coop_H0/init/0/actor"], *net_dims(5, "actor")),
                             unflatten(cm[i], *dc), unflatten(tm[i], *dt), 0.002, 0.01, 0.9, H)
            ag.consensus_hidden_critic([unflatten(cm[j], *dc) for j in in_nodes[i]])
            ag.consensus_hidden_tr([unflatten(tm[j], *dt) for j in in_nodes[i]])
            np.testing.assert_array_equal(flatten(ag.critic), golden[f"hid/H{H}/critic_after"][i])
            np.testing.assert_array_equal(flatten(ag.tr), golden[f"hid/H{H}/tr_after"][i])
            # output layer untouched (aggregated W3,b3 are discarded)
            np.testing.assert_array_equal(flatten(ag.critic)[-21:], cm[i][-21:])


@pytest.mark.parametrize("name", ["g5", "g16"])
def test_env_matches_reference_fixture(golden, name):
    nrow, ncol, n = [int(v) for v in golden[f"env/{name}/dims"]]
    np.random.seed(11)
    desired = np.random.randint(0, 5, size=(n, 2))
    np.testing.assert_array_equal(desired, golden[f"env/{name}/desired"])
    env = O.GridWorldOracle(nrow, ncol, n, desired, None, True, True)
    acts, raw, st, rw = (golden[f"env/{name}/{k}"] for k in ("actions", "raw_states", "states", "rewards"))
    k = r = 0
    for ep in range(3):
        env.reset()
        np.testing.assert_array_equal(env.state, raw[r]); r += 1
        for t in range(40):
            a = np.random.randint(0, 5, size=n).astype(np.float64)
            np.testing.assert_array_equal(a, acts[k])
            env.step(a)
            s, rew = env.get_data()
            np.testing.assert_array_equal(env.state, raw[r]); r += 1
            np.testing.assert_array_equal(s, st[k])
            np.testing.assert_array_equal(rew, rw[k]); k += 1


def _run_oracle_scenario(golden, name):
    args, desired, init, final, sim = golden_scenario(golden, name)
    np.random.seed(args["random_seed"])
    s_desired = np.random.randint(0, 5, size=(5, 2))
    s_initial = np.random.randint(0, 5, size=(5, 2))
    np.testing.assert_array_equal(s_desired, desired)
    agents = [O.make_agent(lab, init[i]["actor"], init[i]["critic"], init[i]["tr"], args["slow_lr"], args["fast_lr"],
                           args["gamma"], args["H"]) for i, lab in enumerate(args["agent_label"])]
    env = O.GridWorldOracle(5, 5, 5, s_desired, s_initial, True, True)
    weights, df = O.train(env, agents, args)
    return args, weights, df, final, sim


@pytest.mark.parametrize("name", ["coop_H0", "malicious_H1", "mixed_H1"])
def test_training_loop_matches_reference_fixture(golden, name):
    """End-to-end: the oracle's train() reproduces the reference's
    train_RPBCAC (run verbatim under the Keras shim) -- returns and final
    weights, including the adversaries' mini-batch fits."""
    args, weights, df, final, sim = _run_oracle_scenario(golden, name)
    for col in ("True_team_returns", "True_adv_returns", "Estimated_team_returns"):
        np.testing.assert_array_equal(df[col].to_numpy(dtype=np.float64), sim[col], err_msg=col)
    for i in range(5):
        for k, key in enumerate(("actor", "critic", "tr")):
            np.testing.assert_array_equal(flatten(weights[i][k]), final[i][key], err_msg=f"agent {i} {key}")
        if "critic_local" in final[i]:
            np.testing.assert_array_equal(flatten(weights[i][3]), final[i]["critic_local"])
