"""Shared helpers for the test-suite (oracle-side utilities)."""
import json

import numpy as np

H20 = 20


def net_dims(n_agents, kind, n_states=2, n_actions=5, hidden=H20):
    in_dim = n_agents * (n_states + (1 if kind == "tr" else 0))
    out_dim = n_actions if kind == "actor" else 1
    return in_dim, hidden, out_dim


def unflatten(vec, in_dim, hidden, out_dim):
    """Flat fp32 vector in Keras order -> [W1,b1,W2,b2,W3,b3]."""
    shapes = [(in_dim, hidden), (hidden,), (hidden, hidden), (hidden,), (hidden, out_dim), (out_dim,)]
    out, o = [], 0
    for sh in shapes:
        n = int(np.prod(sh))
        out.append(np.asarray(vec[o:o + n], dtype=np.float32).reshape(sh).copy())
        o += n
    assert o == len(vec)
    return out


def flatten(params):
    return np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in params])


def golden_scenario(golden, name):
    """Returns (args, desired, init[i][net] param lists, final[i][net] flat, sim dict)."""
    args = json.loads(str(golden[f"train/{name}/args"]))
    n = args["n_agents"]
    init, final = [], []
    for i in range(n):
        init.append({k: unflatten(golden[f"train/{name}/init/{i}/{k}"], *net_dims(n, k)) for k in ("actor", "critic", "tr")})
        fin = {k: golden[f"train/{name}/final/{i}/{k}"] for k in ("actor", "critic", "tr")}
        key = f"train/{name}/final/{i}/critic_local"
        if key in golden.files:
            fin["critic_local"] = golden[key]
        final.append(fin)
    sim = {c: golden[f"train/{name}/sim/{c}"] for c in ("True_team_returns", "True_adv_returns", "Estimated_team_returns")}
    return args, golden[f"train/{name}/desired"], init, final, sim
