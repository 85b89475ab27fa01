"""Drop-in for the reference's ``training/train_agents.py``.

``train_RPBCAC(env, agents, args, exp_buffer=None) -> (weights, DataFrame)`` has
the reference's signature and return types (training/train_agents.py:15,
175-184) but none of its per-agent Python loops: the agents' networks are
stacked into parameter matrices, the whole loop -- rollout (:46-80), update
block (:86-163), episode summaries (:168-180) -- runs in the batched engine on
one MI355X, and the trained weights are written back into the agent objects.

Extra (optional) ``args`` keys on top of the reference's (main.py:26-44):
  rng_mode   'numpy' (default): actions and resets consume NumPy's global legacy
             stream exactly as the reference does (agents:208-219, grid_world:40);
             'device': counter-based Philox stream on the GPU (no host round trip)
  verbose    print one line per episode like the reference (:174)
"""
import numpy as np
import pandas as pd

from ..engine import EngineConfig, RPBCACEngine, flatten_params

_NETS = (("actor", "actor"), ("critic", "critic"), ("tr", "TR"))


def _truthy(v):
    return bool(v)            # the reference's flags have no `type=`: any non-empty CLI string is truthy


def train_RPBCAC(env, agents, args, exp_buffer=None, engine_hook=None):
    n_agents = env.n_agents
    labels = list(args['agent_label'])
    # hidden widths come from the model objects (the reference builds 20-unit networks, main.py:59-82): the critic may
    # be wider (BASELINE configs[4]: 512 units -> dense-GEMM path of the engine); actor and team-reward net stay at 20
    widths = {attr: {int(np.shape(getattr(ag, attr).get_weights()[0])[1]) for ag in agents} for _, attr in _NETS}
    if widths["actor"] != {20} or widths["TR"] != {20} or len(widths["critic"]) != 1:
        raise ValueError("unsupported hidden widths %r: actor and team-reward net must have 20 hidden units and all critics "
                         "the same width" % widths)
    cfg = EngineConfig(n_agents, labels, args['in_nodes'], critic_hid=widths["critic"].pop(), H=args['H'], gamma=args['gamma'],
                       slow_lr=args['slow_lr'],
                       fast_lr=args['fast_lr'], n_actions=args['n_actions'], n_states=args['n_states'],
                       max_ep_len=args['max_ep_len'], n_ep_fixed=args['n_ep_fixed'], n_epochs=args['n_epochs'],
                       buffer_size=args['buffer_size'], common_reward=_truthy(args['common_reward']), nrow=env.nrow,
                       ncol=env.ncol, n_seeds=1, rng_mode=args.get('rng_mode', 'numpy'),
                       scaling=bool(getattr(env, 'scaling', not np.isscalar(env.mean_state))),
                       randomize_state=env.randomize_state)
    for ag in agents:
        if getattr(ag, 'H', cfg.H) != cfg.H:
            raise ValueError("all cooperative agents must use H = args['H']")
    lib, device = engine_hook if engine_hook is not None else (None, "cuda")
    eng = RPBCACEngine(cfg, seeds=[int(args.get('random_seed', 0))], device=device, lib=lib)
    # ---- agents -> stacked parameter matrices
    for i, ag in enumerate(agents):
        for net, attr in _NETS:
            eng.set_weights(0, i, net, getattr(ag, attr).get_weights())
        if labels[i] == 'Malicious':
            eng.set_weights(0, i, "critic_local", ag.critic_local_weights)
        adam = getattr(ag, '_adam', None)
        if adam is not None and adam["t"] > 0:
            eng.load_adam(0, i, adam["m"], adam["v"], adam["t"])
    eng.set_goals(np.asarray(env.desired_state))
    eng.initial_state = env.initial_state
    if exp_buffer:
        eng.load_replay(exp_buffer[0], exp_buffer[1], exp_buffer[2], exp_buffer[3])
    if cfg.rng_mode == 'numpy':
        eng.np_rngs = [np.random]              # the global legacy stream, consumed in the reference's order
    # ---- the whole loop on the GPU
    logs = eng.train(args['n_episodes'])
    # ---- stacked parameter matrices -> agents
    for i, ag in enumerate(agents):
        for net, attr in _NETS:
            getattr(ag, attr).set_weights(eng.get_weights(0, i, net))
        if labels[i] == 'Malicious':
            ag.critic_local_weights = eng.get_weights(0, i, "critic_local")
        if hasattr(ag, '_adam'):
            ag._adam["m"], ag._adam["v"], ag._adam["t"] = eng.dump_adam(0, i)
    if exp_buffer:                             # the reference appends to the caller's lists in place (:36-40, 76-80)
        for lst, new in zip(exp_buffer, eng.dump_replay()):
            lst[:] = new
    env.state = eng.pos[eng.cur][0].detach().cpu().numpy().astype(np.int64)
    sim_data = pd.DataFrame({"True_team_returns": logs["True_team_returns"][:, 0],
                             "True_adv_returns": logs["True_adv_returns"][:, 0],
                             "Estimated_team_returns": logs["Estimated_team_returns"][:, 0]})
    if args.get('verbose'):
        for t in range(len(sim_data)):
            print('| Episode: {} | Est. returns: {} | Returns: {} '.format(t, sim_data["Estimated_team_returns"][t],
                                                                          sim_data["True_team_returns"][t]))
    weights = [agent.get_parameters() for agent in agents]
    return weights, sim_data
