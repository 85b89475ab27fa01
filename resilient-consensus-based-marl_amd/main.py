"""Drop-in for the reference's ``main.py``: same flags (main.py:25-45), same model
factory (:59-82), same agent dispatch by label (:88-104), same artefacts
(``sim_data.pkl``, ``pretrained_weights.npy``, ``desired_state.npy``; :119-121),
with the training loop running on the MI355X.

    python -m rcmarl_amd.main --H 1 --random_seed 100 [--n_episodes 4000 ...]

Additions: ``--nrow/--ncol`` (the reference hard-codes a 5x5 grid, :109-110),
``--rng_mode numpy|device`` (see training/train_agents.py), and
``--agent_label`` / ``--in_nodes`` accept JSON so scenarios can be set from the
command line (in the reference they can only be changed by editing the defaults).
"""
import argparse
import json

import numpy as np

from . import keras_compat as keras
from .agents.adversarial_CAC_agents import Faulty_CAC_agent, Greedy_CAC_agent, Malicious_CAC_agent, set_shuffle_seed
from .agents.resilient_CAC_agents import RPBCAC_agent
from .environments.grid_world import Grid_World
from .training import train_agents as training


def _json_or_str(v):
    try:
        return json.loads(v)
    except (TypeError, ValueError):
        return v


def build_parser():
    parser = argparse.ArgumentParser(description='RPBCAC training on one MI355X (flag names and defaults of the reference main.py)')
    parser.add_argument('--n_agents', help='agents in the team (adversaries included)', type=int, default=5)
    parser.add_argument('--agent_label', help='classification of each agent (Cooperative,Malicious,Faulty,Greedy), JSON list',
                        type=_json_or_str, default=['Cooperative'] * 5)
    parser.add_argument('--in_nodes', help='in-neighbourhood of each agent, own index first, JSON list of lists',
                        type=_json_or_str, default=[[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 0], [3, 4, 0, 1], [4, 0, 1, 2]])
    parser.add_argument('--n_actions', help='discrete actions per agent (the grid-world has 5)', type=int, default=5)
    parser.add_argument('--n_states', help='coordinates per agent (the grid-world has 2)', type=int, default=2)
    parser.add_argument('--n_episodes', help='episodes to run', type=int, default=7000)
    parser.add_argument('--max_ep_len', help='environment steps in one episode', type=int, default=20)
    parser.add_argument('--n_ep_fixed', help='episodes rolled out between two update blocks', type=int, default=50)
    parser.add_argument('--n_epochs', help='local-fit + consensus rounds per update block', type=int, default=10)
    parser.add_argument('--slow_lr', help='Adam step size of the actors', type=float, default=0.01)
    parser.add_argument('--fast_lr', help='SGD step size of critic and team-reward nets', type=float, default=0.01)
    parser.add_argument('--batch_size', help='batch size for policy evaluation (unused, as in the reference)', type=int, default=200)
    parser.add_argument('--buffer_size', help='replay rows kept after an update block', type=int, default=2000)
    parser.add_argument('--gamma', help='discount', type=float, default=0.9)
    parser.add_argument('--H', help='values trimmed on each side by the resilient aggregation', type=int, default=0)
    parser.add_argument('--common_reward', help='any non-empty value: cooperative agents fit on the team-average reward', default=False)
    parser.add_argument('--summary_dir', help='kept for CLI compatibility (unused, as in the reference)', default='./simulation_results/')
    parser.add_argument('--pretrained_agents', help='any non-empty value: warm start from ./pretrained_weights.npy and ./desired_state.npy', default=False)
    parser.add_argument('--random_seed', help='seed of the NumPy stream, the weight init and the shuffles', type=int, default=300)
    parser.add_argument('--nrow', type=int, default=5)
    parser.add_argument('--ncol', type=int, default=5)
    parser.add_argument('--rng_mode', choices=['numpy', 'device'], default='numpy')
    parser.add_argument('--verbose', action='store_true')
    return parser


def build_models(args):
    """Flatten -> Dense(20, LeakyReLU 0.1) x2 -> head, three per agent (main.py:59-82)."""
    n, ns, na = args['n_agents'], args['n_states'], args['n_actions']

    def mlp(width, out, act):
        return keras.Sequential([keras.Input(shape=(n, width)), keras.layers.Flatten(),
                                 keras.layers.Dense(20, activation=keras.layers.LeakyReLU(alpha=0.1)),
                                 keras.layers.Dense(20, activation=keras.layers.LeakyReLU(alpha=0.1)),
                                 keras.layers.Dense(out, activation=act)])
    return mlp(ns, na, 'softmax'), mlp(ns, 1, None), mlp(ns + 1, 1, None)


def build_agents(args, pretrained_weights=None):
    agents = []
    for node in range(args['n_agents']):
        actor, critic, team_reward = build_models(args)
        if pretrained_weights is not None:
            actor.set_weights(pretrained_weights[node][0])
            critic.set_weights(pretrained_weights[node][1])
            team_reward.set_weights(pretrained_weights[node][2])
        label = args['agent_label'][node]
        if label == 'Malicious':
            print("This is a malicious agent")
            agents.append(Malicious_CAC_agent(actor, critic, team_reward, slow_lr=args['slow_lr'], fast_lr=args['fast_lr'],
                                              gamma=args['gamma']))
            if pretrained_weights is not None:
                agents[node].critic_local_weights = pretrained_weights[node][3]
        elif label == 'Faulty':
            print("This is a faulty agent")
            agents.append(Faulty_CAC_agent(actor, critic, team_reward, slow_lr=args['slow_lr'], gamma=args['gamma']))
        elif label == 'Greedy':
            print("This is a greedy agent")
            agents.append(Greedy_CAC_agent(actor, critic, team_reward, slow_lr=args['slow_lr'], fast_lr=args['fast_lr'],
                                           gamma=args['gamma']))
        else:
            print("This is an RPBCAC agent")
            agents.append(RPBCAC_agent(actor, critic, team_reward, slow_lr=args['slow_lr'], fast_lr=args['fast_lr'],
                                       gamma=args['gamma'], H=args['H']))
    return agents


def save_weights(path, agent_weights):
    """[agent][net][array] as an object array (ragged when a Malicious agent carries a 4th net, main.py:120)."""
    obj = np.empty(len(agent_weights), dtype=object)
    for i, w in enumerate(agent_weights):
        obj[i] = w
    np.save(path, obj, allow_pickle=True)


def main(argv=None, engine_hook=None):
    """engine_hook: (lib, device) handed to train_RPBCAC -- tests inject the hipemu build; None = the product library on cuda."""
    args = vars(build_parser().parse_args(argv))
    np.random.seed(args['random_seed'])
    keras.set_seed(args['random_seed'])
    set_shuffle_seed(args['random_seed'])
    s_desired = np.random.randint(0, 5, size=(args['n_agents'], args['n_states']))     # main.py:48 (5 regardless of grid)
    s_initial = np.random.randint(0, 5, size=(args['n_agents'], args['n_states']))
    pretrained_weights = None
    if args['pretrained_agents']:
        pretrained_weights = np.load('pretrained_weights.npy', allow_pickle=True)
        s_desired = np.load('desired_state.npy', allow_pickle=True)
    agents = build_agents(args, pretrained_weights)
    print(args, s_desired)
    env = Grid_World(nrow=args['nrow'], ncol=args['ncol'], n_agents=args['n_agents'], desired_state=s_desired,
                     initial_state=s_initial, randomize_state=True, scaling=True)
    agent_weights, sim_data = training.train_RPBCAC(env, agents, args, engine_hook=engine_hook)
    sim_data.to_pickle("sim_data.pkl")
    save_weights('pretrained_weights.npy', agent_weights)
    np.save('desired_state.npy', s_desired, allow_pickle=True)
    return agent_weights, sim_data


if __name__ == '__main__':
    main()
