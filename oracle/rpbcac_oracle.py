"""CPU restatement of the RPBCAC training hot path (TEST INFRASTRUCTURE).

Follows, function by function, the reference at /root/reference (cited as
file:line) but shares no code with it: networks are explicit NumPy parameter
lists (oracle/mlp_np.py) instead of Keras models.  The loop nest is kept as in
the reference (per agent -> per neighbour -> per layer) because this file is
also the timed CPU baseline of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
import time
import numpy as np
import pandas as pd

from . import mlp_np as M
from . import philox_np as PX

F32 = np.float32


# ----------------------------------------------------------------------------
# A1  resilient aggregation          agents/resilient_CAC_agents.py:42-58
# ----------------------------------------------------------------------------
def aggregation_bounds(values, H):
    """values: [d, ...] fp32, row 0 = the agent's own value.  Returns the clip
    window (lower, upper): lower = min(sorted[H], own), upper =
    max(sorted[d-H-1], own)        (:48-53)."""
    v = np.asarray(values, dtype=F32)
    d = v.shape[0]
    s = np.sort(v, axis=0)
    lower = np.minimum(s[H], v[0])
    upper = np.maximum(s[d - H - 1], v[0])
    return lower, upper, s


def resilient_aggregate(values, H):
    """Clip every one of the d sorted values into [lower, upper] and average
    over ALL d of them (winsorised mean, nothing is dropped)     (:54-56)."""
    lower, upper, s = aggregation_bounds(values, H)
    clipped = np.maximum(np.minimum(s, upper), lower)
    return np.mean(clipped, axis=0, dtype=F32)


# ----------------------------------------------------------------------------
# E1  grid world                      environments/grid_world.py:19-72
# ----------------------------------------------------------------------------
_MOVES = np.array([[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]], dtype=np.int64)


class GridWorldOracle:
    def __init__(self, nrow=5, ncol=5, n_agents=1, desired_state=None, initial_state=None,
                 randomize_state=True, scaling=False, rng_mode="numpy", seed=0):
        self.nrow, self.ncol, self.n_agents = nrow, ncol, n_agents
        self.desired_state = None if desired_state is None else np.asarray(desired_state, dtype=np.int64)
        self.initial_state = initial_state
        self.randomize_state = randomize_state
        self.rng_mode, self.seed, self.episode = rng_mode, seed, 0
        if scaling:                                                     # :29-33
            self.mean_state = np.array([np.mean(np.arange(nrow)), np.mean(np.arange(ncol))])
            self.std_state = np.array([np.std(np.arange(nrow)), np.std(np.arange(ncol))])
        else:
            self.mean_state, self.std_state = 0, 1
        if rng_mode == "numpy":
            self.reset()                       # the reference's ctor draws once (:28)

    def reset(self, episode=None):
        if self.randomize_state:                                        # :39-40
            if self.rng_mode == "numpy":
                self.state = np.random.randint([0, 0], [self.nrow, self.ncol], size=(self.n_agents, 2))
            else:
                ep = self.episode if episode is None else episode
                self.state = PX.reset_positions(self.n_agents, self.nrow, self.ncol, self.seed, ep)
        else:
            self.state = np.array(self.initial_state)
        self.reward = np.zeros(self.n_agents)
        return self.state

    def step(self, action):
        """Sequential per-agent update (:52-64).  dist_to_agents includes the
        agent itself, hence is always 0 and the first branch never fires; kept
        literally so a behavioural change in the rule would surface."""
        for i in range(self.n_agents):
            a = int(action[i])
            d_before = int(np.abs(self.state[i] - self.desired_state[i]).sum())
            self.state[i] = np.clip(self.state[i] + _MOVES[a], 0, self.nrow - 1)   # nrow-1 on both axes
            d_agents = int(np.abs(self.state - self.state[i]).sum(axis=1).min())
            d_after = int(np.abs(self.state[i] - self.desired_state[i]).sum())
            if d_agents > 0:
                self.reward[i] = -d_after
            elif d_before == 0 and a == 0:
                self.reward[i] = 0
            else:
                self.reward[i] = -d_before - 1

    def get_data(self):                                                 # :66-72
        return (self.state - self.mean_state) / self.std_state, self.reward / 5


# ----------------------------------------------------------------------------
# deterministic shuffles for the mini-batch adversaries
# ----------------------------------------------------------------------------
class ShuffleStream:
    """Keras' fit() shuffles with TensorFlow's RNG, which cannot be reproduced here; the shuffle is
    DEFINED (identically in csrc/shuffle.hip): epoch e of the n-th mini-batch fit of a run visits the
    rows in the order that sorts  key(p) = Philox4x32-10(counter=(p, e, n, stream 2), key=seed).word0,
    ties broken by the lower row index.  Counter-based, so the HIP engine produces all permutations of
    all seeds in one launch; it draws from the identical stream, call for call."""

    def __init__(self, seed):
        self.seed, self.calls = int(seed), 0

    def perms(self, epochs, B):
        from . import philox_np as PH
        k0, k1 = PH.seed_key(self.seed)
        rows = np.arange(B, dtype=np.uint64)
        out = []
        for e in range(epochs):
            r0 = PH.philox4x32(rows, e, self.calls, 2, k0, k1)[0].astype(np.uint64)
            out.append(np.argsort((r0 << np.uint64(32)) | rows, kind="stable"))
        self.calls += 1
        return np.stack(out).astype(np.int32)


# ----------------------------------------------------------------------------
# agents
# ----------------------------------------------------------------------------
def _flat(x):
    x = np.asarray(x, dtype=F32)
    return x.reshape(x.shape[0], -1)


class _AgentBase:
    label = "?"

    def __init__(self, actor, critic, tr, slow_lr, gamma):
        self.actor, self.critic, self.tr = M.copy_params(actor), M.copy_params(critic), M.copy_params(tr)
        self.gamma = F32(gamma)
        self.n_actions = self.actor[4].shape[1]
        self.adam = M.AdamState(self.actor, slow_lr)

    # A8 get_action                       agents/resilient_CAC_agents.py:208-219
    def policy(self, state_row):
        return M.softmax(M.forward(self.actor, _flat(state_row)))[0]

    def act_numpy(self, state_row, mu=0.1):
        a_rand = np.random.choice(self.n_actions)
        p = self.policy(state_row)
        a_pol = np.random.choice(self.n_actions, p=p)
        return np.random.choice([a_pol, a_rand], p=[1 - mu, mu])

    def parameters(self):                                               # :221-223
        return [M.copy_params(self.actor), M.copy_params(self.critic), M.copy_params(self.tr)]

    def _own_td_actor_fit(self, critic, s, ns, r_local, a_local, shuffle):
        """Adversaries' actor step: TD error of their own critic, then
        fit(batch_size=200, epochs=1)      adversarial_CAC_agents.py:38-41,111-117,221-225."""
        s, ns = _flat(s), _flat(ns)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        td = r + self.gamma * M.forward(critic, ns) - M.forward(critic, s)
        perms = shuffle.perms(1, s.shape[0]) if s.shape[0] > 200 else None
        return M.fit_actor_ce(self.actor, self.adam, s, a_local, td, epochs=1, batch_size=200, perms=perms)[0]


class CoopAgent(_AgentBase):
    """RPBCAC_agent          agents/resilient_CAC_agents.py:5-223."""
    label = "Cooperative"

    def __init__(self, actor, critic, tr, slow_lr, fast_lr, gamma=0.95, H=0):
        super().__init__(actor, critic, tr, slow_lr, gamma)
        self.fast_lr, self.H = float(fast_lr), int(H)

    # A5 / A6 local fits with rollback                                   :103-140
    def local_fit_critic(self, s, ns, r_local):
        s, ns = _flat(s), _flat(ns)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        target = r + self.gamma * M.forward(self.critic, ns)            # fixed target :114-115
        msg = M.copy_params(self.critic)
        hist = M.fit_mse(msg, s, target, self.fast_lr, epochs=5)        # :118
        return msg, hist[0]                                             # live net untouched = rollback :120

    def local_fit_tr(self, sa, r_local):
        sa = _flat(sa)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        msg = M.copy_params(self.tr)
        hist = M.fit_mse(msg, sa, r, self.fast_lr, epochs=5)            # :136
        return msg, hist[0]

    # A2 hidden-layer consensus                                           :142-166
    def _consensus_hidden(self, net, msgs):
        agg = []
        for layer in zip(*msgs):                                        # per layer, over neighbours
            agg.append(resilient_aggregate(np.stack(layer), self.H))
        for k in range(4):                                              # only W1,b1,W2,b2 are applied ([:-2])
            net[k][...] = agg[k]

    def consensus_hidden_critic(self, msgs):
        self._consensus_hidden(self.critic, msgs)

    def consensus_hidden_tr(self, msgs):
        self._consensus_hidden(self.tr, msgs)

    # A3 consensus over estimates                                         :168-206
    def _consensus_estimates(self, net, x, msgs):
        x = _flat(x)
        est = []
        for msg in msgs:                                                # own *message* head first
            probe = net[:4] + [msg[4], msg[5]]                          # own hidden, neighbour's head
            est.append(M.forward(probe, x))
        return resilient_aggregate(np.stack(est), self.H)               # [B,1]

    def consensus_estimates_critic(self, s, msgs):
        return self._consensus_estimates(self.critic, s, msgs)

    def consensus_estimates_tr(self, sa, msgs):
        return self._consensus_estimates(self.tr, sa, msgs)

    # A4 projection ("team") update of the output layer                   :60-84
    def _projection_step(self, net, x, agg):
        x = _flat(x)
        phi = M.features(net, x)
        phi_norm = np.sum(np.square(phi), axis=1) + 1
        w = 1 / (2 * self.fast_lr * phi_norm)                           # :67-68
        pred, cache = M.forward(net, x, want_cache=True)
        _, dout = M.mse_loss_and_dout(pred, agg, sample_weight=w)
        grads = M.backward(net, cache, dout, hidden_trainable=False)    # hidden frozen :69
        M.sgd_apply(net, grads, self.fast_lr)

    def projection_step_critic(self, s, agg):
        self._projection_step(self.critic, s, agg)

    def projection_step_tr(self, sa, agg):
        self._projection_step(self.tr, sa, agg)

    # A7 actor update                                                     :86-101
    def actor_step(self, s, ns, sa, a_local):
        s, ns, sa = _flat(s), _flat(ns), _flat(sa)
        td = M.forward(self.tr, sa) + self.gamma * M.forward(self.critic, ns) - M.forward(self.critic, s)
        return M.fit_actor_ce(self.actor, self.adam, s, a_local, td, epochs=1)[0]   # train_on_batch


class FaultyAgent(_AgentBase):
    """Faulty_CAC_agent      agents/adversarial_CAC_agents.py:5-73: frozen
    critic/TR are transmitted; only the actor learns."""
    label = "Faulty"

    def __init__(self, actor, critic, tr, slow_lr, gamma=0.95):
        super().__init__(actor, critic, tr, slow_lr, gamma)

    def actor_step(self, s, ns, r_local, a_local, shuffle):
        return self._own_td_actor_fit(self.critic, s, ns, r_local, a_local, shuffle)


class GreedyAgent(_AgentBase):
    """Greedy_CAC_agent      agents/adversarial_CAC_agents.py:184-275: fits
    critic/TR on its own reward with mini-batches, no rollback."""
    label = "Greedy"

    def __init__(self, actor, critic, tr, slow_lr, fast_lr, gamma=0.95):
        super().__init__(actor, critic, tr, slow_lr, gamma)
        self.fast_lr = float(fast_lr)

    def local_fit_critic(self, s, ns, r_local, shuffle):                # :228-241
        s, ns = _flat(s), _flat(ns)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        target = r + self.gamma * M.forward(self.critic, ns)
        hist = M.fit_mse(self.critic, s, target, self.fast_lr, epochs=10, batch_size=32,
                         perms=shuffle.perms(10, s.shape[0]))
        return M.copy_params(self.critic), hist[0]

    def local_fit_tr(self, sa, r_local, shuffle):                       # :243-253
        sa = _flat(sa)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        hist = M.fit_mse(self.tr, sa, r, self.fast_lr, epochs=10, batch_size=32,
                         perms=shuffle.perms(10, sa.shape[0]))
        return M.copy_params(self.tr), hist[0]

    def actor_step(self, s, ns, r_local, a_local, shuffle):
        return self._own_td_actor_fit(self.critic, s, ns, r_local, a_local, shuffle)


class MaliciousAgent(_AgentBase):
    """Malicious_CAC_agent   agents/adversarial_CAC_agents.py:75-182: a private
    critic for its own actor, and transmitted ("compromised") critic/TR trained
    on whatever reward the trainer hands it (-r_coop, train_agents.py:113-116)."""
    label = "Malicious"

    def __init__(self, actor, critic, tr, slow_lr, fast_lr, gamma=0.95):
        super().__init__(actor, critic, tr, slow_lr, gamma)
        self.fast_lr = float(fast_lr)
        self.critic_local = M.copy_params(self.critic)                  # :101

    def local_fit_private_critic(self, s, ns, r_local, shuffle):        # :137-152
        s, ns = _flat(s), _flat(ns)
        r = np.asarray(r_local, dtype=F32).reshape(-1, 1)
        target = r + self.gamma * M.forward(self.critic_local, ns)
        M.fit_mse(self.critic_local, s, target, self.fast_lr, epochs=10, batch_size=32,
                  perms=shuffle.perms(10, s.shape[0]))

    def fit_compromised_critic(self, s, ns, r_comp, shuffle):           # :121-135
        s, ns = _flat(s), _flat(ns)
        r = np.asarray(r_comp, dtype=F32).reshape(-1, 1)
        target = r + self.gamma * M.forward(self.critic, ns)
        hist = M.fit_mse(self.critic, s, target, self.fast_lr, epochs=10, batch_size=32,
                         perms=shuffle.perms(10, s.shape[0]))
        return M.copy_params(self.critic), hist[0]

    def fit_compromised_tr(self, sa, r_comp, shuffle):                  # :154-165
        sa = _flat(sa)
        r = np.asarray(r_comp, dtype=F32).reshape(-1, 1)
        hist = M.fit_mse(self.tr, sa, r, self.fast_lr, epochs=10, batch_size=32,
                         perms=shuffle.perms(10, sa.shape[0]))
        return M.copy_params(self.tr), hist[0]

    def actor_step(self, s, ns, r_local, a_local, shuffle):             # :103-119
        return self._own_td_actor_fit(self.critic_local, s, ns, r_local, a_local, shuffle)

    def parameters(self):                                               # :180-182
        return super().parameters() + [M.copy_params(self.critic_local)]


def make_agent(label, actor, critic, tr, slow_lr, fast_lr, gamma, H):
    """Dispatch by label as main.py:88-104 does."""
    if label == "Malicious":
        return MaliciousAgent(actor, critic, tr, slow_lr, fast_lr, gamma)
    if label == "Faulty":
        return FaultyAgent(actor, critic, tr, slow_lr, gamma)
    if label == "Greedy":
        return GreedyAgent(actor, critic, tr, slow_lr, fast_lr, gamma)
    return CoopAgent(actor, critic, tr, slow_lr, fast_lr, gamma, H)


# ----------------------------------------------------------------------------
# T1/T2 training loop                 training/train_agents.py:15-184
# ----------------------------------------------------------------------------
def update_block(agents, labels, in_nodes, s, ns, r, a, n_epochs, common_reward, max_ep_len, n_ep_fixed,
                 shuffle, timers=None):
    """One update block (train_agents.py:86-153) on the replay tensors
    s,ns [B,N,2]  r,a [B,N,1]  (fp32).  Returns (actor, critic, TR) loss arrays."""
    n_agents = len(agents)
    coop = [i for i in range(n_agents) if labels[i] == "Cooperative"]
    n_coop = len(coop)
    sa = np.concatenate([s, a], axis=-1)                                # :93
    r_coop = np.zeros((r.shape[0], r.shape[2]), F32)                    # :96-98
    for i in coop:
        r_coop += r[:, i] / n_coop
    actor_loss, critic_loss, tr_loss = np.zeros(n_agents), np.zeros(n_agents), np.zeros(n_agents)
    tm = timers if timers is not None else {}
    for _ in range(n_epochs):                                           # :100
        t0 = time.perf_counter()
        critic_msgs, tr_msgs = [], []
        for i in range(n_agents):                                       # phase I :105-121
            ag = agents[i]
            if labels[i] == "Cooperative":
                r_applied = r_coop if common_reward else r[:, i]
                x, tr_loss[i] = ag.local_fit_tr(sa, r_applied)
                y, critic_loss[i] = ag.local_fit_critic(s, ns, r_applied)
            elif labels[i] == "Greedy":
                x, tr_loss[i] = ag.local_fit_tr(sa, r[:, i], shuffle)
                y, critic_loss[i] = ag.local_fit_critic(s, ns, r[:, i], shuffle)
            elif labels[i] == "Malicious":
                ag.local_fit_private_critic(s, ns, r[:, i], shuffle)
                x, tr_loss[i] = ag.fit_compromised_tr(sa, -r_coop, shuffle)
                y, critic_loss[i] = ag.fit_compromised_critic(s, ns, -r_coop, shuffle)
            elif labels[i] == "Faulty":
                x, y = M.copy_params(ag.tr), M.copy_params(ag.critic)
            tr_msgs.append(x)
            critic_msgs.append(y)
        t1 = time.perf_counter()
        for i in coop:                                                  # phase II :125-145
            ag = agents[i]
            c_in = [critic_msgs[j] for j in in_nodes[i]]
            t_in = [tr_msgs[j] for j in in_nodes[i]]
            ag.consensus_hidden_critic(c_in)
            ag.consensus_hidden_tr(t_in)
            c_agg = ag.consensus_estimates_critic(s, c_in)
            t_agg = ag.consensus_estimates_tr(sa, t_in)
            ag.projection_step_critic(s, c_agg)
            ag.projection_step_tr(sa, t_agg)
        t2 = time.perf_counter()
        tm["phase1"] = tm.get("phase1", 0.0) + (t1 - t0)
        tm["phase2"] = tm.get("phase2", 0.0) + (t2 - t1)
    t0 = time.perf_counter()
    n_last = max_ep_len * n_ep_fixed                                    # phase III :149-153
    for i in range(n_agents):
        if labels[i] == "Cooperative":
            actor_loss[i] = agents[i].actor_step(s[-n_last:], ns[-n_last:], sa[-n_last:], a[-n_last:, i])
        else:
            actor_loss[i] = agents[i].actor_step(s[-n_last:], ns[-n_last:], r[-n_last:, i], a[-n_last:, i], shuffle)
    tm["phase3"] = tm.get("phase3", 0.0) + (time.perf_counter() - t0)
    return actor_loss, critic_loss, tr_loss


def train(env, agents, args, exp_buffer=None, rng_mode="numpy", shuffle=None, timers=None, verbose=False):
    """Restatement of train_RPBCAC (training/train_agents.py:15-184).
    rng_mode 'numpy' = the reference's global legacy stream; 'device' = the
    engine's Philox stream (oracle/philox_np.py).  Returns (weights, DataFrame)."""
    labels = args["agent_label"]
    n_agents = env.n_agents
    n_coop = labels.count("Cooperative")
    gamma, in_nodes = args["gamma"], args["in_nodes"]
    max_ep_len, n_episodes, n_ep_fixed = args["max_ep_len"], args["n_episodes"], args["n_ep_fixed"]
    n_epochs, buffer_size = args["n_epochs"], args["buffer_size"]
    seed = args.get("random_seed", 0)
    shuffle = shuffle if shuffle is not None else ShuffleStream(seed)
    tm = timers if timers is not None else {}
    if exp_buffer:
        states, nstates, actions, rewards = exp_buffer[0], exp_buffer[1], exp_buffer[2], exp_buffer[3]
    else:
        states, nstates, actions, rewards = [], [], [], []
    paths = []
    for t in range(n_episodes):
        t_roll = time.perf_counter()
        j, ep_returns = 0, 0
        est_returns, mean_ret, mean_ret_adv = [], 0, 0
        action = np.zeros(n_agents)
        i_fixed = t % n_ep_fixed
        env.reset() if rng_mode == "numpy" else env.reset(episode=t)
        state, _ = env.get_data()
        for i in range(n_agents):                                       # :60-62
            if labels[i] == "Cooperative":
                est_returns.append(M.forward(agents[i].critic, _flat(state[None]))[0][0])
        while j < max_ep_len:
            if rng_mode == "numpy":
                for i in range(n_agents):                               # :67-68
                    action[i] = agents[i].act_numpy(state[None])
            else:
                probs = np.stack([agents[i].policy(state[None]) for i in range(n_agents)])
                action[:] = PX.sample_actions(probs, seed, t, j)
            env.step(action)
            nstate, reward = env.get_data()
            ep_returns = ep_returns + reward * (gamma ** j)             # :71 (float64)
            j += 1
            states.append(np.array(state))
            nstates.append(np.array(nstate))
            actions.append(np.array(action).reshape(-1, 1))
            rewards.append(np.array(reward).reshape(-1, 1))
            state = np.array(nstate)
            if i_fixed == n_ep_fixed - 1 and j == max_ep_len:           # :86
                tm["rollout"] = tm.get("rollout", 0.0) + (time.perf_counter() - t_roll)
                s = np.asarray(states, dtype=F32)
                ns = np.asarray(nstates, dtype=F32)
                r = np.asarray(rewards, dtype=F32)
                a = np.asarray(actions, dtype=F32)
                update_block(agents, labels, in_nodes, s, ns, r, a, n_epochs, args["common_reward"],
                             max_ep_len, n_ep_fixed, shuffle, tm)
                if len(states) > buffer_size:                           # :158-163
                    q = len(states) - buffer_size
                    del states[:q], nstates[:q], actions[:q], rewards[:q]
                t_roll = time.perf_counter()
        tm["rollout"] = tm.get("rollout", 0.0) + (time.perf_counter() - t_roll)
        for i in range(n_agents):                                       # :168-172
            if labels[i] == "Cooperative":
                mean_ret += ep_returns[i] / n_coop
            else:
                mean_ret_adv += ep_returns[i] / (n_agents - n_coop)
        if verbose:
            print("| Episode: {} | Est. returns: {} | Returns: {}".format(t, est_returns, mean_ret))
        paths.append({"True_team_returns": mean_ret, "True_adv_returns": mean_ret_adv,
                      "Estimated_team_returns": np.mean(est_returns)})
    return [ag.parameters() for ag in agents], pd.DataFrame.from_dict(paths)
