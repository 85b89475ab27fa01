"""Drop-in for the reference's ``agents/adversarial_CAC_agents.py``: the three
message generators the resilient consensus must withstand.

Same constructors and methods as the reference (Faulty :19-73, Malicious :90-182,
Greedy :200-275).  Keras' shuffled mini-batch ``fit`` calls are single kernel
launches (csrc/minibatch_fit.hip).  The shuffle permutations come from a
process-wide stream (``set_shuffle_seed``): epoch e of the n-th mini-batch fit of a run visits
the rows in the order that sorts Philox(counter=(row, e, n, 2), key=seed) (csrc/shuffle.hip) -- a
definition, shared with the oracle, since TensorFlow's own shuffle RNG cannot be reproduced
outside TensorFlow.
"""
import numpy as np

from .. import single

_shuffle = {"seed": 0, "calls": 0}


def set_shuffle_seed(seed):
    _shuffle["seed"], _shuffle["calls"] = int(seed), 0


def _perms(epochs, B):
    p = single.get_ops().shuffle_perms(_shuffle["seed"], _shuffle["calls"], epochs, B)
    _shuffle["calls"] += 1
    return p


def _nrows(x):
    return int(np.asarray(x).shape[0])


class _Adversary():
    def _init_common(self, actor, critic, team_reward, slow_lr, gamma):
        self.actor = actor
        self.critic = critic
        self.TR = team_reward
        self.gamma = gamma
        self.slow_lr = slow_lr
        self.n_actions = self.actor.output_shape[1]
        n = single.flat(self.actor.get_weights()).size
        self._adam = {"m": np.zeros(n, np.float32), "v": np.zeros(n, np.float32), "t": 0}

    def _actor_fit(self, critic_weights, s, ns, r_local, a_local):
        """TD error of the agent's own critic, then actor.fit(batch_size=200, epochs=1) (:38-41)."""
        ops = single.get_ops()
        B = _nrows(s)
        V, nV = ops.value(critic_weights, s), ops.value(critic_weights, ns)
        TD_error = np.asarray(r_local, np.float32).reshape(B, 1) + np.float32(self.gamma) * nV - V
        perms = _perms(1, B) if B > 200 else None
        new, loss = ops.minibatch_actor(self.actor.get_weights(), self._adam, s, a_local, TD_error, self.slow_lr,
                                        batch_size=200, epochs=1, perms=perms)
        self.actor.set_weights(new)
        return loss

    def get_action(self, state, mu=0.1):
        random_action = np.random.choice(self.n_actions)
        action_prob = self.actor.predict(state).ravel()
        action_from_policy = np.random.choice(self.n_actions, p=action_prob)
        self.action = np.random.choice([action_from_policy, random_action], p=[1 - mu, mu])
        return self.action

    def get_parameters(self):
        return [self.actor.get_weights(), self.critic.get_weights(), self.TR.get_weights()]

    def _fit32(self, model, x, y):
        """model.fit(x, y, epochs=10, batch_size=32) in place; returns (weights, first-epoch loss)."""
        B = _nrows(x)
        new, loss = single.get_ops().minibatch_fit(model.get_weights(), x, y, self.fast_lr, batch_size=32, epochs=10,
                                                   perms=_perms(10, B))
        model.set_weights(new)
        return new, loss


class Faulty_CAC_agent(_Adversary):
    """Transmits frozen critic / TR parameters; only its actor learns (:5-73)."""

    def __init__(self, actor, critic, team_reward, slow_lr, gamma=0.95):
        self._init_common(actor, critic, team_reward, slow_lr, gamma)

    def actor_update(self, s, ns, r_local, a_local):
        return self._actor_fit(self.critic.get_weights(), s, ns, r_local, a_local)

    def get_critic_weights(self):
        return self.critic.get_weights()

    def get_TR_weights(self):
        return self.TR.get_weights()


class Malicious_CAC_agent(_Adversary):
    """Private critic for its own actor; transmits a critic / TR trained on the reward the trainer
    hands it (-r_coop, training/train_agents.py:113-116) (:75-182)."""

    def __init__(self, actor, critic, team_reward, slow_lr, fast_lr, gamma=0.95):
        self._init_common(actor, critic, team_reward, slow_lr, gamma)
        self.fast_lr = fast_lr
        self.critic_local_weights = self.critic.get_weights()

    def actor_update(self, s, ns, r_local, a_local):
        return self._actor_fit(self.critic_local_weights, s, ns, r_local, a_local)

    def critic_update_compromised(self, s, ns, r_compromised):
        B = _nrows(s)
        nV = single.get_ops().value(self.critic.get_weights(), ns)
        target = np.asarray(r_compromised, np.float32).reshape(B, 1) + np.float32(self.gamma) * nV
        return self._fit32(self.critic, s, target)

    def critic_update_local(self, s, ns, r_local):
        B = _nrows(s)
        ops = single.get_ops()
        nV = ops.value(self.critic_local_weights, ns)
        target = np.asarray(r_local, np.float32).reshape(B, 1) + np.float32(self.gamma) * nV
        new, _ = ops.minibatch_fit(self.critic_local_weights, s, target, self.fast_lr, batch_size=32, epochs=10,
                                   perms=_perms(10, B))
        self.critic_local_weights = new

    def TR_update_compromised(self, sa, r_compromised):
        return self._fit32(self.TR, sa, r_compromised)

    def get_parameters(self):
        return [self.actor.get_weights(), self.critic.get_weights(), self.TR.get_weights(), self.critic_local_weights]


class Greedy_CAC_agent(_Adversary):
    """Fits critic / TR on its own reward with mini-batches, no rollback, and transmits them (:184-275)."""

    def __init__(self, actor, critic, team_reward, slow_lr, fast_lr, gamma=0.95):
        self._init_common(actor, critic, team_reward, slow_lr, gamma)
        self.fast_lr = fast_lr

    def actor_update(self, s, ns, r_local, a_local):
        return self._actor_fit(self.critic.get_weights(), s, ns, r_local, a_local)

    def critic_update_local(self, s, ns, r_local):
        B = _nrows(s)
        nV = single.get_ops().value(self.critic.get_weights(), ns)
        target = np.asarray(r_local, np.float32).reshape(B, 1) + np.float32(self.gamma) * nV
        return self._fit32(self.critic, s, target)

    def TR_update_local(self, sa, r_local):
        return self._fit32(self.TR, sa, r_local)
