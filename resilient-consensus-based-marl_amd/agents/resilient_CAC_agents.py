"""Drop-in for the reference's ``agents/resilient_CAC_agents.py``.

``RPBCAC_agent`` keeps the reference's constructor, attributes and method set
(agents/resilient_CAC_agents.py:28-223); every method runs on the MI355X through
the C-ABI kernels (rcmarl_amd.single.RowOps) instead of TensorFlow/Keras:

  critic_update_local / TR_update_local      :103-140   5 full-batch SGD steps on a copy, rollback
  resilient_consensus_{critic,TR}_hidden     :142-166   K1 (select-clip-mean over neighbour messages)
  resilient_consensus_{critic,TR}            :168-206   K2 (neighbour heads on own features)
  critic_update_team / TR_update_team        :60-84     K3 (normalised projection step of the head)
  actor_update                               :86-101    TD error + one Adam step
  get_action                                 :208-219   three draws from NumPy's global legacy stream

The models (``actor``, ``critic``, ``team_reward``) are rcmarl_amd.keras_compat
objects or anything exposing ``get_weights/set_weights/output_shape`` with the
architecture of main.py:59-82.  ``train_RPBCAC`` does NOT call these methods in
a loop: it batches all agents into one engine.  They exist so that code written
against the reference's per-agent API keeps working.
"""
import numpy as np

from .. import single


class RPBCAC_agent():
    def __init__(self, actor, critic, team_reward, slow_lr, fast_lr, gamma=0.95, H=0):
        self.actor = actor
        self.critic = critic
        self.TR = team_reward
        self.gamma = gamma
        self.H = H
        self.n_actions = self.actor.output_shape[1]
        self.fast_lr = fast_lr
        self.slow_lr = slow_lr
        n = single.flat(self.actor.get_weights()).size
        # Adam slots of the actor (keras.optimizers.Adam compiled once, :38): persist across updates
        self._adam = {"m": np.zeros(n, np.float32), "v": np.zeros(n, np.float32), "t": 0}

    # ---- the aggregation rule itself (:42-58) ------------------------------------------------
    def _resilient_aggregation(self, values_innodes):
        """values_innodes: [d, ...], row 0 = own value -> clipped mean over the d rows."""
        v = np.asarray(values_innodes, dtype=np.float32)
        d, shape = v.shape[0], v.shape[1:]
        cols = v.reshape(d, -1)
        ops = single.get_ops()
        ldp = single.pad64(cols.shape[1])
        msg = np.zeros((1, d, ldp), np.float32)
        msg[0, :, :cols.shape[1]] = cols
        msg_d, theta = ops._t(msg), ops._zeros(1, d, ldp)
        nbr = ops._t(np.array([[(i + k) % d for k in range(d)] for i in range(d)], np.int32))
        coop = ops._t(np.array([1] + [0] * (d - 1), np.int32))
        ops.lib.rcmarl_consensus_params(msg_d.data_ptr(), theta.data_ptr(), nbr.data_ptr(), coop.data_ptr(), 1, d, ldp,
                                        cols.shape[1], d, int(self.H), None, None, ops.stream)
        return ops._host(theta)[0, 0, :cols.shape[1]].reshape(shape).copy()

    # ---- A4 -----------------------------------------------------------------------------------
    def critic_update_team(self, s, critic_agg):
        w = self.critic.get_weights()
        W3, b3 = single.get_ops().projection_step(w, s, critic_agg)
        self.critic.set_weights(w[:4] + [W3, b3])

    def TR_update_team(self, sa, TR_agg):
        w = self.TR.get_weights()
        W3, b3 = single.get_ops().projection_step(w, sa, TR_agg)
        self.TR.set_weights(w[:4] + [W3, b3])

    # ---- A7 -----------------------------------------------------------------------------------
    def actor_update(self, s, ns, sa, a_local, pretrain=False):
        ops = single.get_ops()
        r_team = ops.value(self.TR.get_weights(), sa)
        cw = self.critic.get_weights()
        V, nV = ops.value(cw, s), ops.value(cw, ns)
        global_TD_error = r_team + np.float32(self.gamma) * nV - V
        new, loss = ops.actor_step(self.actor.get_weights(), self._adam, s, a_local, global_TD_error, self.slow_lr)
        self.actor.set_weights(new)
        return loss

    # ---- A5 / A6 -------------------------------------------------------------------------------
    def critic_update_local(self, s, ns, r_local):
        msg, loss = single.get_ops().fit_full_batch(self.critic.get_weights(), s, r_local, self.fast_lr, steps=5,
                                                    bootstrap_x=ns, gamma=self.gamma)
        return msg, loss                                  # the live critic is untouched (= the reference's rollback)

    def TR_update_local(self, sa, r_local):
        msg, loss = single.get_ops().fit_full_batch(self.TR.get_weights(), sa, r_local, self.fast_lr, steps=5)
        return msg, loss

    # ---- A2 -----------------------------------------------------------------------------------
    def resilient_consensus_critic_hidden(self, critic_weights_innodes):
        hid = single.get_ops().consensus_hidden(critic_weights_innodes, self.H)
        self.critic.set_weights(hid + self.critic.get_weights()[4:])      # aggregated W3,b3 are discarded (:150-153)

    def resilient_consensus_TR_hidden(self, TR_weights_innodes):
        hid = single.get_ops().consensus_hidden(TR_weights_innodes, self.H)
        self.TR.set_weights(hid + self.TR.get_weights()[4:])

    # ---- A3 -----------------------------------------------------------------------------------
    def resilient_consensus_critic(self, s, critic_weights_innodes):
        return single.get_ops().consensus_estimates(self.critic.get_weights(), s, critic_weights_innodes, self.H)

    def resilient_consensus_TR(self, sa, TR_weights_innodes):
        return single.get_ops().consensus_estimates(self.TR.get_weights(), sa, TR_weights_innodes, self.H)

    # ---- A8 -----------------------------------------------------------------------------------
    def get_action(self, state, mu=0.1):
        random_action = np.random.choice(self.n_actions)
        action_prob = self.actor.predict(state).ravel()
        action_from_policy = np.random.choice(self.n_actions, p=action_prob)
        self.action = np.random.choice([action_from_policy, random_action], p=[1 - mu, mu])
        return self.action

    def get_parameters(self):
        return [self.actor.get_weights(), self.critic.get_weights(), self.TR.get_weights()]
