"""Drop-in for the reference's ``environments/grid_world.py`` (host-side object).

Same constructor, attributes and ``reset / step / get_data / close`` as
``Grid_World`` (environments/grid_world.py:19-72), vectorised over agents in
NumPy.  ``train_RPBCAC`` reads the scenario from this object (grid size, goals,
scaling, randomisation) and runs the transitions on the GPU
(csrc/rollout.hip); the methods here serve code that steps the environment by
hand.  Behaviour kept from the reference: the
reward uses the L1 distance to the goal BEFORE the move; the "moved to a free
cell" branch (:59-60) can never fire because the distance to the nearest agent
includes the agent itself; both coordinates are clipped to [0, nrow-1];
``reset`` draws from NumPy's global legacy stream.
"""
import numpy as np

_MOVES = np.array([[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]], dtype=np.int64)


class Grid_World():
    metadata = {'render.modes': ['console']}

    def __init__(self, nrow=5, ncol=5, n_agents=1, desired_state=None, initial_state=None, randomize_state=True,
                 scaling=False):
        self.nrow = nrow
        self.ncol = ncol
        self.n_agents = n_agents
        self.initial_state = initial_state
        self.desired_state = desired_state
        self.randomize_state = randomize_state
        self.n_states = 2
        self.scaling = bool(scaling)
        self.actions_dict = {k: _MOVES[k].copy() for k in range(5)}
        self.reset()                                         # one draw from the global stream, like the reference (:28)
        if scaling:
            x, y = np.arange(nrow), np.arange(ncol)
            self.mean_state = np.array([np.mean(x), np.mean(y)])
            self.std_state = np.array([np.std(x), np.std(y)])
        else:
            self.mean_state, self.std_state = 0, 1

    def reset(self):
        if self.randomize_state:
            self.state = np.random.randint([0, 0], [self.nrow, self.ncol], size=(self.n_agents, self.n_states))
        else:
            self.state = np.array(self.initial_state)
        self.reward = np.zeros(self.n_agents)
        return self.state

    def step(self, action):
        a = np.asarray(action).astype(np.int64).reshape(self.n_agents)
        goal = np.asarray(self.desired_state)
        dist = np.abs(self.state - goal).sum(axis=1)         # before the move
        self.state = np.clip(self.state + _MOVES[a], 0, self.nrow - 1)
        self.reward = np.where((dist == 0) & (a == 0), 0.0, -dist - 1.0).astype(np.float64)

    def get_data(self):
        state_scaled = (self.state - self.mean_state) / self.std_state
        reward_scaled = self.reward / 5
        return state_scaled, reward_scaled

    def close(self):
        pass
