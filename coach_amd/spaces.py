"""The space descriptions a network is constructed from — the fields of rl_coach/spaces.py that the
device networks read (observation shape, number of discrete actions / action dimension and bounds)."""
import numpy as np


class Space(object):
    def __init__(self, shape, low=-np.inf, high=np.inf):
        self.shape = np.array([shape]) if np.isscalar(shape) else np.array(shape)
        self.low, self.high = low, high

    @property
    def num_dimensions(self):
        return len(self.shape)


class ObservationSpace(Space):
    """VectorObservationSpace: shape (D,); image observations: (H, W, stack) uint8."""


class DiscreteActionSpace(Space):
    def __init__(self, num_actions, descriptions=None):
        super().__init__(1, 0, num_actions - 1)
        self.actions = list(range(num_actions))
        self.descriptions = descriptions


class BoxActionSpace(Space):
    def __init__(self, shape, low=-1.0, high=1.0):
        super().__init__(shape, low, high)


class StateSpace(object):
    def __init__(self, sub_spaces):
        self.sub_spaces = dict(sub_spaces)

    def __getitem__(self, item):
        return self.sub_spaces[item]


class SpacesDefinition(object):
    def __init__(self, state, goal, action, reward=None):
        self.state, self.goal, self.action, self.reward = state, goal, action, reward
