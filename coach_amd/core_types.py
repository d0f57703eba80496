"""Step units, run phases and the device-resident Batch — host-side mirror of the parts of
rl_coach/core_types.py the hot path touches (StepMethod :55-127, RunPhase :186-190, Batch :405-649).
"""
import math
from enum import Enum

import numpy as np


class StepMethod(object):                                # core_types.py:55-101
    def __init__(self, num_steps):
        self.num_steps = num_steps

    def __eq__(self, other):
        return self.num_steps == other.num_steps

    def __truediv__(self, other):
        if isinstance(other, type(self)):
            return math.ceil(self.num_steps / other.num_steps)
        elif isinstance(other, int):
            return type(self)(math.ceil(self.num_steps / other))
        raise TypeError("cannot divide {} by {}".format(type(self), type(other)))

    def __repr__(self):
        return "{}({})".format(type(self).__name__, self.num_steps)


class EnvironmentSteps(StepMethod):
    pass


class EnvironmentEpisodes(StepMethod):
    pass


class TrainingSteps(StepMethod):
    pass


class RunPhase(Enum):                                    # core_types.py:186-190
    HEATUP = "Heatup"
    TRAIN = "Training"
    TEST = "Testing"
    UNDEFINED = "Undefined"


class DeviceBatch(object):
    """Batch (core_types.py:405-649) whose columns are device tensors gathered by one launch.
    Accessors keep the reference's names; each returns the device tensor, `.numpy(name)` copies a
    column to the host for API parity (Batch.states()[k] etc. return ndarrays in the reference)."""

    def __init__(self, size, states, next_states=None, actions=None, rewards=None, game_overs=None,
                 info=None):
        self.size = size
        self._states, self._next_states = states, next_states or {}
        self._actions, self._rewards, self._game_overs = actions, rewards, game_overs
        self._info = info or {}

    def states(self, fetches=('observation',)):
        return {k: self._states[k] for k in fetches}

    def next_states(self, fetches=('observation',)):
        return {k: self._next_states[k] for k in fetches}

    def actions(self):
        return self._actions

    def rewards(self):
        return self._rewards

    def game_overs(self):
        return self._game_overs

    def info(self, key):
        return self._info[key]

    def numpy(self, tensor):
        return tensor.detach().cpu().numpy()
