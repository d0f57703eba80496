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


class Transition(object):
    """core_types.py:236-340 — the object agents written against the reference store and receive
    (coach_amd/memories/reference_api.py); the device agents move columns instead and never build one."""

    def __init__(self, state=None, action=None, reward=None, next_state=None, game_over=None, info=None, goal=None):
        self.state = state if state is not None else {}
        self.action = action
        self.reward = reward
        self.n_step_discounted_rewards = None
        self.next_state = next_state if next_state is not None else {}
        self.game_over = game_over
        self.goal = goal
        self.info = info if info is not None else {}

    def add_info(self, new_info):
        if not new_info.keys().isdisjoint(self.info.keys()):
            raise ValueError("The new info dictionary can not be appended to the existing info dictionary since there "
                             "are overlapping keys between the two. old keys: {}, new keys: {}"
                             .format(self.info.keys(), new_info.keys()))
        self.info.update(new_info)

    def update_info(self, new_info):
        self.info.update(new_info)


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
