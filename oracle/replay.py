"""Uniform replay + observation stacking — numpy restatement (pinned: golden er.npz / stack.npz
generated from the reference, plus the reference's own stacking-filter tests).

Follows rl_coach/memories/non_episodic/experience_replay.py:71-150 (FIFO list, np.random.randint
sampling with replacement) and rl_coach/filters/observation/observation_stacking_filter.py:27-41,
89-101 (deque of the last `stack` frames, first frame replicated, np.stack(axis=-1)).
"""
from collections import deque

import numpy as np


class UniformReplayOracle:
    """Index semantics of ExperienceReplay: logical index i == i-th oldest stored transition."""

    def __init__(self, max_size):
        self.max_size = max_size
        self.rows = []                                   # list of tuples of numpy values

    def store(self, row):                                # :131-150
        self.rows.append(row)
        while self.max_size != 0 and len(self.rows) > self.max_size:   # _enforce_max_length :117-129
            del self.rows[0]

    def num_transitions(self):                           # :65-69
        return len(self.rows)

    def sample_indices(self, size, rng=np.random):       # :80-81 (global legacy RandomState)
        return rng.randint(self.num_transitions(), size=size)

    def gather(self, idx):                               # :90 + Batch collation core_types.py:488-649
        cols = list(zip(*[self.rows[i] for i in idx]))
        return [np.array(c) for c in cols]


class StackingOracle:
    """ObservationStackingFilter for ONE env (:89-101)."""

    def __init__(self, stack_size):
        self.stack_size = stack_size
        self.stack = []

    def reset(self):
        self.stack = []

    def filter(self, observation):
        if len(self.stack) == 0:
            self.stack = deque([observation] * self.stack_size, maxlen=self.stack_size)   # :90-91
        else:
            self.stack.append(observation)                                                 # :93-94
        return np.stack(list(self.stack), axis=-1)                                          # LazyStack :37-41
