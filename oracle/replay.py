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


class EpisodicReplayOracle:
    """EpisodicExperienceReplay for agents that hand over whole episodes (store_episode :264-298, close_last_episode
    :240-262, _enforce_max_length :300-317, sample :102-130) + Episode.update_discounted_rewards
    (core_types.py:771-801).  Pinned by tests/golden/episodic.npz (traces of the reference classes)."""

    def __init__(self, max_size, n_step=-1, discount=0.99, by_episodes=False):
        self.max_size, self.n_step, self.discount, self.by_episodes = max_size, n_step, discount, by_episodes
        self.episodes = []                               # list of lists of rows
        self.rows, self.nsr = [], []                     # flat `transitions` list and its n-step returns

    @staticmethod
    def n_step_returns(rewards, discount, n_step):       # core_types.py:780-793
        rewards = np.asarray(rewards).astype('float')
        n = len(rewards) if (n_step == -1 or n_step > len(rewards)) else n_step
        out = rewards.copy()
        g = discount
        for i in range(1, n):
            out += g * np.pad(rewards[i:], (0, i), 'constant', constant_values=0)
            g *= discount
        return out

    def store_episode(self, rows, rewards):
        self.episodes.append(len(rows))
        self.rows.extend(rows)
        self.nsr.extend(self.n_step_returns(rewards, self.discount, self.n_step).tolist())
        if self.by_episodes:
            while len(self.episodes) > self.max_size:
                self._remove_first()
        else:
            while self.max_size != 0 and len(self.rows) > self.max_size:
                self._remove_first()

    def _remove_first(self):
        n = self.episodes.pop(0)
        del self.rows[:n]
        del self.nsr[:n]

    def num_transitions(self):
        return len(self.rows)

    def num_complete_episodes(self):
        return len(self.episodes)

    def sample_indices(self, size, rng=np.random):
        if not self.episodes:
            raise ValueError("The episodic replay buffer cannot be sampled since there are no complete episodes yet.")
        return rng.randint(len(self.rows), size=size)
