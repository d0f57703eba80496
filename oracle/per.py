"""Prioritized replay segment trees — numpy restatement (pinned: reference tests + golden npz).

Follows rl_coach/memories/non_episodic/prioritized_experience_replay.py:
SegmentTree :43-156, PrioritizedExperienceReplay :159-299.  Recursion is unrolled into loops,
arithmetic (fp64, operation order) is kept identical so that indices AND tree contents are
bit-exact with the reference.
"""
import math

import numpy as np


class SegmentTree:
    """:43-156.  op in {'sum','min','max'}."""
    _INIT = {'sum': 0.0, 'min': float('inf'), 'max': -float('inf')}

    def __init__(self, size, op):
        if not (size > 0 and size & (size - 1) == 0):               # :62-63
            raise ValueError("A segment tree size must be a positive power of 2. "
                             "The given size is {}".format(size))
        self.size = size
        self.op = op
        self.tree = np.ones(2 * size - 1) * self._INIT[op]          # :65
        self.next_leaf_idx_to_write = 0

    def _combine(self, a, b):
        if self.op == 'sum':
            return a + b
        if self.op == 'min':
            return b if b < a else a                                # python min(a, b)
        return b if b > a else a                                    # python max(a, b)

    def update(self, leaf_idx, new_val):                            # :116-129 + _propagate :63-74
        node = leaf_idx + self.size - 1
        if not 0 <= node < len(self.tree):
            raise ValueError("The given left index ({}) can not be found in the tree. "
                             "The available leaves are: 0-{}".format(leaf_idx, self.size - 1))
        self.tree[node] = new_val
        while node != 0:
            parent = (node - 1) // 2
            self.tree[parent] = self._combine(self.tree[parent * 2 + 1], self.tree[parent * 2 + 2])
            node = parent

    def add(self, val):                                             # :100-114
        leaf = self.next_leaf_idx_to_write
        self.update(leaf, val)
        self.next_leaf_idx_to_write += 1
        if self.next_leaf_idx_to_write >= self.size:
            self.next_leaf_idx_to_write = 0
        return leaf

    def total_value(self):                                          # :94-98
        return self.tree[0]

    def retrieve(self, val):                                        # _retrieve :76-92
        node = 0
        while True:
            left = 2 * node + 1
            if left >= len(self.tree):
                break
            if val <= self.tree[left]:
                node = left
            else:
                val = val - self.tree[left]
                node = left + 1
        return node

    def get_element_by_partial_sum(self, val):                      # :131-146
        node = self.retrieve(val)
        return node - self.size + 1, self.tree[node]


class PrioritizedReplayOracle:
    """Index/priority side of PrioritizedExperienceReplay (:159-299); payload handled elsewhere."""

    def __init__(self, max_size, alpha=0.6, beta=0.4, epsilon=1e-6):
        self.power_of_2_size = 1
        while self.power_of_2_size < max_size:                      # :176-179
            self.power_of_2_size *= 2
        self.sum_tree = SegmentTree(self.power_of_2_size, 'sum')
        self.min_tree = SegmentTree(self.power_of_2_size, 'min')
        self.max_tree = SegmentTree(self.power_of_2_size, 'max')
        self.alpha, self.beta, self.epsilon = alpha, beta, epsilon
        self.maximal_priority = 1.0                                 # :186
        self._list_len = 0     # len(self.transitions) of the parent ExperienceReplay

    def num_transitions(self):
        return self._list_len

    def store(self):                                                # :264-283
        # the reference calls super().store() twice (:271 and :280) -> the FIFO list grows by 2
        self._list_len = min(self._list_len + 1, self.power_of_2_size)
        p = self.maximal_priority
        leaf = self.sum_tree.add(p ** self.alpha)
        self.min_tree.add(p ** self.alpha)
        self.max_tree.add(p)
        self._list_len = min(self._list_len + 1, self.power_of_2_size)
        return leaf

    def update_priorities(self, indices, error_values):             # :203-217 + :188-201
        if len(indices) != len(error_values):
            raise ValueError("The number of indexes requested for update don't match the number "
                             "of error values given")
        for leaf_idx, error in zip(indices, error_values):
            if error < 0:
                raise ValueError("The priorities must be non-negative values")
            priority = (error + self.epsilon)
            self.sum_tree.update(leaf_idx, priority ** self.alpha)
            self.min_tree.update(leaf_idx, priority ** self.alpha)
            self.max_tree.update(leaf_idx, priority)
            self.maximal_priority = self.max_tree.total_value()

    def sample(self, size, uniforms):
        """:229-255 with the host draws made explicit: uniforms[i] = random.random(), so that
        val = a + (b-a)*u == random.uniform(a, b) (CPython Lib/random.py)."""
        if self.num_transitions() < size:
            raise ValueError("The replay buffer cannot be sampled since there are not enough "
                             "transitions yet. There are currently {} transitions"
                             .format(self.num_transitions()))
        total = self.sum_tree.total_value()
        segment_size = total / size
        min_probability = self.min_tree.total_value() / total
        max_weight = (min_probability * self.num_transitions()) ** -self.beta
        idx = np.zeros(size, dtype=np.int64)
        weights = np.zeros(size, dtype=np.float64)
        for i in range(size):
            a = segment_size * i
            b = segment_size * (i + 1)
            val = a + (b - a) * float(uniforms[i])
            leaf_idx, priority = self.sum_tree.get_element_by_partial_sum(val)
            priority = priority / total
            weight = (self.num_transitions() * priority) ** -self.beta
            idx[i] = leaf_idx
            weights[i] = weight / max_weight
        return idx, weights
