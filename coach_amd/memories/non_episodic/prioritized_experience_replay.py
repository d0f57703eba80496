"""Proportional prioritized replay in HBM — the ``PrioritizedExperienceReplay`` plug point
(rl_coach/memories/non_episodic/prioritized_experience_replay.py:159-299).

Three fp64 array-heaps (sum / min of p**alpha, max of p) with the reference's exact heap indexing
live on the device (rlx_per_*, coach_amd/csrc/sumtree.hip).  The host only makes the reference's
random draws (``random.uniform`` == a + (b-a)*random.random(), :244) and ships B doubles; descent,
importance weights, priority updates and the payload gather never leave the GPU.

Reference quirks kept on purpose:
  * capacity is rounded UP to a power of two (:176-179) and the sampled payload is the tree's data
    ring, so a sampled leaf index IS the physical row (no FIFO translation);
  * ``store`` calls ``super().store()`` twice (:271, :280): ``num_transitions()`` counts every store
    twice (capped at the power-of-two size), which shifts the ``num_transitions() >= size`` gate and
    the N in the importance weights exactly like the reference.
`exact_pow=True` computes p**alpha for the <= B updated leaves with the host libm (bit-identical
trees, one small D2H/H2D per update); the default uses the device pow (faster, may differ from libm
in the last ulp of a leaf — the index-selection contract is then kernel-level: same tree + same
draws => same indices).
"""
import random

import numpy as np
import torch

from ... import _rlx
from ...core_types import DeviceBatch
from ...schedules import ConstantSchedule
from ..memory import MemoryGranularity
from .experience_replay import ExperienceReplay, ExperienceReplayParameters


class PrioritizedExperienceReplayParameters(ExperienceReplayParameters):     # :27-40
    def __init__(self):
        super().__init__()
        self.max_size = (MemoryGranularity.Transitions, 1000000)
        self.alpha = 0.6
        self.beta = ConstantSchedule(0.4)
        self.epsilon = 1e-6

    @property
    def path(self):
        return 'coach_amd.memories.non_episodic.prioritized_experience_replay:PrioritizedExperienceReplay'


class PrioritizedExperienceReplay(ExperienceReplay):
    def __init__(self, max_size, alpha=0.6, beta=None, epsilon=1e-6,
                 allow_duplicates_in_batch_sampling=True, exact_pow=False, **device_kwargs):
        if max_size[0] != MemoryGranularity.Transitions:                                   # :173-175
            raise ValueError("Prioritized Experience Replay currently only support setting the memory "
                             "size in transitions granularity.")
        self.power_of_2_size = 1
        while self.power_of_2_size < max_size[1]:                                          # :176-178
            self.power_of_2_size *= 2
        super().__init__((MemoryGranularity.Transitions, self.power_of_2_size),
                         allow_duplicates_in_batch_sampling, **device_kwargs)
        dev = self.device
        n = 2 * self.power_of_2_size - 1
        self.sum_tree = torch.empty(n, dtype=torch.float64, device=dev)
        self.min_tree = torch.empty(n, dtype=torch.float64, device=dev)
        self.max_tree = torch.empty(n, dtype=torch.float64, device=dev)
        self.max_priority = torch.zeros(1, dtype=torch.float64, device=dev)   # maximal_priority (:186)
        self.alpha, self.epsilon = alpha, epsilon
        self.beta = beta if beta is not None else ConstantSchedule(0.4)
        self.exact_pow = exact_pow
        self.next_leaf_idx_to_write = 0
        self._list_len = 0
        self._per = {}
        self.lib.per_init(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size,
                          self.max_priority, _rlx.current_stream())

    @property
    def maximal_priority(self):
        return float(self.max_priority.item())

    def num_transitions(self):
        return self._list_len                              # the doubled FIFO list (see module doc)

    def clean(self):                                       # :285-299
        super().clean()
        self._list_len = 0
        self.next_leaf_idx_to_write = 0
        self.lib.per_init(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size,
                          self.max_priority, _rlx.current_stream())

    def store(self, actions, rewards, game_overs, next_obs, reset_obs, record=True, dones=None):
        """n_env x PrioritizedExperienceReplay.store (:264-283): payload row == leaf index."""
        leaf0 = self.next_leaf_idx_to_write
        assert leaf0 == self.cursor or not record
        super().store(actions, rewards, game_overs, next_obs, reset_obs, record, dones)
        if not record:
            return
        self.lib.per_store(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size, leaf0,
                           self.n_env, self.alpha, self.max_priority, self.status, _rlx.current_stream())
        self.next_leaf_idx_to_write = (leaf0 + self.n_env) % self.power_of_2_size
        self._list_len = min(self._list_len + 2 * self.n_env, self.power_of_2_size)

    def _per_buffers(self, size):
        p = self._per.get(size)
        if p is None:
            from ...staging import Stager
            dev = self.device
            st = Stager((size,), torch.float64, dev)
            p = dict(u=st.dst, u_stage=st,
                     idx=torch.zeros(size, dtype=torch.int32, device=dev),
                     weight=torch.zeros(size, dtype=torch.float64, device=dev),
                     weight32=torch.zeros(size, dtype=torch.float32, device=dev))
            self._per[size] = p
        return p

    def draw(self, size):
        """Host half of sample(): the `size` random.random() draws behind random.uniform (:244), the
        N and beta of this call; steps the beta schedule (:256)."""
        if self.num_transitions() < size:
            raise ValueError("The replay buffer cannot be sampled since there are not enough "
                             "transitions yet. There are currently {} transitions"
                             .format(self.num_transitions()))
        u = np.array([random.random() for _ in range(size)])
        d = (u, float(self.num_transitions()), float(self.beta.current_value))
        self.beta.step()
        return d

    def collate(self, drawn, size):
        """Device half: stratified descent + importance weights (:229-255) and the payload gather.
        (With several batches per training phase the reference descends the tree of the phase
        start for all of them; here each descent sees the priorities updated by the previous
        batch of the phase.)"""
        u, n_transitions, beta = drawn
        p = self._per_buffers(size)
        b = self._batch_buffers(size)
        p["u_stage"].push(u)
        s = _rlx.current_stream()
        self.lib.per_sample(self.sum_tree, self.min_tree, self.power_of_2_size, p["u"], size,
                            n_transitions, beta, p["idx"], p["weight"], None, s)
        self.gather_device(p["idx"], size, b)
        return DeviceBatch(size, {"observation": b["state"]}, {"observation": b["next_state"]},
                           b["action"], b["reward"], b["game_over"],
                           info={"idx": p["idx"], "weight": p["weight"],
                                 "states_pair": b["states_pair"]})

    def sample(self, size):
        """PrioritizedExperienceReplay.sample (:219-262): stratified draws, one per segment."""
        return self.collate(self.draw(size), size)

    def update_priorities(self, indices, error_values):
        """:203-217.  indices: device int32[n] (info['idx']); error_values: device fp64[n]."""
        n = int(indices.numel())
        if n != int(error_values.numel()):
            raise ValueError("The number of indexes requested for update don't match the number of "
                             "error values given")
        s = _rlx.current_stream()
        if self.exact_pow:
            err = error_values.cpu().numpy()
            if (err < 0).any():
                raise ValueError("The priorities must be non-negative values")
            pr = err + self.epsilon
            pa = np.array([float(x) ** self.alpha for x in pr])       # CPython float.__pow__ == libm pow
            self.lib.per_update_leaves(self.sum_tree, self.min_tree, self.max_tree,
                                       self.power_of_2_size, indices,
                                       torch.from_numpy(pa).to(self.device),
                                       torch.from_numpy(pr).to(self.device), n, self.max_priority,
                                       self.status, s)
        else:
            self.lib.per_update(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size,
                                indices, error_values, n, self.alpha, self.epsilon, self.max_priority,
                                self.status, s)

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            if s & 2:
                raise ValueError("The priorities must be non-negative values")
            raise ValueError("The given leaf index can not be found in the tree")
