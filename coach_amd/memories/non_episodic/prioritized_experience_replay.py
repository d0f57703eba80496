"""Proportional prioritized replay in HBM — the ``PrioritizedExperienceReplay`` plug point
(rl_coach/memories/non_episodic/prioritized_experience_replay.py:159-299).

Three fp64 array-heaps (sum / min of p**alpha, max of p) with the reference's exact heap indexing
live on the device (rlx_per_*, coach_amd/csrc/sumtree.hip).  The host only makes the reference's
random draws (``random.uniform`` == a + (b-a)*random.random(), :244) and ships B doubles; descent,
importance weights, priority updates and the payload gather never leave the GPU.

Reference quirks kept on purpose:
  * capacity is rounded UP to a power of two (:176-179); a transition's leaf is its store number modulo
    that capacity, exactly the `SegmentTree.add` ring (:94-114);
  * ``store`` calls ``super().store()`` twice (:271, :280): ``num_transitions()`` counts every store
    twice (capped at the power-of-two size), which shifts the ``num_transitions() >= size`` gate and
    the N in the importance weights exactly like the reference.

Bit-exactness.  Every ``p ** alpha`` and ``(N * P) ** -beta`` is evaluated ON THE DEVICE by
rlx::libm_pow (coach_amd/csrc/libm_pow.hpp): glibc's pow algorithm on glibc's own tables in the
operation order of its x86-64 FMA build, i.e. what CPython's ``**`` executes on the host.  Trees,
sampled leaves and importance weights are therefore bit-identical to the reference's with no host
round trip (tests/test_per.py replays traces recorded from the reference; tests/test_libm_pow.py
compares the function with math.pow on millions of inputs).  ``exact_pow="host"`` keeps the older
path — |TD error| to the host, CPython pow, leaves back — for hosts whose libm is not glibc >= 2.28.

A transition's leaf exists from the moment the transition is VISIBLE (see ExperienceReplay's module
doc: the reference stores step k at the start of step k+1): the tree is written in
`_became_visible`, and `rlx_per_sample` translates a sampled leaf into the payload row of the
transition that currently owns it (the payload ring has n_env rows more than the tree has leaves).
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
    ZERO_COPY_DRAWS = True       # the B uniform draws of a sample() are read by the kernel from pinned host memory (no blit)

    def __init__(self, max_size, alpha=0.6, beta=None, epsilon=1e-6,
                 allow_duplicates_in_batch_sampling=True, exact_pow="device", **device_kwargs):
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
        # "device" (default; True is accepted as an alias): rlx::libm_pow; "host": CPython pow
        self.host_pow = exact_pow == "host"
        self.next_leaf_idx_to_write = 0
        self._list_len = 0
        self._per = {}
        self._live_slots = set()     # draws made but not collated yet: each owns its idx / rows / weight buffers
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
        self._live_slots = set()
        self.next_leaf_idx_to_write = 0
        self.lib.per_init(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size,
                          self.max_priority, _rlx.current_stream())

    def _became_visible(self, n):
        """n x PrioritizedExperienceReplay.store (:264-283) for the rows entering the window: leaves
        next_leaf .. next_leaf + n - 1 get maximal_priority (** alpha), num_transitions() += 2 n."""
        leaf0 = self.next_leaf_idx_to_write
        assert leaf0 == self.committed_total % self.power_of_2_size
        super()._became_visible(n)
        self.lib.per_store(self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size, leaf0,
                           n, self.alpha, self.max_priority, self.status, _rlx.current_stream())
        self.next_leaf_idx_to_write = (leaf0 + n) % self.power_of_2_size
        self._list_len = min(self._list_len + 2 * n, self.power_of_2_size)

    def _per_buffers(self, size, slot=0):
        p = self._per.get((size, slot))
        if p is None:
            from ...staging import Stager
            dev = self.device
            st = Stager((size,), torch.float64, dev)
            p = dict(u=st.dst, u_stage=st,
                     idx=torch.zeros(size, dtype=torch.int32, device=dev),
                     rows=torch.zeros(size, dtype=torch.int32, device=dev),
                     weight=torch.zeros(size, dtype=torch.float64, device=dev))
            self._per[(size, slot)] = p
        return p

    def draw(self, size):
        """sample() up to the payload (:219-262): the `size` random.random() draws behind
        random.uniform (:244), then — on the device, from the tree as it is NOW — the stratified
        descent and the importance weights; steps the beta schedule (:256).  Agent.train draws every
        batch of a training phase before it learns from the first (agent.py:726), so all batches of a
        phase descend the phase-start tree; each outstanding draw owns its own idx / weight buffers."""
        if self.num_transitions() < size:
            raise ValueError("The replay buffer cannot be sampled since there are not enough "
                             "transitions yet. There are currently {} transitions"
                             .format(self.num_transitions()))
        u = np.array([random.random() for _ in range(size)])
        slot = 0
        while slot in self._live_slots:      # the lowest slot no outstanding draw owns (any draw / collate interleaving)
            slot += 1
        self._live_slots.add(slot)
        p = self._per_buffers(size, slot)
        st = p["u_stage"]
        mapped = self.ZERO_COPY_DRAWS and st.cuda
        u_dev = st.push_mapped(u) if mapped else st.push(u)       # mapped: rlx_per_sample reads the pinned slot itself
        self.lib.per_sample(self.sum_tree, self.min_tree, self.power_of_2_size, u_dev, size,
                            float(self.num_transitions()), float(self.beta.current_value), p["idx"],
                            p["weight"], None, self.committed_total, self.rows, p["rows"],
                            _rlx.current_stream())
        if mapped:
            st.mapped_done()
        self.beta.step()
        return slot

    def collate(self, drawn, size):
        """The payload gather of a drawn batch (Batch collation, core_types.py:488-649).  Slot 0's
        buffers are the ones a captured update graph reads; a later slot is copied into them."""
        if drawn not in self._live_slots:
            raise ValueError("collate() of a draw that is not outstanding (slot %r)" % (drawn,))
        if drawn != 0 and 0 in self._live_slots:
            raise ValueError("slot 0's buffers are what the captured update reads: the draw that owns them must be "
                             "collated before a later one")
        self._live_slots.discard(drawn)
        p = self._per_buffers(size, 0)
        if drawn != 0:
            q = self._per_buffers(size, drawn)
            for k in ("idx", "rows", "weight"):
                p[k].copy_(q[k])
        b = self._batch_buffers(size)
        self.gather_device(p["rows"], size, b)
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
        if self.host_pow:
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

    def priority_update_args(self, indices, error_values):
        """the arguments of rlx_per_update for update_priorities(indices, error_values), or None where the update cannot ride
        on another launch (host pow, more than 64 leaves): what nn.graph.Context.per_tail takes."""
        n = int(indices.numel())
        if self.host_pow or n < 1 or n > 64 or n != int(error_values.numel()):
            return None
        return (self.sum_tree, self.min_tree, self.max_tree, self.power_of_2_size, indices, error_values, n,
                self.alpha, self.epsilon, self.max_priority, self.status)

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            if s & 2:
                raise ValueError("The priorities must be non-negative values")
            if s & 4:
                raise ValueError("a priority left the domain of the bit-exact pow (non-finite TD error?)")
            raise ValueError("The given leaf index can not be found in the tree")
