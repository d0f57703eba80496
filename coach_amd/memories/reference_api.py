"""The reference's memory plug point, call for call, over the device replays.

An agent written against rl_coach's `Memory` (rl_coach/memories/memory.py:41-77 + the uses in agents/agent.py) makes
these calls: `store(transition)` with a `Transition` OBJECT, `sample(size)` returning a list of Transitions whose
`info['idx']` / `info['weight']` are set by the prioritized replay
(memories/non_episodic/prioritized_experience_replay.py:250-251), `update_priorities(indices, error_values)` with Python
lists, `get_shuffled_training_data_generator(size)`, `get / get_transition`, `num_transitions`, `length`, `clean`,
`set_memory_backend`.  The device replays (experience_replay.py, prioritized_experience_replay.py) take and hand out
device COLUMNS instead — what the device agents need.  The classes here are the adapters in between: the storage, the
index draws and the sum-tree arithmetic are the device replays' (same kernels, same host RNG streams — the draws of
np.random / random are the reference's, so a reference agent samples the SAME transitions); each call converts objects
<-> columns with small host copies.  This is an API-compatibility path, not the hot path: a host round trip per call.

Returned Transition objects are host views; like the reference's stored objects, their `info` dict is the SAME dict
on every visit of a stored transition (mutations persist until the row is overwritten).
Vector observations only (an image state would have to be re-split into the frames of the de-duplicating ring).
"""
import random

import numpy as np
import torch

from ..core_types import Transition
from .memory import MemoryGranularity
from .non_episodic.experience_replay import ExperienceReplay as _DeviceExperienceReplay
from .non_episodic.prioritized_experience_replay import PrioritizedExperienceReplay as _DevicePrioritizedExperienceReplay


class _ReferenceApi(object):
    _device_class = None

    def _setup(self, device, observation_shape, action_dim, mem):
        self.device = device
        self.mem = mem
        self.memory_backend = None
        self._info = {}                          # physical row -> the stored transition's info dict
        D = int(observation_shape[0])
        self._s = torch.empty(1, D, dtype=torch.float32, pin_memory=True)
        self._ns = torch.empty(1, D, dtype=torch.float32, pin_memory=True)
        self._a = torch.empty((1,) if action_dim is None else (1, action_dim),
                              dtype=torch.int32 if action_dim is None else torch.float32, pin_memory=True)
        self._r = torch.empty(1, dtype=torch.float32, pin_memory=True)
        self._g = torch.empty(1, dtype=torch.uint8, pin_memory=True)
        dev = device
        self._ds, self._dns = self._s.to(dev), self._ns.to(dev)
        self._da, self._dr, self._dg = self._a.to(dev), self._r.to(dev), self._g.to(dev)

    # ---- Memory interface (memory.py:41-77)
    def set_memory_backend(self, memory_backend):
        self.memory_backend = memory_backend

    def length(self):
        return self.mem.length()

    def num_transitions(self):
        return self.mem.num_transitions()

    def clean(self, lock=True):
        self.mem.clean()
        self._info = {}

    def store(self, transition, lock=True):
        """ExperienceReplay.store (experience_replay.py:131-150): append, evict the oldest beyond max_size."""
        if self.memory_backend:
            self.memory_backend.store(transition)
        mem = self.mem
        self._s[0] = torch.as_tensor(np.asarray(transition.state['observation'], dtype=np.float32).reshape(-1))
        self._ns[0] = torch.as_tensor(np.asarray(transition.next_state['observation'], dtype=np.float32).reshape(-1))
        self._a[0] = torch.as_tensor(np.asarray(transition.action, dtype=self._a.numpy().dtype))
        self._r[0] = float(transition.reward)
        self._g[0] = int(bool(transition.game_over))
        for host, dev in ((self._s, self._ds), (self._ns, self._dns), (self._a, self._da), (self._r, self._dr),
                          (self._g, self._dg)):
            dev.copy_(host, non_blocking=True)
        mem.cur_state.copy_(self._ds)
        row = mem.cursor
        mem.store(self._da, self._dr, self._dg, self._dns, self._dns, record=True)
        torch.cuda.current_stream().synchronize()            # the pinned staging rows are free again
        self._info[row] = transition.info if transition.info is not None else {}

    def store_episode(self, episode, lock=True):
        """Memory.store_episode (memory.py:53-55): a transition store takes the episode's transitions one by one
        (agent.py:576-584 calls it for episodic memories only; kept for plug-point completeness)."""
        for t in episode.transitions:
            self.store(t)

    def _views(self, batch, rows):
        obs = batch.states()["observation"].cpu().numpy()
        nxt = batch.next_states()["observation"].cpu().numpy()
        act, rew, go = batch.actions().cpu().numpy(), batch.rewards().cpu().numpy(), batch.game_overs().cpu().numpy()
        out = []
        for i, row in enumerate(rows):
            t = Transition(state={'observation': obs[i].copy()}, action=act[i].item() if act.ndim == 1 else act[i].copy(),
                           reward=float(rew[i]), next_state={'observation': nxt[i].copy()}, game_over=bool(go[i]))
            t.info = self._info.setdefault(int(row), {})
            out.append(t)
        return out

    def _by_logical(self, idx):
        mem = self.mem
        rows = mem.physical_rows(idx)
        b = mem.gather(rows, len(rows))
        from ..core_types import DeviceBatch
        return self._views(DeviceBatch(len(rows), {"observation": b["state"]}, {"observation": b["next_state"]},
                                       b["action"], b["reward"], b["game_over"]), rows)

    def sample(self, size):
        """ExperienceReplay.sample (:71-93): np.random.randint(num_transitions, size) on the global legacy stream."""
        return self._by_logical(self.mem.sample_indices(size))

    def get_shuffled_training_data_generator(self, size):
        """:95-115 — one random.shuffle of all indices, whole batches only."""
        order = list(range(self.mem.num_transitions()))
        random.shuffle(order)
        for i in range(int(len(order) / size)):
            yield self._by_logical(order[i * size:(i + 1) * size])

    def get_transition(self, transition_index, lock=True):
        if transition_index >= self.mem.num_transitions() or transition_index < 0:
            return None
        return self._by_logical([transition_index])[0]

    get = get_transition

    def get_last_transition(self, lock=True):
        return self.get_transition(self.mem.num_transitions() - 1)

    def mean_reward(self):
        n = self.mem.num_transitions()
        return float(np.mean([t.reward for t in self._by_logical(list(range(n)))])) if n else 0.0

    def check_status(self):
        self.mem.check_status()


class ExperienceReplay(_ReferenceApi):
    """rl_coach.memories.non_episodic.experience_replay.ExperienceReplay, reference signature + where to keep it."""

    def __init__(self, max_size, allow_duplicates_in_batch_sampling=True, device=None, observation_shape=None,
                 action_dim=None):
        if max_size[0] != MemoryGranularity.Transitions:
            raise ValueError("Experience replay size can only be configured in terms of transitions")
        self.max_size = max_size
        self.allow_duplicates_in_batch_sampling = allow_duplicates_in_batch_sampling
        self._setup(device, observation_shape, action_dim, _DeviceExperienceReplay(
            max_size, allow_duplicates_in_batch_sampling, device=device, n_env=1,
            observation_shape=observation_shape, action_dim=action_dim))


class PrioritizedExperienceReplay(_ReferenceApi):
    """rl_coach.memories.non_episodic.prioritized_experience_replay.PrioritizedExperienceReplay (:159-283)."""

    def __init__(self, max_size, alpha=0.6, beta=None, epsilon=1e-6, allow_duplicates_in_batch_sampling=True,
                 device=None, observation_shape=None, action_dim=None):
        self.max_size = max_size
        self._setup(device, observation_shape, action_dim, _DevicePrioritizedExperienceReplay(
            max_size, alpha, beta, epsilon, allow_duplicates_in_batch_sampling, device=device, n_env=1,
            observation_shape=observation_shape, action_dim=action_dim))
        self.power_of_2_size = self.mem.power_of_2_size
        self.alpha, self.beta, self.epsilon = self.mem.alpha, self.mem.beta, self.mem.epsilon

    @property
    def maximal_priority(self):
        return self.mem.maximal_priority

    def sample(self, size):
        """:219-262 — stratified draws on the Python `random` stream; sets info['idx'] / info['weight']."""
        mem = self.mem
        batch = mem.sample(size)
        idx = batch.info("idx").cpu().numpy()
        w = batch.info("weight").cpu().numpy()
        rows = mem._per_buffers(size, 0)["rows"].cpu().numpy()
        out = self._views(batch, rows)
        for t, i, wi in zip(out, idx, w):
            t.info['idx'] = int(i)
            t.info['weight'] = float(wi)
        return out

    def update_priorities(self, indices, error_values):
        """:203-217 with Python lists, as Agent.update_transition_priorities_and_get_weights passes them."""
        if len(indices) != len(error_values):
            raise ValueError("The number of indexes requested for update don't match the number of error values given")
        dev = self.device
        self.mem.update_priorities(torch.as_tensor(np.asarray(indices, dtype=np.int32)).to(dev),
                                   torch.as_tensor(np.asarray(error_values, dtype=np.float64)).to(dev))
        self.mem.check_status()

    def sum_tree_total(self):
        return float(self.mem.sum_tree[0].item())

    def min_tree_total(self):
        return float(self.mem.min_tree[0].item())
