"""Episodic replay in HBM — the ``EpisodicExperienceReplay`` plug point of DDPG / TD3
(rl_coach/memories/episodic/episodic_experience_replay.py:49-130,240-317,412-426 and Episode,
core_types.py:700-820) for N envs per GPU whose episodes may end on DIFFERENT steps.

What the reference does: an agent keeps the running episode in `current_episode_buffer` and hands it to the
memory when it ends (`store_episode`, agent.py:576-584); the memory appends the episode's transitions to one flat
`transitions` list, computes the n-step discounted returns of the episode (`close_last_episode` ->
`Episode.update_transitions_rewards_and_bootstrap_data`), and evicts WHOLE oldest episodes while it holds more
transitions than `max_size` (`_enforce_max_length` :300-317).  `sample` draws
`np.random.randint(num_transitions_in_complete_episodes(), size=B)` into that list (:121).

Here the payload never moves: every vector step writes its n_env rows time-major into a ring
(row = (step mod ring_steps) * n_env + env), exactly like the flat replay.  The episode structure lives in a small
HOST table — the env front end knows which envs finished a step without asking the device — that lists, in
completion order (ties in env order), the ring rows of every complete episode: logical index i of the reference's
`transitions` list -> `order[(head + i) mod len]` -> ring row.  Completing an episode appends its rows to the table
and launches ONE small kernel for its n-step returns (rlx_episode_nstep_returns, the reference's summation order in
fp64); eviction pops whole episodes from the front; an abandoned episode (a reset in the middle) is simply never
listed.  The ring is sized so that no listed row can be overwritten: capacity + 3 * max_episode_length vector steps
(complete rows <= capacity + one episode, open rows <= n_env * max_episode_length).

With n_env = 1 the table is the identity and every draw selects what the reference selects.
"""
from collections import deque

import numpy as np
import torch

from ... import _rlx
from ...core_types import DeviceBatch
from ..memory import MemoryGranularity
from ..non_episodic.experience_replay import ExperienceReplay, ExperienceReplayParameters


class EpisodicExperienceReplayParameters(ExperienceReplayParameters):        # :32-42
    def __init__(self):
        super().__init__()
        self.max_size = (MemoryGranularity.Transitions, 1000000)
        self.n_step = -1
        self.train_to_eval_ratio = 1

    @property
    def path(self):
        return 'coach_amd.memories.episodic.episodic_experience_replay:EpisodicExperienceReplay'


class EpisodicExperienceReplay(ExperienceReplay):
    def __init__(self, max_size, allow_duplicates_in_batch_sampling=True, n_step=-1, discount=0.99,
                 max_episode_length=None, **device_kwargs):
        """
        :param max_size: (MemoryGranularity.Transitions | Episodes, n)               (reference signature)
        :param n_step: steps summed into n_step_discounted_rewards (-1: to the episode end)   (reference signature)
        :param discount: Episode(discount=...) — the agent's discount factor (agent.py:619)
        :param max_episode_length: the env's time limit; bounds the ring (defaults to min_episode_length)
        """
        if not isinstance(n_step, int) or (n_step < 1 and n_step != -1):
            raise ValueError("n-step should be an integer with value >= 1, or set to -1 for always setting to "
                             "episode length.")
        if device_kwargs.get("stack") is not None:
            raise ValueError("the episodic replay holds vector observations (DDPG / TD3); image agents use the "
                             "frame-dedup replay")
        self.n_step, self.discount = n_step, float(discount)
        self.Tmax = int(max_episode_length or device_kwargs.get("min_episode_length", 1))
        unit, amount = max_size
        n_env = int(device_kwargs.get("n_env", 1))
        if unit == MemoryGranularity.Episodes:
            self.max_episodes, transitions = int(amount), int(amount) * self.Tmax
        else:
            self.max_episodes, transitions = None, int(amount)
        transitions = -(-transitions // n_env) * n_env            # the parent wants a multiple of n_env
        self._ring_steps = transitions // n_env + 3 * self.Tmax + 2
        super().__init__((MemoryGranularity.Transitions, transitions), allow_duplicates_in_batch_sampling,
                         **device_kwargs)
        self.max_size = (unit, int(amount))
        self.n_step_discounted_rewards = torch.zeros(self.rows, dtype=torch.float64, device=self.device)
        self._order = np.zeros(transitions + 2 * self.Tmax + 1, dtype=np.int64)   # ring of listed rows
        self.clean()

    def _physical_rows(self, cap, n_env):
        return n_env * self._ring_steps

    # ------------------------------------------------------------------ Memory interface
    def clean(self):                                              # :412-426
        super().clean()
        self._gstep = 0                                           # vector steps written so far
        self._ep_start = np.zeros(self.n_env, dtype=np.int64)     # step at which each env's running episode began
        self._episodes = deque()                                  # lengths of the complete episodes, oldest first
        self._episode_first_step = deque()                        # ... and the vector step each one began at
        self._order_head, self._order_len = 0, 0

    def length(self):
        """episodes in the memory (the reference counts a non-empty open episode too; agents that store whole
        episodes never have one)."""
        return len(self._episodes)

    def num_complete_episodes(self):
        return len(self._episodes)

    def num_transitions(self):
        return self._order_len

    def num_transitions_in_complete_episodes(self):               # :84-88
        return self._order_len

    def open_transitions(self):
        """rows of the running episodes (written, not sampleable): n_env sequential current_episode_buffers."""
        return int((self._gstep - self._ep_start).sum())

    # ------------------------------------------------------------------------------ rollout side
    def store(self, actions, rewards, game_overs, next_obs, reset_obs, record=True, dones=None, defer=False,
              episode_end=False, dones_host=None):
        """One vector step (Agent.observe_transition -> current_episode_buffer.insert, agent.py:956-962), then
        `store_episode` for every env whose episode ended (dones_host: host bool[n_env]; None = lockstep envs,
        all ended iff episode_end).  `defer` is irrelevant here: an episode becomes sampleable only when it is
        complete, and its terminal transition is observed at once (level_manager.py:260-264)."""
        s = _rlx.current_stream()
        stored_go = game_overs
        if dones is not None:
            game_overs = dones
        if record:
            if self._evaluating:
                raise RuntimeError("the replay memory is not written during evaluation")
            # BEFORE the write: the ring slot of this step may still hold the oldest listed episode when steps were
            # written and never listed (open episodes dropped by a forced reset / an evaluation): evict it, like the
            # reference keeps evicting its oldest episodes (:300-317).  A RUNNING episode that old is a sizing error.
            while self._episode_first_step and self._gstep - self._episode_first_step[0] >= self._ring_steps:
                self._evict_first()
            if self._gstep - int(self._ep_start.min()) >= self._ring_steps:
                raise RuntimeError("episodic replay ring overrun: a running episode is %d vector steps old, the ring "
                                   "holds %d (episodes longer than max_episode_length=%d?)"
                                   % (self._gstep - int(self._ep_start.min()), self._ring_steps, self.Tmax))
            row0 = (self._gstep % self._ring_steps) * self.n_env
            pairs = [(actions, self.action), (rewards, self.reward), (stored_go, self.game_over),
                     (self.cur_state, self.obs), (next_obs, self.next_obs)]
            self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), None, None, 0, row0, self.n_env,
                                  self.rows, self.n_env, self.status, s)
        self.lib.select_rows(game_overs, reset_obs, next_obs, self.cur_state, self.n_env, self.obs_dim * 4, s)
        if not record:
            return
        self._gstep += 1
        if dones_host is None:
            ended = range(self.n_env) if episode_end else ()
        else:
            ended = np.nonzero(dones_host)[0]
        for e in ended:
            self._store_episode(int(e))

    def _store_episode(self, e):
        """EpisodicExperienceReplay.store_episode + close_last_episode (:264-317) for env e's finished episode."""
        s0, T = int(self._ep_start[e]), int(self._gstep - self._ep_start[e])
        self._ep_start[e] = self._gstep
        if T <= 0:
            return
        if T > self.Tmax:
            raise ValueError("an episode of %d steps exceeds max_episode_length=%d the ring was sized for"
                             % (T, self.Tmax))
        k = np.arange(T, dtype=np.int64)
        rows = ((s0 + k) % self._ring_steps) * self.n_env + e
        pos = (self._order_head + self._order_len + k) % self._order.size
        self._order[pos] = rows
        self._order_len += T
        self._episodes.append(T)
        self._episode_first_step.append(s0)
        self.lib.episode_nstep_returns(self.reward, self.n_step_discounted_rewards, None, s0, T, e, self.n_env,
                                       self._ring_steps, self.discount, self.n_step, _rlx.current_stream())
        # _enforce_max_length: whole oldest episodes leave (:300-317)
        if self.max_episodes is not None:
            while len(self._episodes) > self.max_episodes:
                self._evict_first()
        else:
            while self.max_size[1] != 0 and self._order_len > self.max_size[1]:
                self._evict_first()

    def _evict_first(self):
        T = self._episodes.popleft()
        self._episode_first_step.popleft()
        self._order_head = (self._order_head + T) % self._order.size
        self._order_len -= T

    def close_last_episode(self):
        """kept for callers that signal lockstep episode ends separately: store() already did the work."""

    def commit_pending(self):
        pass

    def drop_pending(self):
        pass

    def drop_open_episode(self):
        """Agent.reset_internal_state in the middle of an episode replaces current_episode_buffer
        (agent.py:619): the transitions of every running episode never reach the memory."""
        self._ep_start[:] = self._gstep

    # ----------------------------------------------------------------------------- training side
    def sample_indices(self, size):
        if self.num_complete_episodes() < 1:                      # :112,126-128
            raise ValueError("The episodic replay buffer cannot be sampled since there are no complete "
                             "episodes yet. There is currently 1 episodes with {} transitions"
                             .format(self.open_transitions()))
        return np.random.randint(self.num_transitions_in_complete_episodes(), size=size)   # :121

    def physical_rows(self, logical_idx):
        pos = (self._order_head + np.asarray(logical_idx, dtype=np.int64)) % self._order.size
        return self._order[pos].astype(np.int32)

    def _batch_buffers(self, size):
        b = super()._batch_buffers(size)
        if "n_step_discounted_rewards" not in b:
            b["n_step_discounted_rewards"] = torch.empty(size, dtype=torch.float64, device=self.device)
        return b

    def _extra_gather_columns(self):
        return [(self.n_step_discounted_rewards, "n_step_discounted_rewards")]

    def collate(self, drawn, size, rows_dev=None):
        b = self._gather_rows(drawn, size, rows_dev)
        return DeviceBatch(size, {"observation": b["state"]}, {"observation": b["next_state"]},
                           b["action"], b["reward"], b["game_over"],
                           info={"logical_idx": drawn, "states_pair": b["states_pair"],
                                 "n_step_discounted_rewards": b["n_step_discounted_rewards"]})

    def _steps_written_now(self):
        return self._gstep

    def episode_lengths(self):
        return list(self._episodes)
