"""On-policy rollout storage in HBM — the role EpisodicExperienceReplay plays for Clipped PPO
(rl_coach/memories/episodic/episodic_experience_replay.py:102-130,240-317,412-426: the whole buffer
is the training set, `transitions` is the concatenation of the stored episodes, clean() after each
training phase).

Layout (struct of arrays, one row per transition, written time-major: row = step * n_env + env):
  image observations : frame-dedup ring u8 [n_env][ring_frames][H*W] + (t_fpos, t_epoff) per row
                       (rlx_imgreplay_*; a stacked state is materialised only when gathered)
  vector observations: f32 [capacity][D]
  action i32 / f32[A], reward f32, game_over u8
The DATASET order the agent trains on is episode-major like the reference's `transitions` list:
dataset index i = env * steps + step  ->  row(i) = step * n_env + env   (`dataset_rows`).

Envs whose episodes end on different steps (`note_episode_ends` is then called after every step with the env front
end's host-side `dones`): the dataset is the list of COMPLETE episodes in the order they completed (ties in env
order — the order `store_episode` calls would arrive, episodic_experience_replay.py:264-317); the transitions of
the still-open episodes are not part of it (num_transitions_in_complete_episodes, :84-88) and go with clean().
"""
import numpy as np
import torch

from ... import _rlx


class DeviceEpisodicRolloutBuffer(object):
    def __init__(self, device, n_env, max_steps_per_env, frame_shape=None, stack=4, obs_dim=None,
                 action_dim=None):
        self.device, self.n_env, self.T = device, n_env, max_steps_per_env
        self.lib = _rlx.lib()
        self.cap = n_env * max_steps_per_env
        self.image = frame_shape is not None
        self.stack = stack
        if self.image:
            self.frame_shape = tuple(frame_shape)
            self.fb = int(np.prod(frame_shape))
            # every step adds a frame, every episode end one more (the post-reset frame)
            self.F = 2 * max_steps_per_env + stack + 4
            self.ring = torch.zeros(n_env, self.F, self.fb, dtype=torch.uint8, device=device)
            self.fpos = torch.zeros(n_env, dtype=torch.int32, device=device)
            self.epoff = torch.zeros(n_env, dtype=torch.int32, device=device)
            self.t_fpos = torch.zeros(self.cap, dtype=torch.int32, device=device)
            self.t_epoff = torch.zeros(self.cap, dtype=torch.uint8, device=device)
            self.cur_state = torch.empty((n_env,) + self.frame_shape + (stack,), dtype=torch.uint8,
                                         device=device)
        else:
            self.obs_dim = obs_dim
            self.obs = torch.zeros(self.cap, obs_dim, dtype=torch.float32, device=device)
            # Transition.next_state: a pre-network filter that keeps running statistics sees the next states of the
            # dataset too (InputFilter.filter walks state AND next_state of every Transition, filters/filter.py:314-333)
            self.next_obs = torch.zeros(self.cap, obs_dim, dtype=torch.float32, device=device)
            self.cur_state = torch.empty(n_env, obs_dim, dtype=torch.float32, device=device)
        self.action = torch.zeros(self.cap, dtype=torch.int32, device=device) if action_dim is None \
            else torch.zeros(self.cap, action_dim, dtype=torch.float32, device=device)
        self.reward = torch.zeros(self.cap, dtype=torch.float32, device=device)
        self.game_over = torch.zeros(self.cap, dtype=torch.uint8, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.steps = 0
        self._rows_cache = {}
        # ragged mode (note_episode_ends): host table of the complete episodes + ONE static device row list
        self.ragged = False
        self._episodes = []                                   # (env, first step, end step) in completion order
        self._ep_start = np.zeros(n_env, dtype=np.int64)
        self._rows_dev = None
        self._rows_valid = -1

    def enable_policy_columns(self, n_actions):
        """Two more columns per transition, written by the acting step itself (ClippedPPONet.act_and_record): V(s) of
        the stored state and the action probabilities the action was drawn from — what fill_advantages and the
        old-policy pass of the training phase would otherwise recompute over the whole dataset."""
        self.act_value = torch.zeros(self.cap, dtype=torch.float32, device=self.device)
        self.act_probs = torch.zeros(self.cap, n_actions, dtype=torch.float32, device=self.device)

    # ---- Memory interface (memories/memory.py:41-77) ------------------------------------------
    def num_transitions(self):
        if self.ragged:
            return self.num_transitions_in_complete_episodes()
        return self.steps * self.n_env

    def num_transitions_in_complete_episodes(self):      # :84-88
        if self.ragged:
            return int(sum(b - a for _, a, b in self._episodes))
        return self.steps * self.n_env

    def length(self):
        if self.ragged:
            return len(self._episodes)
        return self.n_env if self.steps > 0 else 0

    def clean(self):                                     # :412-426
        self.steps = 0
        self._episodes = []
        self._ep_start[:] = 0
        self._rows_valid = -1

    def note_episode_ends(self, dones_host):
        """after the store of a step: which envs' episodes ended on it (host bool[n_env]).  Switches the buffer to
        the complete-episodes dataset."""
        self.ragged = True
        for e in np.nonzero(dones_host)[0]:
            self._episodes.append((int(e), int(self._ep_start[e]), int(self.steps)))
            self._ep_start[e] = self.steps

    def drop_open_episodes(self, episode_length=None):
        """Agent.reset_internal_state in the middle of an episode replaces current_episode_buffer (agent.py:619): the
        transitions of every running episode never reach the memory.

        Ragged bookkeeping: the open episodes simply are never listed; their rows stay allocated until clean().
        Lockstep envs (`episode_length` = L, every env ends its episode on steps L-1, 2L-1, ...): the rows of the
        open episodes are the steps since the last multiple of L — `steps` is rewound to it, so the dataset
        (`steps * n_env` rows) holds complete episodes only and the next episode's rows overwrite the abandoned ones.
        The frame ring is rewound with it: the last kept step of every env ended an episode, so its append left the
        env at (t_fpos + 2, epoff 0) — the post-reset frame that the coming reset replaces in place."""
        if self.ragged or episode_length is None:
            self._ep_start[:] = self.steps
            return 0
        keep = self.steps - self.steps % int(episode_length)
        dropped = self.steps - keep
        if dropped and self.image and keep > 0:
            last = self.t_fpos[(keep - 1) * self.n_env:keep * self.n_env]
            self.fpos.copy_(torch.remainder(last + 2, self.F))
            self.epoff.zero_()
        self.steps = keep
        return dropped

    # ---- rollout side -------------------------------------------------------------------------
    def reset(self, first_obs):
        """First observation of every env: the stacking filter replicates it (:90-91)."""
        s = _rlx.current_stream()
        if self.image:
            self.lib.imgreplay_reset(self.ring, self.fpos, self.epoff, first_obs, self.n_env, self.F,
                                     self.fb, s)
        else:
            self.cur_state.copy_(first_obs)

    def current_states(self):
        """Stacked state of every env (policy input).  28 224 B per env for Atari."""
        if self.image:
            self.lib.imgreplay_gather(self.ring, None, None, self.fpos, self.epoff, None, self.n_env,
                                      self.n_env, self.F, self.fb, self.stack, self.cap,
                                      self.cur_state, None, self.status, _rlx.current_stream())
        return self.cur_state

    def store_step(self, actions, rewards, game_overs, next_obs, reset_obs):
        """Agent.observe/observe_transition for n_env transitions (agents/agent.py:905-973)."""
        if self.steps >= self.T:
            raise ValueError("rollout buffer is full (%d steps per env)" % self.T)
        self.store_step_at(self.steps, actions, rewards, game_overs, next_obs, reset_obs)
        self.steps += 1

    def store_step_at(self, step, actions, rewards, game_overs, next_obs, reset_obs, columns=True):
        """Device work of store_step for an explicit step index (hipGraph-capturable: no host state).
        columns=False: the action / reward / game_over columns of this step are already written
        (rlx_rollout_observe_step); only the observations are stored here."""
        s = _rlx.current_stream()
        row0 = step * self.n_env
        pairs = [(actions, self.action), (rewards, self.reward), (game_overs, self.game_over)] if columns else []
        if not self.image:
            pairs.append((self.cur_state, self.obs))
            pairs.append((next_obs, self.next_obs))          # the env's response (the terminal observation at an episode end)
        if pairs:
            self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), None, None, 0, row0, self.n_env,
                                  self.cap, self.n_env, self.status, s)
        if self.image:
            self.lib.imgreplay_append(self.ring, self.fpos, self.epoff, self.t_fpos, self.t_epoff,
                                      next_obs, reset_obs, game_overs, self.n_env, self.F, self.fb,
                                      self.stack, row0, self.cap, 1, s)
        else:
            # next state of a finished episode is the post-reset observation
            self.cur_state.copy_(torch.where(game_overs.view(-1, 1).bool(), reset_obs, next_obs))

    # ---- training side ------------------------------------------------------------------------
    def dataset_rows(self):
        """int32[n_env * steps]: storage row of dataset element i (episode-major order).
        Ragged mode: a static int32[cap] buffer whose first num_transitions() entries list the complete episodes."""
        if self.ragged:
            if self._rows_dev is None:
                from ...staging import Stager
                self._rows_stager = Stager((self.cap,), torch.int32, self.device, depth=4)
                self._rows_dev = self._rows_stager.dst
            if self._rows_valid != len(self._episodes):
                rows = np.zeros(self.cap, dtype=np.int32)
                parts = [np.arange(a, b, dtype=np.int64) * self.n_env + e for e, a, b in self._episodes]
                if parts:
                    flat = np.concatenate(parts)
                    rows[:flat.size] = flat
                self._rows_stager.push(rows)
                self._rows_valid = len(self._episodes)
            return self._rows_dev
        key = self.steps
        if key not in self._rows_cache:
            e = np.arange(self.n_env)[:, None]
            t = np.arange(self.steps)[None, :]
            rows = (t * self.n_env + e).reshape(-1).astype(np.int32)
            self._rows_cache[key] = torch.from_numpy(rows).to(self.device)
        return self._rows_cache[key]

    def gather_states(self, rows, n, out):
        """Stacked states (or vectors) of the given storage rows -> out [n, ...]."""
        s = _rlx.current_stream()
        if self.image:
            self.lib.imgreplay_gather(self.ring, self.t_fpos, self.t_epoff, None, None, rows, n,
                                      self.n_env, self.F, self.fb, self.stack, self.cap, out, None,
                                      self.status, s)
        else:
            self.lib.copy_columns(_rlx.make_columns([(self.obs, out)]), 1, rows, None, 0, 0, self.cap,
                                  n, n, self.status, s)
        return out

    def gather_next_states(self, rows, n, out):
        """Transition.next_state of the given storage rows (vector observations) -> out [n, obs_dim]."""
        if self.image:
            raise NotImplementedError("next states of image rollouts are not kept (no filter reads them)")
        self.lib.copy_columns(_rlx.make_columns([(self.next_obs, out)]), 1, rows, None, 0, 0, self.cap, n, n,
                              self.status, _rlx.current_stream())
        return out

    def gather_columns(self, rows, n, pairs):
        self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), rows, None, 0, 0, self.cap, n, n,
                              self.status, _rlx.current_stream())
