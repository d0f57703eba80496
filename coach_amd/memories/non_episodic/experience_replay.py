"""Uniform experience replay in HBM — the ``ExperienceReplay`` plug point
(rl_coach/memories/non_episodic/experience_replay.py:28-150) for N lockstep envs per GPU.

The reference keeps a Python list of Transition objects (append + ``del list[0]`` FIFO) and samples
``np.random.randint(num_transitions, size=B)`` on the global legacy RandomState (:81).  Here:

  * rows live in a struct-of-arrays ring; logical index i of the reference's list (i-th oldest) is
    physical row (head + i) % capacity, so the SAME host draw selects the SAME transition;
  * the host makes the draw with the same generator (bit-exact index selection), ships B int32 and
    one launch (rlx_copy_columns / rlx_imgreplay_gather) collates the whole Batch
    (core_types.py:488-649) on the device;
  * ``store`` takes the n_env transitions of one vector step, in env order — what n_env sequential
    ``store(transition)`` calls would append.

Image observations use the frame-dedup ring of rlx_imgreplay_* (each 84x84 frame stored once per
env; a 4-stack is materialised only when gathered): a 1 M-transition Atari buffer is
~7-14 GB of HBM instead of 56 GB of stacked states.

WHEN a transition becomes visible.  In the reference the response of env.step at step k is observed
— turned into a Transition and stored — at the START of step k+1, after the train() call of step k
(level_manager.py:236-251, agent.py:905-973); only a terminal response is observed at once
(level_manager.py:260-264).  ``store(..., defer=True)`` reproduces that without a second copy: the
ring has n_env rows more than the capacity, the rows of the step are written at once (the acting path
needs nothing from them) but stay outside the sampled window — `num_transitions()`, the FIFO head and
every eviction lag — until `commit_pending()` (called at the start of the next vector step) moves the
window over them.  `drop_pending()` is the reference's reset in the middle of an episode: the last
response was never observed, so its transition never existed.

Evaluation (TEST phase) never touches the memory in the reference; here `begin_evaluation()` switches
the acting state to a scratch stack of its own, so evaluation frames never enter the replay ring.
"""
import numpy as np
import torch

from ... import _rlx
from ...core_types import DeviceBatch
from ..memory import Memory, MemoryGranularity, MemoryParameters


class ExperienceReplayParameters(MemoryParameters):      # experience_replay.py:28-36
    def __init__(self):
        super().__init__()
        self.max_size = (MemoryGranularity.Transitions, 1000000)
        self.allow_duplicates_in_batch_sampling = True

    @property
    def path(self):
        return 'coach_amd.memories.non_episodic.experience_replay:ExperienceReplay'


class ExperienceReplay(Memory):
    GATHER_ONE_LAUNCH = True     # image replay: the batch's small columns ride on the frame gather's launch

    def __init__(self, max_size, allow_duplicates_in_batch_sampling=True, device=None, n_env=1,
                 observation_shape=None, stack=None, action_dim=None, min_episode_length=1):
        """
        :param max_size: (MemoryGranularity.Transitions, n)                    (reference signature)
        :param allow_duplicates_in_batch_sampling: sample with replacement      (reference signature)
        :param n_env: envs stepping in lockstep (a vector step stores n_env rows)
        :param observation_shape: (D,) vector observations, or (H, W) frames when `stack` is given
        :param stack: frames per stacked image state (ObservationStackingFilter), None for vectors
        :param action_dim: None = discrete (int32), else continuous fp32[action_dim]
        :param min_episode_length: bounds the frames an env adds per stored step (1 + 1/L)
        """
        super().__init__(max_size)
        if max_size[0] != MemoryGranularity.Transitions:                        # :48-49
            raise ValueError("Experience replay size can only be configured in terms of transitions")
        if device is None or observation_shape is None:
            raise ValueError("the device replay needs a device and an observation shape")
        self.lib = _rlx.lib()
        self.device, self.n_env = device, int(n_env)
        self.allow_duplicates_in_batch_sampling = allow_duplicates_in_batch_sampling
        cap = int(max_size[1])
        if cap <= 0 or cap % self.n_env:
            raise ValueError("replay capacity %d must be a positive multiple of the %d lockstep envs"
                             % (cap, self.n_env))
        self.cap = cap
        # physical rows: the capacity plus one vector step that is written but not visible yet
        rows = self.rows = self._physical_rows(cap, self.n_env)
        self.image = stack is not None
        self.stack = stack
        dev = device
        if self.image:
            self.frame_shape = tuple(observation_shape)
            self.fb = int(np.prod(observation_shape))
            per_env = rows // self.n_env
            # frames an env can add while `per_env` of its transitions are alive: one per step, one
            # more per episode end (the post-reset frame), + the stack of the oldest state + slack
            self.F = per_env + per_env // max(1, int(min_episode_length)) + stack + 6
            self.ring = torch.zeros(self.n_env, self.F, self.fb, dtype=torch.uint8, device=dev)
            self.fpos = torch.zeros(self.n_env, dtype=torch.int32, device=dev)
            self.epoff = torch.zeros(self.n_env, dtype=torch.int32, device=dev)
            self.t_fpos = torch.zeros(rows, dtype=torch.int32, device=dev)
            self.t_epoff = torch.zeros(rows, dtype=torch.uint8, device=dev)
            self.cur_state = torch.empty((self.n_env,) + self.frame_shape + (stack,),
                                         dtype=torch.uint8, device=dev)
            self.state_shape = self.frame_shape + (stack,)
            self.state_dtype = torch.uint8
        else:
            self.obs_dim = int(observation_shape[0])
            self.obs = torch.zeros(rows, self.obs_dim, dtype=torch.float32, device=dev)
            self.next_obs = torch.zeros(rows, self.obs_dim, dtype=torch.float32, device=dev)
            self.cur_state = torch.empty(self.n_env, self.obs_dim, dtype=torch.float32, device=dev)
            self.state_shape = (self.obs_dim,)
            self.state_dtype = torch.float32
        self.action_dim = action_dim
        self.action = torch.zeros(rows, dtype=torch.int32, device=dev) if action_dim is None else \
            torch.zeros(rows, action_dim, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(rows, dtype=torch.float32, device=dev)
        self.game_over = torch.zeros(rows, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cursor = 0          # physical row the next vector step is written to
        self.count = 0           # len(self.transitions) of the reference (the VISIBLE transitions)
        self.pending = 0         # rows written at [cursor - pending, cursor) that are not visible yet
        self.committed_total = 0  # transitions ever made visible (PER: leaf = this % capacity)
        self._steps_written = 0   # vector steps written (cursor = steps_written * n_env mod rows)
        self._evaluating = False
        self._eval = None
        self._batches = {}
        self._pinned = {}
        # host mirror of the frame ring's fill (image mode): frames an env has appended so far, the
        # value of that counter at every stored vector step, steps of the running episode — enough
        # to PROVE before every write that no frame of a visible transition is overwritten
        self._frames_total = 0
        self._step_frames = np.zeros(rows // self.n_env, dtype=np.int64)
        self._episode_steps = 0

    def _physical_rows(self, cap, n_env):
        return cap + n_env

    def _extra_gather_columns(self):
        """[(stored column, batch-buffer key)] a subclass wants collated with the Batch (same launch)."""
        return []

    # ---------------------------------------------------------------- Memory interface (:56-69)
    def length(self):
        return self.num_transitions()

    def num_transitions(self):
        return self.count

    def clean(self):
        self.cursor = 0
        self.count = 0
        self.pending = 0
        self.committed_total = 0
        self._steps_written = 0

    def head(self):
        """physical row of logical index 0 (the oldest VISIBLE transition)."""
        return (self.cursor - self.pending - self.count) % self.rows

    def _steps_written_now(self):
        """vector steps whose rows were written into the time-major ring so far (modulo nothing)."""
        return self._steps_written

    def episode_discounted_returns(self, env, length, discount, n_step=-1):
        """Episode.update_discounted_rewards (core_types.py:771-801) over env's `length` most recently stored
        transitions — the samples of the 'Discounted Return' signal (agent.py:565-566) — as a device fp64 vector."""
        if length > self.rows // self.n_env:
            raise ValueError("the episode is longer than the replay ring")
        if getattr(self, "_dr_scratch", None) is None or self._dr_scratch.numel() < length:
            self._dr_scratch = torch.empty(max(length, 1024), dtype=torch.float64, device=self.device)
        self.lib.episode_nstep_returns(self.reward, None, self._dr_scratch, self._steps_written_now() - length, length,
                                       env, self.n_env, self.rows // self.n_env, float(discount), int(n_step),
                                       _rlx.current_stream())
        return self._dr_scratch[:length]

    # ------------------------------------------------------------- visibility (see module doc)
    def commit_pending(self):
        """The observe() at the start of the next step: the held rows enter the sampled window and
        the oldest rows beyond the capacity leave it (_enforce_max_length, :141-150)."""
        n = self.pending
        if n:
            self.pending = 0
            self._became_visible(n)

    def drop_pending(self):
        """Reset before the held response was observed: its transitions are never stored."""
        if self.pending:
            self.cursor = (self.cursor - self.pending) % self.rows
            self._steps_written -= self.pending // self.n_env
            self.pending = 0

    def _became_visible(self, n):
        self.count = min(self.count + n, self.cap)
        self.committed_total += n

    # ------------------------------------------------------------------------------ rollout side
    def reset(self, first_obs):
        """First observation of a new episode in every env (an episode start, not a stored step)."""
        s = _rlx.current_stream()
        if self.image:
            ring, fpos, epoff, F = self._stack_state()
            self.lib.imgreplay_reset(ring, fpos, epoff, first_obs, self.n_env, F, self.fb, s)
            if not self._evaluating:
                if self._episode_steps:          # an un-stepped episode is restarted in place
                    self._frames_total += 1
                self._episode_steps = 0
        else:
            self.cur_state.copy_(first_obs)

    def _account_frames(self, row0, episode_end):
        """Raise BEFORE the append that would overwrite a frame some visible transition still needs
        (more episode starts than `min_episode_length` promised — e.g. many aborted episodes)."""
        self._step_frames[row0 // self.n_env] = self._frames_total
        self._frames_total += 2 if episode_end else 1
        self._episode_steps = 0 if episode_end else self._episode_steps + 1
        if self.count:
            oldest = self._step_frames[self.head() // self.n_env]
            if self._frames_total - oldest + self.stack > self.F:
                raise RuntimeError(
                    "image replay frame ring overrun: %d frames are alive but the ring holds %d per env "
                    "(episodes shorter than min_episode_length=%s?)"
                    % (self._frames_total - oldest + self.stack, self.F, "configured"))

    def _stack_state(self):
        """(ring, fpos, epoff, ring_frames) the ACTING path reads and advances: the replay ring while
        training / heating up, a scratch ring of stack + 4 frames per env while evaluating."""
        if not self._evaluating:
            return self.ring, self.fpos, self.epoff, self.F
        if self._eval is None:
            dev, Fe = self.device, self.stack + 4
            self._eval = (torch.zeros(self.n_env, Fe, self.fb, dtype=torch.uint8, device=dev),
                          torch.zeros(self.n_env, dtype=torch.int32, device=dev),
                          torch.zeros(self.n_env, dtype=torch.int32, device=dev), Fe)
        return self._eval

    def begin_evaluation(self, first_obs):
        """GraphManager.evaluate resets every level before it acts in TEST phase
        (graph_manager.py:505-507): the running training episode is abandoned — its last response is
        never observed — and nothing evaluation does reaches the memory."""
        self.drop_pending()
        self._evaluating = True
        self.reset(first_obs)

    def end_evaluation(self, first_obs):
        """Back to training: the next training step starts a fresh episode from `first_obs`."""
        self._evaluating = False
        self.reset(first_obs)

    def current_states(self):
        if self.image:
            ring, fpos, epoff, F = self._stack_state()
            self.lib.imgreplay_gather(ring, None, None, fpos, epoff, None, self.n_env,
                                      self.n_env, F, self.fb, self.stack, self.rows,
                                      self.cur_state, None, self.status, _rlx.current_stream())
        return self.cur_state

    def store(self, actions, rewards, game_overs, next_obs, reset_obs, record=True, dones=None,
              defer=False, episode_end=False, dones_host=None):
        """n_env transitions (state = current state of every env), then advance the env states.
        Reference: ExperienceReplay.store + _enforce_max_length (:117-150) called n_env times.
        record=False only advances the observation state (evaluation episodes are not stored).
        dones: the env's true episode-end flags when the STORED game_over differs from them (TD3
        clears game_over on time-limit terminations, td3_agent.py:215-227).
        defer=True: the rows are written now but become visible at the next commit_pending() (the
        reference observes a non-terminal response at the start of the NEXT step).
        episode_end: host copy of "the episodes ended on this step" (lockstep envs), used only for
        the frame-ring accounting of image observations."""
        s = _rlx.current_stream()
        if record:
            self.commit_pending()
        row0 = self.cursor
        stored_go = game_overs
        if dones is not None:
            game_overs = dones
        if record and self._evaluating:
            raise RuntimeError("the replay memory is not written during evaluation")
        if record and self.image:
            self._account_frames(row0, episode_end)
        if record:
            pairs = [(actions, self.action), (rewards, self.reward), (stored_go, self.game_over)]
            if not self.image:
                pairs += [(self.cur_state, self.obs), (next_obs, self.next_obs)]
            self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), None, None, 0, row0,
                                  self.n_env, self.rows, self.n_env, self.status, s)
        if self.image:
            ring, fpos, epoff, F = self._stack_state()
            self.lib.imgreplay_append(ring, fpos, epoff, self.t_fpos, self.t_epoff,
                                      next_obs, reset_obs, game_overs, self.n_env, F, self.fb,
                                      self.stack, row0, self.rows, int(record), s)
        else:
            # the next state of a finished episode is the post-reset observation
            self.lib.select_rows(game_overs, reset_obs, next_obs, self.cur_state, self.n_env,
                                 self.obs_dim * 4, s)
        if record:
            self.cursor = (self.cursor + self.n_env) % self.rows
            self._steps_written += 1
            if defer:
                self.pending = self.n_env
            else:
                self._became_visible(self.n_env)

    def reserve_step(self, defer):
        """The HOST half of store(): where the rows of this vector step go (-> row0) and how the sampled window
        moves.  The device half may then run from a captured graph whose destination rows arrive as data."""
        self.commit_pending()
        row0 = self.cursor
        self.cursor = (self.cursor + self.n_env) % self.rows
        self._steps_written += 1
        if defer:
            self.pending = self.n_env
        else:
            self._became_visible(self.n_env)
        return row0

    def store_device(self, actions, rewards, game_overs, next_obs, reset_obs, dst_rows):
        """The DEVICE half of store() for vector observations with the destination rows as a device int32[n_env]
        (staged with the step's other host draws): pure launches on static buffers."""
        s = _rlx.current_stream()
        pairs = [(actions, self.action), (rewards, self.reward), (game_overs, self.game_over),
                 (self.cur_state, self.obs), (next_obs, self.next_obs)]
        self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), None, dst_rows, 0, 0, self.n_env, self.rows,
                              self.n_env, self.status, s)
        self.lib.select_rows(game_overs, reset_obs, next_obs, self.cur_state, self.n_env, self.obs_dim * 4, s)

    # ----------------------------------------------------------------------------- training side
    def sample_indices(self, size):
        """The reference's draw (:80-86) on the global legacy np.random stream -> LOGICAL indices."""
        if self.allow_duplicates_in_batch_sampling:
            return np.random.randint(self.num_transitions(), size=size)
        if self.num_transitions() >= size:
            return np.random.choice(self.num_transitions(), size=size, replace=False)
        raise ValueError("The replay buffer cannot be sampled since there are not enough transitions "
                         "yet. There are currently {} transitions".format(self.num_transitions()))

    def physical_rows(self, logical_idx):
        return ((self.head() + np.asarray(logical_idx, dtype=np.int64)) % self.rows).astype(np.int32)

    def _batch_buffers(self, size):
        b = self._batches.get(size)
        if b is None:
            dev = self.device
            # states and next_states of a batch are two halves of ONE buffer, so that the online
            # network on s and the target network on s' can run as two towers of the same launches
            both = torch.empty((2, size) + self.state_shape, dtype=self.state_dtype, device=dev)
            b = dict(
                rows=torch.zeros(size, dtype=torch.int32, device=dev),
                state=both[0], next_state=both[1], states_pair=both,
                action=torch.empty((size,) if self.action_dim is None else (size, self.action_dim),
                                   dtype=self.action.dtype, device=dev),
                reward=torch.empty(size, dtype=torch.float32, device=dev),
                game_over=torch.empty(size, dtype=torch.uint8, device=dev))
            from ...staging import Stager
            st = Stager((size,), torch.int32, dev)
            b["rows"] = st.dst
            self._batches[size] = b
            self._pinned[size] = st
        return b

    def gather(self, rows_host, size):
        """Collate the Batch of the given PHYSICAL rows (one H2D copy of B int32 + 2 launches)."""
        b = self._batch_buffers(size)
        self._pinned[size].push(np.asarray(rows_host, dtype=np.int32))
        self.gather_device(b["rows"], size, b)
        return b

    def gather_device(self, rows, size, b):
        s = _rlx.current_stream()
        pairs = [(self.action, b["action"]), (self.reward, b["reward"]), (self.game_over, b["game_over"])]
        if not self.image:
            pairs += [(self.obs, b["state"]), (self.next_obs, b["next_state"])]
        pairs += [(src, b[key]) for src, key in self._extra_gather_columns()]
        if self.image and self.stack == 4 and self.GATHER_ONE_LAUNCH and \
                all((src[0].numel() if src.dim() > 1 else 1) * src.element_size() * size <= 65536 for src, _ in pairs):
            # the small columns ride on the frame gather's launch (rlx_imgreplay_gather_columns)
            self.lib.imgreplay_gather_columns(self.ring, self.t_fpos, self.t_epoff, rows, size, self.n_env, self.F,
                                              self.fb, self.stack, self.rows, b["state"], b["next_state"],
                                              _rlx.make_columns(pairs), len(pairs), self.status, s)
            return
        self.lib.copy_columns(_rlx.make_columns(pairs), len(pairs), rows, None, 0, 0, self.rows, size,
                              size, self.status, s)
        if self.image:
            self.lib.imgreplay_gather(self.ring, self.t_fpos, self.t_epoff, None, None, rows, size,
                                      self.n_env, self.F, self.fb, self.stack, self.rows, b["state"],
                                      b["next_state"], self.status, s)

    def draw(self, size):
        """The host-RNG half of sample(): made for every batch of a training phase up front, in the
        reference's order ([sample(B) for _ in range(num_consecutive_training_steps)], agent.py:726)."""
        return self.sample_indices(size)

    def _gather_rows(self, drawn, size, rows_dev):
        """the batch buffers filled from the physical rows of a draw: shipped here (one small copy), or already on the
        device in `rows_dev` — the agent staged them together with the update's other host draws (RecordStager)."""
        if rows_dev is None:
            return self.gather(self.physical_rows(drawn), size)
        b = self._batch_buffers(size)
        self.gather_device(rows_dev, size, b)
        return b

    def collate(self, drawn, size, rows_dev=None):
        """The device half of sample(): gather the Batch of a draw."""
        b = self._gather_rows(drawn, size, rows_dev)
        return DeviceBatch(size, {"observation": b["state"]}, {"observation": b["next_state"]},
                           b["action"], b["reward"], b["game_over"],
                           info={"logical_idx": drawn, "states_pair": b["states_pair"]})

    def sample(self, size):
        """ExperienceReplay.sample (:71-90) -> DeviceBatch (the Batch the agent would build)."""
        return self.collate(self.draw(size), size)

    def get(self, index):
        b = self.gather(self.physical_rows([index]), 1)
        return {k: v.clone() for k, v in b.items() if k not in ("rows", "states_pair")}

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise IndexError("replay kernel reported an out-of-range row (status bits %d)" % s)
