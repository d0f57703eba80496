"""Real-environment front end: N CPU emulators behind one device-resident vector environment.

What stays on the host is what only the emulator can do — advancing the game.  Everything the reference's
GymEnvironment wraps around `env.step` (rl_coach/environments/gym_environment.py) is done with its semantics, per env:
  * frame skip with reward summing and early stop on `done`, the newest `max_over_num_frames` raw frames kept
    (MaxOverFramesAndFrameskipEnvWrapper.step :154-175);
  * a lost life ends the episode in HEATUP / TRAIN and re-presses FIRE in TEST (`_update_state` :418-431);
  * on reset: continue the game after a lost life or really reset, random 0..30 no-op steps, FIRE (`_restart_
    environment_episode`, `_random_noop`, `_press_fire` :440-474).
The kept raw frames of ALL envs travel in ONE pinned host buffer -> ONE async copy; the maximum over them
(rlx_max_over_frames_u8) and the Atari observation chain (rescale 84x84 -> luminance -> uint8; filters.hip) run on the
device, so the agent's replay receives the same 84x84 uint8 frames the reference's input filter would produce.
Actions come back to the host once per step (the emulators need them: one small D2H copy, the only sync).

Emulator protocol (gym / ALE like): reset() -> frame[H,W,3] u8; step(a) -> (frame, reward, done); lives() -> int;
action_meanings() -> list of str.  gym / ALE are not installable in this image: tests drive the front end with
FakeAtariEmulator below and compare every frame / reward / done with oracle/frontend.py on the same emulator.
"""
import random

import numpy as np
import torch

from .. import _rlx
from ..core_types import RunPhase


class FakeAtariEmulator(object):
    """Deterministic stand-in for an ALE game: frames are a hash of (seed, episode, frame index), a life is lost every
    `life_every` frames, the game is over when no lives are left (or after `game_len` frames)."""

    def __init__(self, seed, shape=(210, 160, 3), lives=3, life_every=37, game_len=400, n_actions=4):
        self.seed, self.shape, self.n_lives, self.life_every, self.game_len = seed, shape, lives, life_every, game_len
        self.n_actions = n_actions
        self.episode, self.t, self._lives = -1, 0, 0

    def _frame(self):
        rng = np.random.RandomState((self.seed * 1000003 + self.episode * 7919 + self.t) % (2 ** 31))
        return rng.randint(0, 256, size=self.shape, dtype=np.uint8)

    def action_meanings(self):
        return ['NOOP', 'FIRE', 'RIGHT', 'LEFT'][:self.n_actions]

    def lives(self):
        return self._lives

    def reset(self):
        self.episode += 1
        self.t, self._lives = 0, self.n_lives
        return self._frame()

    def step(self, action):
        self.t += 1
        if self.t % self.life_every == 0:
            self._lives -= 1
        done = self._lives <= 0 or self.t >= self.game_len
        reward = float(((self.seed + self.episode + self.t + int(action)) % 5) - 2)
        return self._frame(), reward, done


class EmulatorEnvironmentParameters(object):
    def __init__(self, emulators, num_actions, frame_skip=4, max_over_num_frames=2, random_initialization_steps=30,
                 max_episode_steps=None, raw_shape=(210, 160, 3), observation_shape=(84, 84), episode_length=None):
        self.emulators = list(emulators)
        self.kind, self.num_envs = "image", len(self.emulators)
        self.observation_shape, self.raw_shape = tuple(observation_shape), tuple(raw_shape)
        self.num_actions, self.action_dim = num_actions, None
        self.action_low, self.action_high = -1.0, 1.0
        self.frame_skip, self.max_over_num_frames = frame_skip, max_over_num_frames
        self.random_initialization_steps = random_initialization_steps
        self.max_episode_steps = max_episode_steps
        # upper bound of agent steps per episode (sizes the replay's frame ring / episode tables)
        self.episode_length = episode_length or (max_episode_steps or 108000) // frame_skip + 1
        self.min_episode_length = 1          # a life can be lost on the first step: the frame ring budgets for it
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.environments.emulator_frontend:EmulatorVectorEnvironment'


class _Game(object):
    """Host-side state of one emulator: the reference's per-environment bookkeeping."""

    def __init__(self, emu, p):
        self.emu, self.p = emu, p
        self.first_kept = p.frame_skip - p.max_over_num_frames
        self.elapsed, self.lives, self.started = 0, None, False
        self.done, self.reward = False, 0.0

    def wrapped_step(self, action, kept):
        """frame skip; `kept` [K,H,W,3] receives the newest frames (repeated when fewer were produced)."""
        p, total, done, n = self.p, 0.0, None, 0
        for i in range(p.frame_skip):
            obs, r, done = self.emu.step(action)
            self.elapsed += 1
            if p.max_episode_steps is not None and self.elapsed >= p.max_episode_steps:
                done = True
            if i >= self.first_kept:
                kept[n] = obs
                n += 1
            total += r
            if done:
                if n == 0:
                    kept[0] = obs
                    n = 1
                break
        for k in range(n, kept.shape[0]):
            kept[k] = kept[n - 1]
        return total, done

    def step(self, action, kept, train):
        self.reward, self.done = self.wrapped_step(action, kept)
        if self.lives is not None and self.lives != self.emu.lives():
            if train:
                self.done = True
            elif not self.done:
                self.press_fire(kept, train)
            self.lives = self.emu.lives()
        self.started = True

    def press_fire(self, kept, train):
        if self.emu.action_meanings()[1] == 'FIRE':
            self.lives = self.emu.lives()
            self.step(1, kept, train)
            if self.done:
                self.reset(kept, train, False)

    def reset(self, kept, train, force=False):
        p = self.p
        if self.emu.lives() > 0 and not force and self.started and \
                (p.max_episode_steps is None or self.elapsed < p.max_episode_steps):
            self.step(0, kept, train)
        else:
            kept[:] = self.emu.reset()
            self.elapsed, self.started = 0, True
            self.lives = self.emu.lives()
        n = random.randint(0, p.random_initialization_steps)
        for _ in range(n):
            self.step(0, kept, train)
        self.press_fire(kept, train)
        self.lives = self.emu.lives()
        self.done, self.reward = False, 0.0


class EmulatorVectorEnvironment(object):
    def __init__(self, params, device, rank=0):
        self.p, self.device = params, device
        self.lib = _rlx.lib()
        self.n = params.num_envs
        self.games = [_Game(e, params) for e in params.emulators]
        K, raw = params.max_over_num_frames, params.raw_shape
        self.raw_bytes = int(np.prod(raw))
        if self.raw_bytes % 16:
            raise ValueError("raw frames must be a multiple of 16 bytes")
        # [2 (next | reset)][n][K][H][W][3]: one pinned buffer, one copy per step
        self.host = torch.empty((2, self.n, K) + raw, dtype=torch.uint8, pin_memory=True)
        self.host_np = self.host.numpy()
        self.dev_raw = torch.empty_like(self.host, device=device)
        self.maxed = torch.empty((2, self.n) + raw, dtype=torch.uint8, device=device)
        oh, ow = params.observation_shape
        self.small = torch.empty((2, self.n, oh, ow, raw[2]), dtype=torch.uint8, device=device)
        self.frames = torch.empty((2, self.n, oh, ow), dtype=torch.uint8, device=device)
        self.obs = self.frames[0]                       # after reset_internal_state: the first observations
        self.next_obs, self.reset_obs = self.frames[0], self.frames[1]
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.game_over = torch.zeros(self.n, dtype=torch.uint8, device=device)
        self._rw = torch.empty(self.n, dtype=torch.float32, pin_memory=True)
        self._go = torch.empty(self.n, dtype=torch.uint8, pin_memory=True)
        self.dones_host = np.zeros(self.n, dtype=bool)
        self.phase = RunPhase.HEATUP
        self.total_steps = 0

    # ---- device half: max over the kept frames, then the Atari chain (gym_environment.py:106-113)
    def _process(self, halves):
        s = _rlx.current_stream()
        K, raw = self.p.max_over_num_frames, self.p.raw_shape
        oh, ow = self.p.observation_shape
        for h in halves:
            self.dev_raw[h].copy_(self.host[h], non_blocking=True)
            self.lib.max_over_frames_u8(self.dev_raw[h], self.maxed[h], self.n, K, self.raw_bytes, s)
            self.lib.resize_bilinear_u8(self.maxed[h], self.small[h], self.n, raw[0], raw[1], raw[2], oh, ow, s)
            self.lib.rgb_to_y_u8(self.small[h], self.frames[h], self.n * oh * ow, 0.0, 255.0, s)

    def reset_internal_state(self, force_environment_reset=True):
        train = self.phase != RunPhase.TEST
        torch.cuda.current_stream().synchronize()          # the pinned buffer of the previous copy is free again
        for e, g in enumerate(self.games):
            g.reset(self.host_np[0, e], train, force_environment_reset)
        self.dones_host[:] = False
        self._process((0,))
        return self.frames[0]

    def step(self, actions):
        """actions: device int32[n_env].  -> (next_obs, reset_obs, reward, game_over) device tensors like the synthetic
        environment; `dones_host` says which envs finished (no second sync)."""
        a = actions.cpu().numpy()                           # the one device->host sync of a step
        train = self.phase != RunPhase.TEST
        any_done = False
        for e, g in enumerate(self.games):
            g.step(int(a[e]), self.host_np[0, e], train)
            self._rw[e], self._go[e] = g.reward, int(g.done)
            self.dones_host[e] = g.done
            if g.done:                                      # the next episode's first observation
                any_done = True
                g.reset(self.host_np[1, e], train, False)
        self.reward.copy_(self._rw, non_blocking=True)
        self.game_over.copy_(self._go, non_blocking=True)
        self._process((0, 1) if any_done else (0,))
        return self.frames[0], self.frames[1], self.reward, self.game_over
