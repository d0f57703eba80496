"""CPU restatement of the reference's Atari environment front end — test infrastructure only.

Follows rl_coach/environments/gym_environment.py:
  MaxOverFramesAndFrameskipEnvWrapper.step (:154-175)  frame skip, reward sum, max over the newest frames, early stop
  GymEnvironment._update_state (:418-431)              a lost life ends the episode in HEATUP / TRAIN, fire again in TEST
  GymEnvironment._restart_environment_episode (:459-474) continue after a lost life, else reset; random no-ops, fire
  GymEnvironment._random_noop / _press_fire (:440-457)
and Environment.step / reset_internal_state (environment.py:276-327,340-372).
parity unpinned against gym / ALE themselves (not installable here): pinned to the wrapper code above by reading;
the emulator behind it is any object with reset() / step(a) / lives() / action_meanings()."""
import random

import numpy as np


class FrontEndOracle(object):
    def __init__(self, emulator, frame_skip=4, max_over_num_frames=2, random_initialization_steps=30,
                 max_episode_steps=None, train=True):
        self.env, self.frame_skip, self.max_over = emulator, frame_skip, max_over_num_frames
        self.first_frame_to_max_over = frame_skip - max_over_num_frames
        self.random_initialization_steps = random_initialization_steps
        self.max_episode_steps = max_episode_steps
        self.train = train
        self.elapsed = 0
        self.state = None
        self.done = False
        self.reward = 0.0
        self.lives = None

    # MaxOverFramesAndFrameskipEnvWrapper.step
    def _wrapped_step(self, action):
        total, done, stack = 0.0, None, []
        for i in range(self.frame_skip):
            obs, r, done = self.env.step(action)
            self.elapsed += 1
            if self.max_episode_steps is not None and self.elapsed >= self.max_episode_steps:
                done = True                                  # gym's TimeLimit wrapper
            if i >= self.first_frame_to_max_over:
                stack.append(obs)
            total += r
            if done:
                if not stack:
                    stack.append(obs)
                break
        return np.max(stack, axis=0), total, done

    def step(self, action):                                  # Environment.step + _take_action + _update_state
        self.state, self.reward, self.done = self._wrapped_step(action)
        if self.lives is not None and self.lives != self.env.lives():
            if self.train:
                self.done = True
            elif not self.done:
                self._press_fire()
            self.lives = self.env.lives()
        return self.state, self.reward, self.done

    def _press_fire(self):
        if self.env.action_meanings()[1] == 'FIRE':
            self.lives = self.env.lives()
            self.step(1)
            if self.done:
                self.reset(False)

    def reset(self, force=False):                            # reset_internal_state + _restart_environment_episode
        if self.env.lives() > 0 and not force and \
                (self.max_episode_steps is None or self.elapsed < self.max_episode_steps) and self.state is not None:
            self.step(0)
        else:
            self.state = self.env.reset()
            self.elapsed = 0
            self.lives = self.env.lives()
        step_count = 0                                       # _random_noop
        n = random.randint(0, self.random_initialization_steps)
        while self.state is None or step_count < n:
            step_count += 1
            self.step(0)
        self._press_fire()
        self.lives = self.env.lives()
        self.done, self.reward = False, 0.0
        return self.state
