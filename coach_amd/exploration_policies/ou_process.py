"""Ornstein-Uhlenbeck action noise for N lockstep envs — mirror of
rl_coach/exploration_policies/ou_process.py (OUProcessParameters :28-38, noise :61-65,
get_action :67-73): dx = theta (mu - x) dt + sigma sqrt(dt) N(0,1); action = mean + x.
The O(n_env * A) state update stays on the host (fp64, the reference's arithmetic and RNG stream);
the device adds it to the policy output (rlx_gaussian_action with std = 1, z = noise; no clipping: the environment
clips what it executes, environments/environment.py:283, the transition keeps the noisy action, agent.py:935)."""
import numpy as np
import torch

from .. import _rlx
from ..core_types import RunPhase


class OUProcessParameters(object):                       # ou_process.py:28-38
    def __init__(self):
        self.mu = 0
        self.theta = 0.15
        self.sigma = 0.2
        self.dt = 0.01

    @property
    def path(self):
        return 'coach_amd.exploration_policies.ou_process:OUProcess'


class OUProcess(object):
    def __init__(self, action_low, action_high, n_env, device, params):
        self.low = np.asarray(action_low, dtype=np.float32)
        self.high = np.asarray(action_high, dtype=np.float32)
        self.A, self.n_env, self.device = self.low.size, n_env, device
        self.mu = float(params.mu) * np.ones(self.A)
        self.theta = float(params.theta)
        self.sigma = float(params.sigma) * np.ones(self.A)
        self.dt = params.dt
        self.state = np.zeros((n_env, self.A))
        self.phase = RunPhase.HEATUP
        self.lib = _rlx.lib()
        self.d_low = torch.from_numpy(self.low).to(device)
        self.d_high = torch.from_numpy(self.high).to(device)
        self.d_one = torch.ones(self.A, dtype=torch.float32, device=device)
        from ..staging import Stager
        self._z = Stager((n_env, self.A), torch.float64, device)
        self.d_z = self._z.dst

    def reset(self, envs=None):
        """envs: indices of the envs whose episode ended (every env's process restarts when None)."""
        if envs is None or not hasattr(self, "state"):
            self.state = np.zeros((self.n_env, self.A))
        else:
            self.state[np.asarray(envs)] = 0.0

    def noise(self):
        for e in range(self.n_env):                                           # :61-65 per env
            x = self.state[e]
            dx = self.theta * (self.mu - x) * self.dt + self.sigma * np.random.randn(self.A) * np.sqrt(self.dt)
            self.state[e] = x + dx
        return self.state

    def get_action(self, action_means, out_actions):
        noise = self.noise() if self.phase == RunPhase.TRAIN else np.zeros((self.n_env, self.A))
        self._z.push(noise)
        self.lib.gaussian_action(action_means, self.d_one, None, self.d_z, None, None,
                                 self.n_env, self.A, out_actions, _rlx.current_stream())
        return out_actions
