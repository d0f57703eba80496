"""Gaussian action noise for N lockstep envs — mirror of
rl_coach/exploration_policies/additive_noise.py (AdditiveNoiseParameters :28-38, get_action :75-111).

``np.random.normal(mean, std)`` of the legacy generator is mean + std * z with z the next values of
its standard-normal stream, so the host draws z [n_env, A] in env order (identical to n_env
sequential get_action calls) and the device forms the actions (rlx_gaussian_action).  They are NOT clipped here: the
reference's transition records the noisy action as the policy returned it (agent.py:854,935) and it is the
ENVIRONMENT that clips what it executes (environments/environment.py:283, gym_environment.py:434).
"""
import numpy as np
import torch

from .. import _rlx
from ..core_types import RunPhase
from ..schedules import LinearSchedule


class AdditiveNoiseParameters(object):                   # additive_noise.py:28-38
    def __init__(self):
        self.noise_schedule = LinearSchedule(0.1, 0.1, 50000)
        self.evaluation_noise = 0.05
        self.noise_as_percentage_from_action_space = True

    @property
    def path(self):
        return 'coach_amd.exploration_policies.additive_noise:AdditiveNoise'


class AdditiveNoise(object):
    def __init__(self, action_low, action_high, n_env, device, params):
        self.low = np.asarray(action_low, dtype=np.float32)
        self.high = np.asarray(action_high, dtype=np.float32)
        self.A, self.n_env, self.device = self.low.size, n_env, device
        self.noise_schedule = params.noise_schedule
        self.evaluation_noise = params.evaluation_noise
        self.noise_as_percentage_from_action_space = params.noise_as_percentage_from_action_space
        self.phase = RunPhase.HEATUP
        self.lib = _rlx.lib()
        self.d_low = torch.from_numpy(self.low).to(device)
        self.d_high = torch.from_numpy(self.high).to(device)
        self.d_std = torch.zeros(self.A, dtype=torch.float32, device=device)
        from ..staging import Stager
        self._z = Stager((n_env, self.A), torch.float64, device)
        self.d_z = self._z.dst

    def current_std(self):
        noise = self.evaluation_noise if self.phase == RunPhase.TEST else self.noise_schedule.current_value
        if self.noise_as_percentage_from_action_space:                        # :86-89
            return (noise * (self.high - self.low)).astype(np.float32)
        return np.full(self.A, noise, dtype=np.float32)

    def get_action(self, action_means, out_actions):
        """action_means: device fp32 [n_env, A]."""
        std = self.current_std()
        if self.phase != RunPhase.TEST:
            for _ in range(self.n_env):
                self.noise_schedule.step()                                    # :99-100
            z = np.random.standard_normal((self.n_env, self.A))              # np.random.normal (:106)
        else:
            z = np.zeros((self.n_env, self.A))
        self._z.push(z)
        self.d_std.copy_(torch.from_numpy(std), non_blocking=True)
        self.lib.gaussian_action(action_means, self.d_std, None, self.d_z, None, None,
                                 self.n_env, self.A, out_actions, _rlx.current_stream())
        return out_actions
