"""Device-resident vector of synthetic environments — the ``Environment`` plug point
(rl_coach/environments/environment.py:276-327, environment_interface.py:23-75) for the BASELINE
workloads.  One instance = the env vector owned by ONE GPU (one rank); envs are sharded across
ranks by `env_id0 = rank * num_envs` with no cross-rank traffic.

Observations / rewards come from rlx_synth_env_* (Philox4x32-10, coach_amd/csrc/synth_env.hip);
the CPU twin is oracle/synth_env.py.
"""
import numpy as np
import torch

from .. import _rlx


class SyntheticVectorEnvironmentParameters(object):
    def __init__(self, kind="image", num_envs=64, observation_shape=(84, 84), num_actions=6,
                 action_dim=None, episode_length=32, seed=1234, action_low=-1.0, action_high=1.0):
        self.kind, self.num_envs, self.observation_shape = kind, num_envs, tuple(observation_shape)
        self.num_actions, self.action_dim = num_actions, action_dim
        self.action_low, self.action_high = action_low, action_high      # BoxActionSpace bounds
        self.episode_length, self.seed = episode_length, seed

    @property
    def path(self):
        return 'coach_amd.environments.synthetic_vector_environment:SyntheticVectorEnvironment'


class SyntheticVectorEnvironment(object):
    """step(actions) consumes the action tensor (the synthetic dynamics ignore it) and fills
    next_obs / reset_obs / reward / game_over for all envs with one launch."""

    def __init__(self, params, device, rank=0):
        self.p, self.device = params, device
        self.lib = _rlx.lib()
        self.kind = 0 if params.kind == "image" else 1
        self.n = params.num_envs
        self.obs_elems = int(np.prod(params.observation_shape))
        dt = torch.uint8 if self.kind == 0 else torch.float32
        shp = (self.n,) + params.observation_shape
        self.obs = torch.empty(shp, dtype=dt, device=device)            # O(ep, 0) after reset
        self.next_obs = torch.empty(shp, dtype=dt, device=device)
        self.reset_obs = torch.empty(shp, dtype=dt, device=device)
        self.reward = torch.empty(self.n, dtype=torch.float32, device=device)
        self.game_over = torch.empty(self.n, dtype=torch.uint8, device=device)
        self.episode = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.step_in_episode = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.seed = params.seed
        self.env_id0 = rank * self.n
        self.total_steps = 0

    def reset_internal_state(self):
        self.lib.synth_env_reset(self.kind, self.obs, self.episode, self.step_in_episode, self.n,
                                 self.obs_elems, self.seed, self.env_id0, _rlx.current_stream())
        return self.obs

    def step(self, actions=None):
        self.lib.synth_env_step(self.kind, self.next_obs, self.reset_obs, self.reward, self.game_over,
                                self.episode, self.step_in_episode, self.n, self.obs_elems,
                                self.p.episode_length, self.seed, self.env_id0, _rlx.current_stream())
        return self.next_obs, self.reset_obs, self.reward, self.game_over
