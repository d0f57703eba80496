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
                 action_dim=None, episode_length=32, seed=1234, action_low=-1.0, action_high=1.0,
                 episode_lengths=None):
        """episode_lengths: optional time limit PER ENV (len num_envs, each <= episode_length): the envs of the
        vector then finish their episodes on different steps, as real simulators do."""
        self.kind, self.num_envs, self.observation_shape = kind, num_envs, tuple(observation_shape)
        self.num_actions, self.action_dim = num_actions, action_dim
        self.action_low, self.action_high = action_low, action_high      # BoxActionSpace bounds
        self.episode_length, self.seed = episode_length, seed
        self.episode_lengths = None if episode_lengths is None else [int(x) for x in episode_lengths]
        if self.episode_lengths is not None and (len(self.episode_lengths) != num_envs or
                                                 max(self.episode_lengths) > episode_length or
                                                 min(self.episode_lengths) < 1):
            raise ValueError("episode_lengths needs one value in [1, episode_length] per env")

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
        # host mirror of the episode clocks: which envs finished on the last step is known WITHOUT a device
        # sync (a simulator front end knows it from the emulators; here the time limits are data)
        self.lengths_host = np.full(self.n, params.episode_length, dtype=np.int64) if params.episode_lengths is None \
            else np.asarray(params.episode_lengths, dtype=np.int64)
        self.lengths = None if params.episode_lengths is None else \
            torch.from_numpy(self.lengths_host.astype(np.int32)).to(device)
        self.t_host = np.zeros(self.n, dtype=np.int64)
        self.dones_host = np.zeros(self.n, dtype=bool)

    def reset_internal_state(self):
        self.lib.synth_env_reset(self.kind, self.obs, self.episode, self.step_in_episode, self.n,
                                 self.obs_elems, self.seed, self.env_id0, _rlx.current_stream())
        self.t_host[:] = 0
        self.dones_host[:] = False
        return self.obs

    def step(self, actions=None):
        self.launch_step()
        self.host_tick()
        return self.next_obs, self.reset_obs, self.reward, self.game_over

    def host_tick(self):
        """advance the host mirror of the episode clocks (which envs finished: `dones_host`)."""
        self.t_host += 1
        np.greater_equal(self.t_host, self.lengths_host, out=self.dones_host)
        self.t_host[self.dones_host] = 0

    def launch_step(self):
        """the device half of step(): pure launches on static buffers (hipGraph-capturable)."""
        if self.lengths is None:
            self.lib.synth_env_step(self.kind, self.next_obs, self.reset_obs, self.reward, self.game_over,
                                    self.episode, self.step_in_episode, self.n, self.obs_elems,
                                    self.p.episode_length, self.seed, self.env_id0, _rlx.current_stream())
        else:
            self.lib.synth_env_step_lengths(self.kind, self.next_obs, self.reset_obs, self.reward, self.game_over,
                                            self.episode, self.step_in_episode, self.n, self.obs_elems,
                                            self.lengths, self.seed, self.env_id0, _rlx.current_stream())
