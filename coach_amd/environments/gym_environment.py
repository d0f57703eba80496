"""`GymVectorEnvironment(level=...)` / `Atari(level=...)` — the environment parameter classes reference presets
instantiate (rl_coach/environments/gym_environment.py:62-145) and the level tables they index (mujoco_v2,
atari_deterministic_v4, atari_schedule :84-145).

gym, ALE and MuJoCo cannot be installed in this image (no network), so `create(...)` resolves a level to its
SPACES — observation shape, action space, episode time limit, as gym registers them — and builds the
device-resident synthetic environment with those spaces (coach_amd/environments/synthetic_vector_environment.py;
SURVEY.md §8(d) "synthetic fixed-length episodes") — except CartPole-v0 / -v1, whose simulator is restated on the device
(coach_amd/environments/cartpole_vector_environment.py, gym 0.12.5's physics bit for bit).  The agent, memory, filter and schedule side of a preset is
exactly what would run against the real simulator; swapping the simulator in is the (f)3 front end
(coach_amd/environments/emulator_frontend.py) once gym exists.  An unknown level raises instead of guessing.
"""
from ..core_types import EnvironmentEpisodes, EnvironmentSteps
from ..graph_managers.basic_rl_graph_manager import ScheduleParameters
from .environment import EnvironmentParameters, LevelSelection, SingleLevelSelection  # noqa: F401


def lower_under_to_upper(s):
    s = s.replace('_', ' ')
    s = s.title()
    return s.replace(' ', '')


# ---- level tables (names only; gym_environment.py:84-100,131-145)
gym_mujoco_envs = ['inverted_pendulum', 'inverted_double_pendulum', 'reacher', 'hopper', 'walker2d', 'half_cheetah',
                   'ant', 'swimmer', 'humanoid', 'humanoid_standup', 'pusher', 'thrower', 'striker']
mujoco_v2 = {e: "{}".format(lower_under_to_upper(e) + '-v2') for e in gym_mujoco_envs}
mujoco_v2['walker2d'] = 'Walker2d-v2'

gym_atari_envs = ['air_raid', 'alien', 'amidar', 'assault', 'asterix', 'asteroids', 'atlantis', 'bank_heist',
                  'battle_zone', 'beam_rider', 'berzerk', 'bowling', 'boxing', 'breakout', 'carnival', 'centipede',
                  'chopper_command', 'crazy_climber', 'demon_attack', 'double_dunk', 'elevator_action', 'enduro',
                  'fishing_derby', 'freeway', 'frostbite', 'gopher', 'gravitar', 'hero', 'ice_hockey', 'jamesbond',
                  'journey_escape', 'kangaroo', 'krull', 'kung_fu_master', 'montezuma_revenge', 'ms_pacman',
                  'name_this_game', 'phoenix', 'pitfall', 'pong', 'pooyan', 'private_eye', 'qbert', 'riverraid',
                  'road_runner', 'robotank', 'seaquest', 'skiing', 'solaris', 'space_invaders', 'star_gunner',
                  'tennis', 'time_pilot', 'tutankham', 'up_n_down', 'venture', 'video_pinball', 'wizard_of_wor',
                  'yars_revenge', 'zaxxon']
atari_deterministic_v4 = {e: "{}".format(lower_under_to_upper(e) + 'Deterministic-v4') for e in gym_atari_envs}
atari_no_frameskip_v4 = {e: "{}".format(lower_under_to_upper(e) + 'NoFrameskip-v4') for e in gym_atari_envs}

# the Atari schedule every Atari preset shares (gym_environment.py:138-144)
atari_schedule = ScheduleParameters()
atari_schedule.improve_steps = EnvironmentSteps(50000000)
atari_schedule.steps_between_evaluation_periods = EnvironmentSteps(250000)
atari_schedule.evaluation_steps = EnvironmentSteps(135000)
atari_schedule.heatup_steps = EnvironmentSteps(50000)

# ---- spaces of the levels, as gym registers them: (observation dim, action dim | None, discrete actions | None,
#      max_episode_steps).  Minimal-action-set sizes for the Atari games the reference's trace tests name.
_VECTOR_LEVELS = {
    'CartPole-v0': (4, None, 2, 200), 'CartPole-v1': (4, None, 2, 500), 'MountainCar-v0': (2, None, 3, 200),
    'Acrobot-v1': (6, None, 3, 500), 'LunarLander-v2': (8, None, 4, 1000), 'Pendulum-v0': (3, 1, None, 200),
    'InvertedPendulum-v2': (4, 1, None, 1000), 'InvertedDoublePendulum-v2': (11, 1, None, 1000),
    'Reacher-v2': (11, 2, None, 50), 'Hopper-v2': (11, 3, None, 1000), 'Walker2d-v2': (17, 6, None, 1000),
    'HalfCheetah-v2': (17, 6, None, 1000), 'Ant-v2': (111, 8, None, 1000), 'Swimmer-v2': (8, 2, None, 1000),
    'Humanoid-v2': (376, 17, None, 1000), 'HumanoidStandup-v2': (376, 17, None, 1000),
}
_ATARI_ACTIONS = {'breakout': 4, 'pong': 6, 'space_invaders': 6, 'seaquest': 18, 'qbert': 6, 'beam_rider': 9,
                  'enduro': 9, 'ms_pacman': 9, 'asterix': 9, 'boxing': 18, 'freeway': 3}


class GymEnvironmentParameters(EnvironmentParameters):
    def __init__(self, level=None):
        super().__init__(level=level)
        self.random_initialization_steps = 0
        self.max_over_num_frames = 1
        self.additional_simulator_parameters = {}
        self.observation_space_type = None
        # device-engine additions: envs stepped in lockstep per GPU, episode length override, env seed
        self.num_envs = 1
        self.episode_length = None
        self.seed = 1234

    @property
    def path(self):
        return 'coach_amd.environments.gym_environment:create'

    def level_name(self):
        lvl = self.level
        return str(lvl) if isinstance(lvl, LevelSelection) else lvl


class GymVectorEnvironment(GymEnvironmentParameters):       # :76-81 — vector observations, no default filters
    def __init__(self, level=None):
        super().__init__(level=level)
        self.frame_skip = 1
        self.is_atari = False


class Atari(GymEnvironmentParameters):                      # :116-128 — the Atari filter chain is implied
    def __init__(self, level=None):
        super().__init__(level=level)
        self.frame_skip = 4
        self.max_over_num_frames = 2
        self.random_initialization_steps = 30
        self.is_atari = True
        self.num_envs = 64


def vector_parameters(env_params):
    """env_params -> SyntheticVectorEnvironmentParameters with the level's spaces."""
    from .synthetic_vector_environment import SyntheticVectorEnvironmentParameters as P
    name = env_params.level_name()
    if getattr(env_params, "is_atari", False):
        game = next((g for g, n in list(atari_deterministic_v4.items()) + list(atari_no_frameskip_v4.items())
                     if n == name), None)
        if game is None or game not in _ATARI_ACTIONS:
            raise ValueError("no action-set size is tabulated for the Atari level {!r}".format(name))
        # after the Atari input filter chain: 84x84 luminance frames, stack of 4 (gym_environment.py:106-113)
        return P("image", env_params.num_envs, (84, 84), _ATARI_ACTIONS[game],
                 episode_length=env_params.episode_length or 1024, seed=env_params.seed)
    if name not in _VECTOR_LEVELS:
        raise ValueError("the spaces of level {!r} are not tabulated (gym is not installed in this image; known "
                         "levels: {})".format(name, ", ".join(sorted(_VECTOR_LEVELS))))
    obs, act_dim, n_act, limit = _VECTOR_LEVELS[name]
    return P("vector", env_params.num_envs, (obs,), n_act, action_dim=act_dim,
             episode_length=env_params.episode_length or limit, seed=env_params.seed)


def create(env_params, device, rank=0):
    """The `path` target of the parameter classes above: build the environment of a preset on `device`.  CartPole is
    the one level whose simulator exists on the device (csrc/cartpole.hip); every other level gets the synthetic
    environment with the level's spaces."""
    from .synthetic_vector_environment import SyntheticVectorEnvironment
    name = env_params.level_name()
    if name in ('CartPole-v0', 'CartPole-v1') and not getattr(env_params, "synthetic", False):
        from .cartpole_vector_environment import CartPoleVectorEnvironment, CartPoleVectorEnvironmentParameters
        return CartPoleVectorEnvironment(
            CartPoleVectorEnvironmentParameters(env_params.num_envs, name, env_params.seed, env_params.episode_length),
            device, rank=rank)
    return SyntheticVectorEnvironment(vector_parameters(env_params), device, rank=rank)
