"""Reference presets, TEXT UNCHANGED, against the device engine (BASELINE.json north_star: "keeps the rl_coach
Agent/GraphManager/preset API surface so existing presets drop in").

CPU part (needs /root/reference, i.e. the build container; skipped elsewhere): the source of
rl_coach/presets/<name>.py is exec'd with `rl_coach` aliased to this package (coach_amd.compat.install) and the
resulting graph_manager is checked field by field against what the preset wrote.  Nothing of the reference is
copied into the repository: the texts are read where they lie.
GPU part (/root/reference does not exist on the GPU box): presets WRITTEN FOR THIS TEST in the reference's idiom —
`from rl_coach... import ...`, nested network-wrapper access, filter objects, level selection — are created and run
(heat-up, training, evaluation) on the MI355X: the same import layer and parameter plumbing drive the HIP hot path.
"""
import os

import pytest

REF = "/root/reference/rl_coach/presets"
PRESETS = ["CartPole_DQN", "Mujoco_TD3", "Mujoco_DDPG", "Mujoco_SAC", "Atari_DQN", "Atari_DQN_with_PER",
           "Atari_Dueling_DDQN", "Mujoco_ClippedPPO", "CartPole_ClippedPPO"]
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container)")

_HEAD = """
from rl_coach.base_parameters import VisualizationParameters, PresetValidationParameters, EmbedderScheme, MiddlewareScheme
from rl_coach.core_types import TrainingSteps, EnvironmentEpisodes, EnvironmentSteps
from rl_coach.environments.environment import SingleLevelSelection
from rl_coach.environments.gym_environment import GymVectorEnvironment, Atari, mujoco_v2, atari_deterministic_v4
from rl_coach.graph_managers.basic_rl_graph_manager import BasicRLGraphManager
from rl_coach.graph_managers.graph_manager import ScheduleParameters
from rl_coach.memories.memory import MemoryGranularity
from rl_coach.architectures.layers import Dense
from rl_coach.schedules import LinearSchedule
schedule_params = ScheduleParameters()
schedule_params.improve_steps = EnvironmentSteps(192)
schedule_params.steps_between_evaluation_periods = EnvironmentSteps(96)
schedule_params.evaluation_steps = EnvironmentEpisodes(1)
schedule_params.heatup_steps = EnvironmentSteps(128)
"""
_TAIL = """
env_params.num_envs = 4
env_params.episode_length = 8
for _net in agent_params.network_wrappers.values():
    _net.batch_size = 16
graph_manager = BasicRLGraphManager(agent_params=agent_params, env_params=env_params, schedule_params=schedule_params,
                                    vis_params=VisualizationParameters(), preset_validation_params=PresetValidationParameters())
"""
# presets in the reference's idiom, written for this test (hyper-parameters are this test's own)
DEVICE_PRESETS = {
    "dqn_cartpole": """
from rl_coach.agents.dqn_agent import DQNAgentParameters
agent_params = DQNAgentParameters()
agent_params.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(40)
agent_params.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
agent_params.network_wrappers['main'].learning_rate = 0.0005
agent_params.network_wrappers['main'].replace_mse_with_huber_loss = False
agent_params.network_wrappers['main'].input_embedders_parameters['observation'].scheme = [Dense(96)]
agent_params.network_wrappers['main'].middleware_parameters.scheme = [Dense(96)]
agent_params.memory.max_size = (MemoryGranularity.Transitions, 4096)
agent_params.exploration.epsilon_schedule = LinearSchedule(1.0, 0.05, 300)
env_params = GymVectorEnvironment(level='CartPole-v0')
""",
    "td3_half_cheetah": """
from rl_coach.agents.td3_agent import TD3AgentParameters
agent_params = TD3AgentParameters()
agent_params.network_wrappers['actor'].input_embedders_parameters['observation'].scheme = [Dense(64)]
agent_params.network_wrappers['actor'].middleware_parameters.scheme = [Dense(32)]
agent_params.network_wrappers['critic'].input_embedders_parameters['observation'].scheme = EmbedderScheme.Empty
agent_params.network_wrappers['critic'].input_embedders_parameters['action'].scheme = EmbedderScheme.Empty
agent_params.network_wrappers['critic'].middleware_parameters.scheme = [Dense(64), Dense(32)]
agent_params.memory.max_size = (MemoryGranularity.Transitions, 4096)
env_params = GymVectorEnvironment(level=SingleLevelSelection(mujoco_v2))
env_params.level.select('half_cheetah')
""",
    "sac_hopper": """
from rl_coach.agents.soft_actor_critic_agent import SoftActorCriticAgentParameters
from rl_coach.filters.filter import InputFilter
from rl_coach.filters.reward.reward_rescale_filter import RewardRescaleFilter
agent_params = SoftActorCriticAgentParameters()
agent_params.network_wrappers['v'].middleware_parameters.scheme = [Dense(64)]
agent_params.network_wrappers['q'].heads_parameters[0].network_layers_sizes = (64, 64)
agent_params.network_wrappers['policy'].middleware_parameters.scheme = [Dense(64)]
agent_params.input_filter = InputFilter()
agent_params.input_filter.add_reward_filter('rescale', RewardRescaleFilter(3))
agent_params.memory.max_size = (MemoryGranularity.Transitions, 4096)
env_params = GymVectorEnvironment(level=SingleLevelSelection(mujoco_v2))
env_params.level.select('hopper')
""",
    "dueling_ddqn_pong": """
import math
from rl_coach.agents.ddqn_agent import DDQNAgentParameters
from rl_coach.architectures.head_parameters import DuelingQHeadParameters
from rl_coach.memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplayParameters
agent_params = DDQNAgentParameters()
agent_params.network_wrappers['main'].middleware_parameters.scheme = MiddlewareScheme.Empty
agent_params.network_wrappers['main'].heads_parameters = [DuelingQHeadParameters(rescale_gradient_from_head_by_factor=1/math.sqrt(2))]
agent_params.network_wrappers['main'].clip_gradients = 10
agent_params.memory = PrioritizedExperienceReplayParameters()
agent_params.memory.beta = LinearSchedule(0.4, 1, 1000)
agent_params.memory.max_size = (MemoryGranularity.Transitions, 1024)
env_params = Atari(level=SingleLevelSelection(atari_deterministic_v4))
env_params.level.select('pong')
""",
    "clipped_ppo_hopper": """
from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgentParameters
from rl_coach.filters.filter import InputFilter
from rl_coach.filters.observation.observation_normalization_filter import ObservationNormalizationFilter
agent_params = ClippedPPOAgentParameters()
agent_params.network_wrappers['main'].input_embedders_parameters['observation'].activation_function = 'tanh'
agent_params.network_wrappers['main'].input_embedders_parameters['observation'].scheme = [Dense(32)]
agent_params.network_wrappers['main'].middleware_parameters.scheme = [Dense(32)]
agent_params.algorithm.num_consecutive_playing_steps = EnvironmentSteps(64)
agent_params.algorithm.optimization_epochs = 2
agent_params.algorithm.beta_entropy = 0
agent_params.input_filter = InputFilter()
agent_params.pre_network_filter = InputFilter()
agent_params.pre_network_filter.add_observation_filter('observation', 'normalize_observation',
                                                       ObservationNormalizationFilter(name='normalize_observation'))
env_params = GymVectorEnvironment(level=SingleLevelSelection(mujoco_v2))
env_params.level.select('hopper')
schedule_params.heatup_steps = EnvironmentSteps(0)
""",
}


def _exec_preset(text):
    import coach_amd.compat as compat
    compat.install()
    ns = {"__name__": "preset"}
    exec(compile(text, "<preset>", "exec"), ns)
    return ns


def _text(name):
    return open(os.path.join(REF, name + ".py")).read()


@needs_reference
@pytest.mark.parametrize("name", PRESETS)
def test_reference_preset_text_executes_unchanged(name):
    from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager
    ns = _exec_preset(_text(name))
    gm = ns["graph_manager"]
    assert isinstance(gm, BasicRLGraphManager)
    assert gm.agent_params is ns["agent_params"] and gm.env_params is ns["env_params"]
    assert gm.preset_validation_params is ns["preset_validation_params"]
    assert type(gm.agent_params).__module__.startswith("coach_amd.agents.")


@needs_reference
def test_cartpole_dqn_preset_values_reach_the_device_parameter_objects():
    from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps, TrainingSteps
    from coach_amd.environments.gym_environment import vector_parameters
    from coach_amd.memories.memory import MemoryGranularity
    ns = _exec_preset(_text("CartPole_DQN"))
    ap, gm = ns["agent_params"], ns["graph_manager"]
    assert ap.algorithm.num_steps_between_copying_online_weights_to_target == EnvironmentSteps(100)
    assert ap.algorithm.num_consecutive_playing_steps == EnvironmentSteps(1)
    assert ap.network_wrappers['main'].learning_rate == 0.00025
    assert ap.network_wrappers['main'].replace_mse_with_huber_loss is False
    assert ap.memory.max_size == (MemoryGranularity.Transitions, 40000)
    s = ap.exploration.epsilon_schedule
    assert (s.initial_value, s.final_value, s.decay_steps) == (1.0, 0.01, 10000)
    assert gm.schedule.heatup_steps == EnvironmentSteps(1000)
    assert isinstance(gm.schedule.improve_steps, TrainingSteps)
    assert gm.schedule.steps_between_evaluation_periods == EnvironmentEpisodes(10)
    assert gm.preset_validation_params.min_reward_threshold == 150
    vp = vector_parameters(ns["env_params"])                      # CartPole-v0: 4 observations, 2 actions, 200 steps
    assert (vp.kind, vp.observation_shape, vp.num_actions, vp.episode_length) == ("vector", (4,), 2, 200)


@needs_reference
def test_mujoco_presets_network_schemes_and_level_selection():
    from coach_amd.compat import resolve_reference_style
    from coach_amd.environments.gym_environment import vector_parameters
    ns = _exec_preset(_text("Mujoco_TD3"))
    ap = ns["agent_params"]
    assert ap.network_wrappers['actor'].observation_embedder_scheme == (400,)
    assert ap.network_wrappers['actor'].middleware_scheme == (300,)
    assert ap.network_wrappers['critic'].observation_embedder_scheme == ()
    assert ap.network_wrappers['critic'].action_embedder_scheme == ()
    assert ap.network_wrappers['critic'].middleware_scheme == (400, 300)
    with pytest.raises(ValueError, match="No level has been selected"):
        vector_parameters(ns["env_params"])
    ns["env_params"].level.select('half_cheetah')
    vp = vector_parameters(ns["env_params"])
    assert (vp.observation_shape, vp.action_dim, vp.episode_length) == ((17,), 6, 1000)
    # SAC: reward rescale filter and the Q head sizes
    ns = _exec_preset(_text("Mujoco_SAC"))
    ap = ns["agent_params"]
    ap.algorithm.reward_rescale = 1.0
    resolve_reference_style(ap, ns["env_params"])
    assert ap.algorithm.reward_rescale == 5.0 and ap.algorithm.reward_clipping is None
    assert ap.network_wrappers['q'].network_layers_sizes == (256, 256)
    assert ap.network_wrappers['policy'].middleware_scheme == (256,)
    # Clipped PPO on MuJoCo: tanh 64-64, observation normalisation as the pre-network filter
    ns = _exec_preset(_text("Mujoco_ClippedPPO"))
    ap = ns["agent_params"]
    net = ap.network_wrappers['main']
    assert (net.embedder_scheme, net.middleware_scheme, net.activation_function) == ([64], [64], 'tanh')
    assert net.optimizer_epsilon == 1e-5 and net.adam_optimizer_beta2 == 0.999
    resolve_reference_style(ap, ns["env_params"])
    assert ap.algorithm.normalize_observations is True and ap.algorithm.reward_clipping is None
    # Atari: PER with the annealed beta, reward clipping from the Atari input filter
    ns = _exec_preset(_text("Atari_DQN_with_PER"))
    ap = ns["agent_params"]
    assert type(ap.memory).__name__ == "PrioritizedExperienceReplayParameters"
    assert (ap.memory.beta.initial_value, ap.memory.beta.final_value) == (0.4, 1)
    ns["env_params"].level.select('breakout')
    resolve_reference_style(ap, ns["env_params"])
    assert ap.algorithm.reward_clipping == (-1.0, 1.0)
    vp = vector_parameters(ns["env_params"])
    assert (vp.kind, vp.observation_shape, vp.num_actions) == ("image", (84, 84), 4)


@needs_reference
def test_mujoco_ddpg_preset_with_batchnorm_switch():
    """The reference's Mujoco_DDPG preset with its one-line variation `DDPGAgentParameters(use_batchnorm=True)`
    (agents/ddpg_agent.py:111-122): the switch reaches both device networks, the preset's scheme assignments still apply,
    and the reference's nested access path reads it back."""
    text = _text("Mujoco_DDPG")
    assert "DDPGAgentParameters()" in text
    ap = _exec_preset(text.replace("DDPGAgentParameters()", "DDPGAgentParameters(use_batchnorm=True)"))["agent_params"]
    for name in ("actor", "critic"):
        net = ap.network_wrappers[name]
        assert net.batchnorm is True
        assert net.input_embedders_parameters['observation'].batchnorm is True
        assert net.middleware_parameters.batchnorm is True
        assert net.observation_embedder_scheme == (400,) and net.middleware_scheme == (300,)
    plain = _exec_preset(text)["agent_params"]
    assert plain.network_wrappers["actor"].batchnorm is False and plain.network_wrappers["critic"].batchnorm is False


def test_reference_idiom_presets_build_on_cpu():
    """the GPU part's presets go through the same import layer; building them needs no device"""
    for name, body in DEVICE_PRESETS.items():
        ns = _exec_preset(_HEAD + body + _TAIL)
        assert type(ns["graph_manager"]).__name__ == "BasicRLGraphManager", name


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DEVICE_PRESETS))
def test_reference_idiom_preset_runs_on_the_device(dev, name):
    """heat-up / training / evaluation of a preset written against `rl_coach.*` on the MI355X."""
    import torch
    ns = _exec_preset(_HEAD + DEVICE_PRESETS[name] + _TAIL)
    gm = ns["graph_manager"]
    gm.device = dev
    rows = gm.improve()
    agent = gm.agent
    assert agent.training_iteration > 0
    if "ppo" not in name:
        assert any(r.get("Evaluation Reward", "") != "" for r in rows)
    for net in agent.networks.values():
        assert torch.isfinite(net.params.weights).all()
    if name == "sac_hopper":
        assert agent.ap.algorithm.reward_rescale == 3.0
    if name == "dqn_cartpole":
        assert agent.networks["main"]._fused is not None       # 4 -> 96 -> 96 -> 2 takes the one-launch update
