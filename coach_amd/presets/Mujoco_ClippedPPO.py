"""rl_coach/presets/Mujoco_ClippedPPO.py on the device engine: continuous Clipped-PPO head
(MultivariateNormalDiag with a state-independent log-std), observation normalisation as the
pre-network filter, lr 3e-4, beta_entropy 0, clipping decayed to 0 over 1M steps; the MuJoCo level is
replaced by the synthetic vector environment (HalfCheetah-like shapes)."""
from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgentParameters
from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps
from coach_amd.environments.synthetic_vector_environment import SyntheticVectorEnvironmentParameters
from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters
from coach_amd.schedules import LinearSchedule

schedule_params = ScheduleParameters()
schedule_params.improve_steps = EnvironmentSteps(10 * 2048)
schedule_params.steps_between_evaluation_periods = EnvironmentSteps(2048)
schedule_params.evaluation_steps = EnvironmentEpisodes(0)
schedule_params.heatup_steps = EnvironmentSteps(0)

agent_params = ClippedPPOAgentParameters()
net = agent_params.network_wrappers['main']
net.learning_rate = 0.0003
net.activation_function = 'tanh'
net.embedder_scheme = [64]                      # input_embedders_parameters['observation'].scheme = [Dense(64)]
net.middleware_scheme = [64]                    # middleware_parameters.scheme = [Dense(64)]
net.batch_size = 64
net.optimizer_epsilon = 1e-5
net.adam_optimizer_beta2 = 0.999
agent_params.algorithm.clip_likelihood_ratio_using_epsilon = 0.2
agent_params.algorithm.clipping_decay_schedule = LinearSchedule(1.0, 0, 1000000)
agent_params.algorithm.beta_entropy = 0
agent_params.algorithm.gae_lambda = 0.95
agent_params.algorithm.discount = 0.99
agent_params.algorithm.optimization_epochs = 10
agent_params.algorithm.num_consecutive_playing_steps = EnvironmentSteps(2048)
agent_params.algorithm.reward_clipping = None
agent_params.algorithm.normalize_observations = True

env_params = SyntheticVectorEnvironmentParameters("vector", 64, (17,), None, action_dim=6, episode_length=32,
                                                  seed=1234)

graph_manager = BasicRLGraphManager(agent_params=agent_params, env_params=env_params,
                                    schedule_params=schedule_params)
