"""Atari Clipped-PPO (BASELINE config C2) on the device engine: ClippedPPOAgentParameters defaults
(rl_coach/agents/clipped_ppo_agent.py:41-131), 64 lockstep synthetic Atari-like envs per GPU."""
from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgentParameters
from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps
from coach_amd.environments.synthetic_vector_environment import SyntheticVectorEnvironmentParameters
from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters

schedule_params = ScheduleParameters()
schedule_params.improve_steps = EnvironmentSteps(10 * 2048)
schedule_params.steps_between_evaluation_periods = EnvironmentSteps(2048)
schedule_params.evaluation_steps = EnvironmentEpisodes(0)
schedule_params.heatup_steps = EnvironmentSteps(0)

agent_params = ClippedPPOAgentParameters()
env_params = SyntheticVectorEnvironmentParameters("image", 64, (84, 84), 6, episode_length=32, seed=1234)

graph_manager = BasicRLGraphManager(agent_params=agent_params, env_params=env_params,
                                    schedule_params=schedule_params)
