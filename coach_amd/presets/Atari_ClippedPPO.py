"""Atari Clipped-PPO (BASELINE config C2) on the device engine: ClippedPPOAgentParameters defaults
(rl_coach/agents/clipped_ppo_agent.py:41-131) — 2048-step rollouts, 10 epochs of minibatches of 64, GAE(.99, .95),
clip .2, entropy bonus .01, reward clipping to [-1, 1] (the Atari input filter, gym_environment.py:106-113) — on 64
lockstep synthetic Atari-like envs per GPU (84x84 uint8 frames, stacks of 4).  The schedule follows the reference's
Clipped-PPO presets (presets/Mujoco_ClippedPPO.py:19-22, CartPole_ClippedPPO.py): an evaluation period after every
2048 environment steps — greedy whole episodes on a scratch frame stack, the rollout untouched
(graph_manager.py:491-523) — and a forced reset at the start of every period (:477)."""
from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgentParameters
from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps
from coach_amd.environments.synthetic_vector_environment import SyntheticVectorEnvironmentParameters
from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters

ROLLOUT = 2048


def make(num_envs=64, episode_length=32, seed=1234, agent_seed=0, playing_steps=ROLLOUT, batch_size=None,
         optimization_epochs=None, improve_steps=10 * ROLLOUT, steps_between_evaluation_periods=ROLLOUT,
         evaluation_episodes=1):
    """evaluation_episodes: EnvironmentEpisodes(n) per evaluation period — every env of the vector plays n episodes."""
    agent = ClippedPPOAgentParameters()
    agent.seed = agent_seed
    agent.algorithm.num_consecutive_playing_steps = EnvironmentSteps(playing_steps)
    if optimization_epochs is not None:
        agent.algorithm.optimization_epochs = optimization_epochs
    if batch_size is not None:
        agent.network_wrappers["main"].batch_size = batch_size
    env = SyntheticVectorEnvironmentParameters("image", num_envs, (84, 84), 6, episode_length=episode_length, seed=seed)
    sched = ScheduleParameters()
    sched.heatup_steps = EnvironmentSteps(0)
    sched.improve_steps = EnvironmentSteps(improve_steps)
    sched.steps_between_evaluation_periods = EnvironmentSteps(steps_between_evaluation_periods)
    sched.evaluation_steps = EnvironmentEpisodes(evaluation_episodes)
    return BasicRLGraphManager(agent_params=agent, env_params=env, schedule_params=sched)


graph_manager = make()
schedule_params, agent_params, env_params = graph_manager.schedule_params, graph_manager.agent_params, graph_manager.env_params
