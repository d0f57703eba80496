"""CartPole Clipped PPO for the device engine, with the hyper-parameters of rl_coach/presets/CartPole_ClippedPPO.py:
tanh 64-64 value and policy towers, lr 3e-4, Adam beta2 .999 / eps 1e-5, batch 64, 10 epochs per 2048-step rollout,
GAE(.99, .95), no entropy bonus, clip .2 decayed to 0 over 1 M steps, observation normalisation as the pre-network
filter, five evaluation episodes every 2048 env-steps — and its golden test: an averaged evaluation reward of 150
within 400 episodes (presets/CartPole_ClippedPPO.py:66-70).  The level is CartPole-v0 on the device
(coach_amd/environments/cartpole_vector_environment.py: gym 0.12.5's physics)."""
from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgentParameters
from coach_amd.base_parameters import PresetValidationParameters
from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps, TrainingSteps
from coach_amd.environments.cartpole_vector_environment import CartPoleVectorEnvironmentParameters
from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters
from coach_amd.schedules import LinearSchedule

ROLLOUT = 2048


def make(num_envs=1, seed=1234, agent_seed=0):
    """seed: the environments' reset-state streams; agent_seed: the agent's host generators and initial weights
    (the reference's `--seed`, rl_coach/tests/test_golden.py:122 runs its golden tests with 0)."""
    agent = ClippedPPOAgentParameters()
    agent.seed = agent_seed
    net, alg = agent.network_wrappers['main'], agent.algorithm
    for key, value in dict(learning_rate=3e-4, activation_function='tanh', embedder_scheme=[64],
                           middleware_scheme=[64], batch_size=64, optimizer_epsilon=1e-5,
                           adam_optimizer_beta2=0.999).items():
        setattr(net, key, value)
    for key, value in dict(clip_likelihood_ratio_using_epsilon=0.2, beta_entropy=0, gae_lambda=0.95,
                           discount=0.99, optimization_epochs=10, reward_clipping=None,
                           normalize_observations=True).items():
        setattr(alg, key, value)
    alg.clipping_decay_schedule = LinearSchedule(1.0, 0, 1000000)
    alg.num_consecutive_playing_steps = EnvironmentSteps(ROLLOUT)
    env = CartPoleVectorEnvironmentParameters(num_envs, "CartPole-v0", seed=seed)
    sched = ScheduleParameters()
    sched.heatup_steps = EnvironmentSteps(0)
    sched.improve_steps = TrainingSteps(10000000)
    sched.steps_between_evaluation_periods = EnvironmentSteps(ROLLOUT)
    sched.evaluation_steps = EnvironmentEpisodes(5)
    validation = PresetValidationParameters(test=True, min_reward_threshold=150, max_episodes_to_achieve_reward=400)
    return BasicRLGraphManager(agent_params=agent, env_params=env, schedule_params=sched,
                               preset_validation_params=validation)


graph_manager = make()
