"""CartPole DQN (BASELINE config C1) for the device engine.

Hyper-parameters are the ones of rl_coach/presets/CartPole_DQN.py (discount .99, target copy every
100 env-steps, one update per env-step, lr 2.5e-4, MSE loss, 40 k-transition uniform replay,
epsilon 1 -> 0.01 over 10 k steps, 1 000 heat-up steps, evaluation of one episode every 10 episodes) and so is
its golden test: an averaged evaluation reward of 150 within 250 episodes (presets/CartPole_DQN.py:47-51).  The level
is CartPole-v0 on the device (coach_amd/environments/cartpole_vector_environment.py: gym 0.12.5's physics).
`make(num_envs=...)` builds the same experiment with more envs per GPU; `make(synthetic=True)` swaps in the
fixed-length synthetic workload of the throughput benchmarks (actions do not influence it: nothing to learn).
"""
from coach_amd.agents.dqn_agent import DQNAgentParameters
from coach_amd.base_parameters import PresetValidationParameters
from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps, TrainingSteps
from coach_amd.environments.cartpole_vector_environment import CartPoleVectorEnvironmentParameters
from coach_amd.environments.synthetic_vector_environment import SyntheticVectorEnvironmentParameters
from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters
from coach_amd.memories.memory import MemoryGranularity
from coach_amd.schedules import LinearSchedule

EPISODE_LENGTH = 200          # CartPole-v0 time limit
HYPER = dict(discount=0.99, target_copy_every=100, env_steps_per_update=1, learning_rate=2.5e-4,
             replay_transitions=40000, epsilon=(1.0, 0.01, 10000), heatup_steps=1000,
             improve_steps=10000, episodes_between_evaluations=10)


def make(num_envs=1, seed=1234, synthetic=False, agent_seed=0, **overrides):
    """seed: the environments' reset-state streams; agent_seed: the agent's host generators and initial weights
    (the reference's `--seed`, rl_coach/tests/test_golden.py:122 runs its golden tests with 0)."""
    h = dict(HYPER, **overrides)
    agent = DQNAgentParameters()
    agent.seed = agent_seed
    alg, net = agent.algorithm, agent.network_wrappers['main']
    alg.discount = h["discount"]
    alg.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(h["target_copy_every"])
    alg.num_consecutive_playing_steps = EnvironmentSteps(h["env_steps_per_update"])
    net.learning_rate = h["learning_rate"]
    net.replace_mse_with_huber_loss = False
    agent.memory.max_size = (MemoryGranularity.Transitions, h["replay_transitions"])
    agent.exploration.epsilon_schedule = LinearSchedule(*h["epsilon"])
    sched = ScheduleParameters()
    sched.heatup_steps = EnvironmentSteps(h["heatup_steps"])
    sched.evaluation_steps = EnvironmentEpisodes(1)
    if synthetic:
        env = SyntheticVectorEnvironmentParameters("vector", num_envs, (4,), 2, episode_length=EPISODE_LENGTH,
                                                   seed=seed)
        sched.improve_steps = EnvironmentSteps(h["improve_steps"])
        sched.steps_between_evaluation_periods = EnvironmentSteps(h["episodes_between_evaluations"] * EPISODE_LENGTH)
        return BasicRLGraphManager(agent_params=agent, env_params=env, schedule_params=sched)
    env = CartPoleVectorEnvironmentParameters(num_envs, "CartPole-v0", seed=seed)
    sched.improve_steps = TrainingSteps(10000000000)
    sched.steps_between_evaluation_periods = EnvironmentEpisodes(h["episodes_between_evaluations"])
    validation = PresetValidationParameters(test=True, min_reward_threshold=150, max_episodes_to_achieve_reward=250)
    return BasicRLGraphManager(agent_params=agent, env_params=env, schedule_params=sched,
                               preset_validation_params=validation)


graph_manager = make()
