"""Double DQN — host-side mirror of rl_coach/agents/ddqn_agent.py (DDQNAgentParameters :24-34, DDQNAgent :38-43).

The only difference to DQN is which network picks the action the target network is evaluated at:
``select_actions(next_states, q_st_plus_1) = argmax_a Q_online(s', a)`` (:42-43) instead of ``argmax_a Q_target(s', a)``
(dqn_agent.py:76-77).  Here that is the class flag ``double_dqn`` DQNAgent._learn_device hands to the network's
``learn_from_batch``: the TD-target kernel (`csrc/targets.hip dqn_targets_kernel`) takes the selector's Q values as its
second operand, and the one-launch MLP update (`csrc/mlp_fused.hip`, field ``ddqn``) evaluates the online tower at s' as well.
Parity: tests/test_targets.py (fixtures of the reference's DDQNAgent.learn_from_batch), tests/test_dqn_agent.py[double].
"""
from ..core_types import EnvironmentSteps
from ..schedules import LinearSchedule
from .dqn_agent import DQNAgent, DQNAgentParameters


class DDQNAgentParameters(DQNAgentParameters):           # ddqn_agent.py:24-34
    def __init__(self):
        super().__init__()
        # Double DQN's own defaults: slower target copies, lower final / evaluation epsilon
        self.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(30000)
        self.exploration.epsilon_schedule = LinearSchedule(1, 0.01, 1000000)
        self.exploration.evaluation_epsilon = 0.001

    @property
    def path(self):
        return 'coach_amd.agents.ddqn_agent:DDQNAgent'


class DDQNAgent(DQNAgent):
    """select_actions = argmax of the ONLINE network at s' (ddqn_agent.py:43)."""
    double_dqn = True
