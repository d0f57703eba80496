"""DQN / Double DQN on one MI355X — host-side mirror of rl_coach/agents/dqn_agent.py (parameter
classes :33-65, DQNAgent.learn_from_batch :81-113), rl_coach/agents/ddqn_agent.py:43 and the
ValueOptimizationAgent pieces it uses (agents/value_optimization_agent.py: choose_action through
the exploration policy, update_transition_priorities_and_get_weights).

Per vector step: stacked states -> online Q network -> epsilon-greedy (host draws, device argmax /
tie-break) -> env.step -> reward filter -> episode statistics -> replay store.
Per update: replay draw (host RNG, reference order) -> device gather -> target-network and online
forward -> rlx_dqn_targets (TD targets + |TD error|) -> Huber/MSE loss -> backward -> TF1 Adam;
prioritized replay gets its priorities from the device TD errors without a host round trip.
"""
import numpy as np
import torch

from .. import _rlx
from ..architectures.head_parameters import DuelingQHeadParameters, QHeadParameters
from ..architectures.scheme_views import SchemeViews
from ..core_types import EnvironmentSteps, RunPhase
from ..exploration_policies.e_greedy import EGreedy, EGreedyParameters
from ..memories.non_episodic.experience_replay import ExperienceReplayParameters
from ..memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplay
from ..nn.networks import DQNNet
from ..schedules import LinearSchedule
from .vector_agent import AlgorithmParameters, VectorOffPolicyAgent


class DQNAlgorithmParameters(AlgorithmParameters):       # dqn_agent.py:33-40
    def __init__(self):
        super().__init__()
        self.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(10000)
        self.num_consecutive_playing_steps = EnvironmentSteps(4)
        self.discount = 0.99
        self.supports_parameter_noise = True


class DQNNetworkParameters(SchemeViews):                      # dqn_agent.py:43-53 + NetworkParameters defaults
    def __init__(self):
        self.activation_function = 'relu'
        self.embedder_scheme = 'Medium'
        self.middleware_scheme = 'Medium'
        self.optimizer_type = 'Adam'
        self.batch_size = 32
        self.replace_mse_with_huber_loss = True
        self.create_target_network = True
        self.learning_rate = 0.00025
        self.adam_optimizer_beta1 = 0.9
        self.adam_optimizer_beta2 = 0.99
        self.optimizer_epsilon = 0.0001
        self.scale_down_gradients_by_number_of_workers_for_sync_training = True
        self.heads_parameters = [QHeadParameters()]      # [DuelingQHeadParameters()] for dueling DQN
        self.clip_gradients = None                       # ClipByGlobalNorm threshold (base_parameters.py)


class DQNAgentParameters(object):                        # dqn_agent.py:56-66
    def __init__(self):
        self.algorithm = DQNAlgorithmParameters()
        self.exploration = EGreedyParameters()
        self.memory = ExperienceReplayParameters()
        self.network_wrappers = {"main": DQNNetworkParameters()}
        self.exploration.epsilon_schedule = LinearSchedule(1, 0.1, 1000000)
        self.exploration.evaluation_epsilon = 0.05
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.dqn_agent:DQNAgent'


class DQNAgent(VectorOffPolicyAgent):
    double_dqn = False
    SIGNAL_NAMES = VectorOffPolicyAgent.SIGNAL_NAMES + ["Q"]            # value_optimization_agent.py:36

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        super().__init__(agent_parameters, environment, device, dist, use_graphs)
        ep, net = environment.p, self.ap.network_wrappers["main"]
        self.A = ep.num_actions
        self.batch_size = net.batch_size
        obs_shape = tuple(ep.observation_shape) + (self.stack,) if self.image else tuple(ep.observation_shape)
        self.networks = {"main": DQNNet(
            self.device, obs_shape, self.A, activation=net.activation_function,
            embedder=net.embedder_scheme, middleware=net.middleware_scheme,
            learning_rate=net.learning_rate, adam_beta1=net.adam_optimizer_beta1,
            adam_beta2=net.adam_optimizer_beta2, optimizer_epsilon=net.optimizer_epsilon,
            replace_mse_with_huber_loss=net.replace_mse_with_huber_loss, seed=self.ap.seed or 0,
            dueling=isinstance(net.heads_parameters[0], DuelingQHeadParameters),
            head_activation=net.heads_parameters[0].activation_function,
            head_gradient_rescale=net.heads_parameters[0].rescale_gradient_from_head_by_factor,
            clip_gradients=net.clip_gradients)}
        self.memory = self._make_memory(action_dim=None)
        self.exploration_policy = EGreedy(self.A, self.n_env, self.device, self.ap.exploration)
        self.actions = torch.zeros(self.n_env, dtype=torch.int32, device=self.device)
        self.td_errors = torch.zeros(self.batch_size, dtype=torch.float64, device=self.device)
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._finish_init()

    # --------------------------------------------------------------------------------- acting
    def random_actions(self):
        """spaces.DiscreteActionSpace.sample (spaces.py:406-407): np.random.choice(actions) per env."""
        a = np.array([np.random.choice(self.A) for _ in range(self.n_env)], dtype=np.int32)
        self.actions.copy_(self._to_device("rand_act", a, torch.int32))
        return self.actions

    def choose_action(self, states):
        net = self.networks["main"]
        self.exploration_policy.phase = self.phase
        draws = self.exploration_policy.draw()                       # host RNG, per env, in order
        self._run(("q", self.n_env), lambda: self._q_forward(states))
        self.exploration_policy.get_action(self._q_act, draws, self.actions)
        return self.actions

    def _q_forward(self, states):
        q = self.networks["main"].q_values(states, self.n_env, tag="act")
        self._q_act = q.data.view(self.n_env, self.A)

    # ------------------------------------------------------------------------------- training
    def _learn_device(self, b, weights):
        net = self.networks["main"]
        net.learn_from_batch(b._states["observation"], b._next_states["observation"], self.batch_size,
                             b.actions(), b.rewards(), b.game_overs(), self.ap.algorithm.discount,
                             importance_weights=weights, td_errors=self.td_errors,
                             double_dqn=self.double_dqn, grad_scale=self._grad_scale(),
                             sync=self if self.dist is not None else None, states_pair=b._info.get("states_pair"))

    def _grad_scale(self):
        netp = self.ap.network_wrappers["main"]
        return self.dist.grad_scale(netp.scale_down_gradients_by_number_of_workers_for_sync_training) \
            if self.dist else 1.0

    def learn_from_batch(self, batch):
        """DQNAgent.learn_from_batch (dqn_agent.py:81-113)."""
        per = isinstance(self.memory, PrioritizedExperienceReplay)
        weights = batch.info("weight") if per else None           # fp64, as rlx_per_sample wrote them
        self._run(("learn", per), lambda: self._learn_device(batch, weights))
        if per:                                   # update_transition_priorities_and_get_weights
            self.memory.update_priorities(batch.info("idx"), self.td_errors)
        loss = self.networks["main"].loss
        self.signals = {"Loss": loss, "Grads (unclipped)": self.networks["main"].norm}
        return loss


class DDQNAgentParameters(DQNAgentParameters):           # ddqn_agent.py:24-34
    def __init__(self):
        super().__init__()
        # Double DQN's own defaults: slower target copies, lower final / evaluation epsilon
        self.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(30000)
        self.exploration.epsilon_schedule = LinearSchedule(1, 0.01, 1000000)
        self.exploration.evaluation_epsilon = 0.001

    @property
    def path(self):
        return 'coach_amd.agents.dqn_agent:DDQNAgent'


class DDQNAgent(DQNAgent):
    """select_actions = argmax of the ONLINE network at s' (ddqn_agent.py:43)."""
    double_dqn = True
