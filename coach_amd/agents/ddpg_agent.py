"""DDPG on one MI355X — host-side mirror of rl_coach/agents/ddpg_agent.py (parameter classes
:36-122, DDPGAgent.learn_from_batch :137-195, choose_action :200-230).

One update = 5 network passes in the reference (5 sess.run with numpy round trips); here they are
device launches on static buffers, captured into one hipGraph:
  a' = mu_target(s'), a_mu = mu_online(s); y = r + (1-done) gamma Q_target(s', a') (optional clip);
  g = d mean(Q_online(s, a_mu)) / d a  (BEFORE the critic update, :169-173);
  critic: MSE(y, Q_online(s, a)) -> Adam;   actor: d sum(mu(s) * (-g)) / d theta -> Adam.
"""
from collections import OrderedDict

import numpy as np
import torch

from .. import _rlx
from ..core_types import EnvironmentSteps, RunPhase
from ..exploration_policies.ou_process import OUProcess, OUProcessParameters
from ..exploration_policies.additive_noise import AdditiveNoise, AdditiveNoiseParameters
from ..memories.episodic.episodic_experience_replay import EpisodicExperienceReplayParameters
from ..nn.actor_critic_nets import ActorNet, CriticNet
from ..architectures.scheme_views import SchemeViews
from .vector_agent import AlgorithmParameters, VectorOffPolicyAgent


class DDPGCriticNetworkParameters(SchemeViews):           # ddpg_agent.py:36-52 (+ Mujoco_DDPG preset schemes)
    _EMBEDDER_FIELDS = {"observation": "observation_embedder_scheme", "action": "action_embedder_scheme"}
    _TUPLE_SCHEMES = True

    def __init__(self, use_batchnorm=False):
        self.batchnorm = bool(use_batchnorm)             # observation embedder + middleware (ddpg_agent.py:37-45)
        self.observation_embedder_scheme = (400,)
        self.action_embedder_scheme = ()
        self.middleware_scheme = (300,)
        self.num_streams = 1
        self.head_initializer = "normalized_columns"
        self.activation_function = 'relu'
        self.optimizer_type = 'Adam'
        self.batch_size = 64
        self.learning_rate = 0.001
        self.adam_optimizer_beta1 = 0.9
        self.adam_optimizer_beta2 = 0.999
        self.optimizer_epsilon = 1e-8
        self.create_target_network = True
        self.scale_down_gradients_by_number_of_workers_for_sync_training = False


class DDPGActorNetworkParameters(SchemeViews):            # ddpg_agent.py:55-70
    _EMBEDDER_FIELDS = {"observation": "observation_embedder_scheme"}
    _TUPLE_SCHEMES = True

    def __init__(self, use_batchnorm=False):
        self.batchnorm = bool(use_batchnorm)             # embedder, middleware and the actor head (ddpg_agent.py:56-60)
        self.observation_embedder_scheme = (400,)
        self.middleware_scheme = (300,)
        self.activation_function = 'relu'
        self.optimizer_type = 'Adam'
        self.batch_size = 64
        self.learning_rate = 0.0001
        self.adam_optimizer_beta1 = 0.9
        self.adam_optimizer_beta2 = 0.999
        self.optimizer_epsilon = 1e-8
        self.create_target_network = True
        self.scale_down_gradients_by_number_of_workers_for_sync_training = False


class DDPGAlgorithmParameters(AlgorithmParameters):      # ddpg_agent.py:73-111
    def __init__(self):
        super().__init__()
        self.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(1)
        self.rate_for_copying_weights_to_target = 0.001
        self.num_consecutive_playing_steps = EnvironmentSteps(1)
        self.use_target_network_for_evaluation = False
        self.action_penalty = 0
        self.clip_critic_targets = None
        self.use_non_zero_discount_for_terminal_states = False


class DDPGAgentParameters(object):                       # ddpg_agent.py:111-122
    def __init__(self, use_batchnorm=False):
        self.algorithm = DDPGAlgorithmParameters()
        self.exploration = OUProcessParameters()
        self.memory = EpisodicExperienceReplayParameters()
        self.network_wrappers = OrderedDict([("actor", DDPGActorNetworkParameters(use_batchnorm=use_batchnorm)),
                                             ("critic", DDPGCriticNetworkParameters(use_batchnorm=use_batchnorm))])
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.ddpg_agent:DDPGAgent'


class DDPGAgent(VectorOffPolicyAgent):
    continuous = True
    SIGNAL_NAMES = VectorOffPolicyAgent.SIGNAL_NAMES + ["Q", "TD targets", "actions"]   # ddpg_agent.py:129-131

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        super().__init__(agent_parameters, environment, device, dist, use_graphs)
        if self.image:
            raise ValueError("DDPG works only for continuous control problems (vector observations)")
        ep = environment.p
        an, cn = self.ap.network_wrappers["actor"], self.ap.network_wrappers["critic"]
        self.obs_dim, self.A = int(ep.observation_shape[0]), int(ep.action_dim)
        self.batch_size = cn.batch_size
        self.low = np.broadcast_to(np.asarray(ep.action_low, dtype=np.float32), (self.A,)).copy()
        self.high = np.broadcast_to(np.asarray(ep.action_high, dtype=np.float32), (self.A,)).copy()
        scale = float(np.maximum(np.abs(self.low), np.abs(self.high)).max())     # max_abs_range
        seed = self.ap.seed or 0
        actor = ActorNet(self.device, self.obs_dim, self.A, scale, an.observation_embedder_scheme,
                         an.middleware_scheme, an.activation_function, an.learning_rate,
                         an.adam_optimizer_beta1, an.adam_optimizer_beta2, an.optimizer_epsilon, seed,
                         batchnorm=bool(getattr(an, "batchnorm", False)))
        critic = CriticNet(self.device, self.obs_dim, self.A, cn.observation_embedder_scheme,
                           cn.middleware_scheme, cn.num_streams, cn.activation_function,
                           cn.head_initializer, cn.learning_rate, cn.adam_optimizer_beta1,
                           cn.adam_optimizer_beta2, cn.optimizer_epsilon, seed + 1,
                           batchnorm=bool(getattr(cn, "batchnorm", False)))
        self.networks = OrderedDict([("actor", actor), ("critic", critic)])
        self.memory = self._make_memory(action_dim=self.A)
        self.exploration_policy = self._make_exploration()
        dev, B = self.device, self.batch_size
        self.actions = torch.zeros(self.n_env, self.A, dtype=torch.float32, device=dev)
        self.d_low = torch.from_numpy(self.low).to(dev)
        self.d_high = torch.from_numpy(self.high).to(dev)
        self.td_targets = torch.zeros(B, dtype=torch.float32, device=dev)
        self.neg_action_grad = torch.zeros(B, self.A, dtype=torch.float32, device=dev)
        self._finish_init()

    def _make_exploration(self):
        p = self.ap.exploration
        cls = OUProcess if isinstance(p, OUProcessParameters) else AdditiveNoise
        return cls(self.low, self.high, self.n_env, self.device, p)

    # --------------------------------------------------------------------------------- acting
    def random_actions(self):
        """BoxActionSpace.sample (spaces.py:151-162): np.random.uniform(low, high, shape) per env."""
        a = np.random.uniform(self.low, self.high, (self.n_env, self.A)).astype(np.float32)
        self.actions.copy_(self._to_device("rand_act", a, torch.float32))
        return self.actions

    def choose_action(self, states):
        alg = self.ap.algorithm
        self.exploration_policy.phase = self.phase
        use_t = bool(getattr(alg, "use_target_network_for_evaluation", False))
        self._run(("mu", use_t), lambda: self._mu_forward(states, use_t))
        self.exploration_policy.get_action(self._mu_act, self.actions)
        return self.actions

    def _mu_forward(self, states, use_target):
        self._mu_act, _ = self.networks["actor"].forward(states, self.n_env, use_target=use_target, tag="act")

    def handle_episode_ended(self):
        if hasattr(self.exploration_policy, "reset"):
            ended = np.nonzero(self._episode_steps == 0)[0] if self._episode_just_ended else None
            try:                                                              # Agent.reset_internal_state, per env
                self.exploration_policy.reset(ended)
            except TypeError:
                self.exploration_policy.reset()

    # ------------------------------------------------------------------------------- training
    def _td_targets(self, b, q_next):
        alg = self.ap.algorithm
        clip = alg.clip_critic_targets
        self.lib.ac_td_targets(b.rewards(), b.game_overs(), q_next, 1, float(alg.discount),
                               int(bool(alg.use_non_zero_discount_for_terminal_states)),
                               int(clip is not None), float(clip[0]) if clip else 0.0,
                               float(clip[1]) if clip else 0.0, self.batch_size, self.td_targets,
                               _rlx.current_stream())

    def _sync(self, net):
        self._allreduce(net.params.grads)

    def _learn_device(self, b, mix=None):
        """mix: rate of the soft target update due right after this update — each network's Adam pass applies it to its
        own target (neither target is read again before the update ends)."""
        actor, critic = self.networks["actor"], self.networks["critic"]
        B = self.batch_size
        s, ns = b._states["observation"], b._next_states["observation"]
        next_actions, _ = actor.forward(ns, B, use_target=True, tag="next")
        actions_mean, a_saved = actor.forward(s, B, tag="train")
        q_next, _ = critic.forward(ns, next_actions, B, use_target=True, tag="next")
        self._td_targets(b, q_next[0])
        # gradients_wrt_inputs[1]['action'] of mean(Q) at the online actor's action (:169-173)
        _, c_saved = critic.forward(s, actions_mean, B, tag="agrad")
        critic.action_gradient(c_saved, B, self.neg_action_grad, scale=-1.0)
        # critic.train_and_sync_networks (:179-180)
        _, c_saved = critic.forward(s, b.actions(), B, tag="train")
        critic.train_backward(c_saved, self.td_targets, B)
        self._sync(critic)
        critic.apply_gradients(self._scale("critic"), with_norm=True, mix_rate=mix)
        # actor: weighted_gradients[0] with gradients_weights = -action_gradients (:183-193)
        actor.backward(a_saved, self.neg_action_grad, B)
        self._sync(actor)
        actor.apply_gradients(self._scale("actor"), mix_rate=mix)

    def _scale(self, name):
        netp = self.ap.network_wrappers[name]
        return self.dist.grad_scale(netp.scale_down_gradients_by_number_of_workers_for_sync_training) \
            if self.dist else 1.0

    def _update_record_fields(self):
        return []            # no per-update host draws: the record is the sampled rows (staged, eight updates per replay)

    def learn_from_batch(self, batch):
        mix = self._mix_rate
        # Agent.train brackets its updates with set_is_training(True / False) on every network, targets included
        # (agent.py:716,779): batch-norm layers normalise with batch statistics inside, the moving averages when acting
        for net in self.networks.values():
            net.set_is_training(True)
        try:
            self._run(("learn", mix), lambda: self._learn_device(batch, mix))
        finally:
            for net in self.networks.values():
                net.set_is_training(False)
        if mix is not None:
            self._mixed = self._mixed | {"actor", "critic"}
        critic = self.networks["critic"]
        self.signals = {"Loss": critic.loss[0], "Grads (unclipped)": critic.norm}
        return critic.loss[:critic.T].sum()
