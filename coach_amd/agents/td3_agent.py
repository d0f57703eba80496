"""TD3 on one MI355X — host-side mirror of rl_coach/agents/td3_agent.py (parameter classes
:36-131, TD3Agent.learn_from_batch :148-209, train :211-213,
update_transition_before_adding_to_replay_buffer :215-227) for the C4 workload
(HalfCheetah-like: obs 17, act 6, B = 100, 256 envs / GPU).

Twin critic streams run as two towers of one batched GEMM per layer.  Reference quirks kept:
  * training happens at episode end, ``current_episode_steps_counter`` updates in a row (:211-213);
  * game_over is cleared when the episode ended on the time limit (:215-227) — always the case for
    the fixed-length synthetic episodes;
  * the actor gradient is d mean_b(Q1) / d a (carries the 1/B factor, td3_v_head.py:57-58);
  * actor update and soft target update every 2nd training iteration (:186, TrainingSteps(2)).
"""
from collections import OrderedDict

import numpy as np
import torch

from .. import _rlx
from ..core_types import EnvironmentSteps, TrainingSteps
from ..exploration_policies.additive_noise import AdditiveNoiseParameters
from ..memories.episodic.episodic_experience_replay import EpisodicExperienceReplayParameters
from .ddpg_agent import DDPGActorNetworkParameters, DDPGAgent, DDPGCriticNetworkParameters
from .vector_agent import AlgorithmParameters


class TD3CriticNetworkParameters(DDPGCriticNetworkParameters):   # td3_agent.py:36-51 + Mujoco_TD3 preset
    def __init__(self, num_q_networks=2):
        super().__init__()
        self.observation_embedder_scheme = ()
        self.action_embedder_scheme = ()
        self.middleware_scheme = (400, 300)
        self.num_streams = num_q_networks
        self.head_initializer = "xavier"
        self.batch_size = 100
        self.learning_rate = 0.001


class TD3ActorNetworkParameters(DDPGActorNetworkParameters):     # td3_agent.py:54-68
    def __init__(self):
        super().__init__()
        self.batch_size = 100
        self.learning_rate = 0.001


class TD3AlgorithmParameters(AlgorithmParameters):               # td3_agent.py:71-111
    def __init__(self):
        super().__init__()
        self.rate_for_copying_weights_to_target = 0.005
        self.use_target_network_for_evaluation = False
        self.action_penalty = 0
        self.clip_critic_targets = None
        self.use_non_zero_discount_for_terminal_states = False
        self.act_for_full_episodes = True
        self.update_policy_every_x_episode_steps = 2
        self.num_steps_between_copying_online_weights_to_target = \
            TrainingSteps(self.update_policy_every_x_episode_steps)
        self.policy_noise = 0.2
        self.noise_clipping = 0.5
        self.num_q_networks = 2
        self.num_consecutive_playing_steps = EnvironmentSteps(1)
        self.clear_game_over_on_time_limit = True


class TD3AgentExplorationParameters(AdditiveNoiseParameters):    # td3_agent.py:114-117
    def __init__(self):
        super().__init__()
        self.noise_as_percentage_from_action_space = False


class TD3AgentParameters(object):                                # td3_agent.py:120-131
    def __init__(self):
        alg = TD3AlgorithmParameters()
        self.algorithm = alg
        self.exploration = TD3AgentExplorationParameters()
        self.memory = EpisodicExperienceReplayParameters()
        self.network_wrappers = OrderedDict([("actor", TD3ActorNetworkParameters()),
                                             ("critic", TD3CriticNetworkParameters(alg.num_q_networks))])
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.td3_agent:TD3Agent'


class TD3Agent(DDPGAgent):
    # The update as five launches (csrc/ac_fused.hip: row-local chains + one weight-gradient / Adam launch per network)
    # instead of ~36 layer-by-layer ones, wherever the networks have the Mujoco_TD3 topology (nn/fused_updates.py
    # FusedTD3.layers); the tests flip it to compare the two paths.
    FUSED_UPDATE = True

    def _fused(self):
        f = self.__dict__.get("_fused_td3", False)
        if f is False:
            from ..nn.fused_updates import FusedTD3
            f = FusedTD3(self) if FusedTD3.supported(self) else None
            self._fused_td3 = f
        return f if self.FUSED_UPDATE else None

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        super().__init__(agent_parameters, environment, device, dist, use_graphs)
        B, dev = self.batch_size, self.device
        self.noise = torch.zeros(B, self.A, dtype=torch.float64, device=dev)
        self.smoothed = torch.zeros(B, self.A, dtype=torch.float32, device=dev)
        self.q_min = torch.zeros(B, dtype=torch.float32, device=dev)
        self.zero_go = torch.zeros(self.n_env, dtype=torch.uint8, device=dev)
        self.act2 = torch.zeros(2, B, self.A, dtype=torch.float32, device=dev)   # (batch actions, smoothed a')

    def _stored_game_over(self, game_over):
        # the synthetic episodes end on their time limit -> game_over False (:215-227)
        return self.zero_go if self.ap.algorithm.clear_game_over_on_time_limit else game_over

    def _schedule_phase(self, iteration):
        return iteration % self.ap.algorithm.update_policy_every_x_episode_steps      # the delayed actor step (:186)

    def _training_steps_this_phase(self):
        # current_episode_steps_counter updates in a row at each episode end (:211-213): the phases of a vector
        # step take the lengths of the episodes that just finished, in env order
        return self._phase_episode_lengths.pop(0)

    def _critic_device_paired(self, b, obs2, mix=None):
        """The same update with every online / target pass pair sharing its launches: actor online(s)
        + target(s'), then critic online(s, a) + target(s', a') (the online critic pass does not
        depend on the TD targets, only its loss does)."""
        actor, critic = self.networks["actor"], self.networks["critic"]
        alg, B, s_ = self.ap.algorithm, self.batch_size, _rlx.current_stream()
        a2, self._a_saved = actor.forward_pair(obs2, B)
        self._actions_mean = a2[0]
        # merged critic inputs of both passes + the observation half of the action-gradient pass: one launch
        # (target-policy smoothing :162-165 included)
        merged2 = critic.merged_pair_buffer(B)
        self._agrad_merged = critic.merged_buffer(B, "agrad")
        o2 = obs2.view(2, B, critic.obs_dim)
        self.lib.ac_merge_inputs(b.actions(), o2[0], a2[1], self.noise, float(alg.noise_clipping), self.d_low,
                                 self.d_high, o2[1], B, self.A, critic.obs_dim, merged2, self._agrad_merged, s_)
        q2, c_saved = critic.forward_pair_merged(merged2, B)
        # min(Q1', Q2') (output #2, :168), TD targets (:171-180) and both streams' losses: one launch
        clip = alg.clip_critic_targets
        self.lib.ac_critic_losses(q2[1][0], q2[1][1], b.rewards(), b.game_overs(), float(alg.discount),
                                  int(bool(alg.use_non_zero_discount_for_terminal_states)), int(clip is not None),
                                  float(clip[0]) if clip else 0.0, float(clip[1]) if clip else 0.0, q2[0],
                                  critic.T, B, 1.0, self.q_min, self.td_targets, c_saved[2].ensure_grad(),
                                  critic.loss, s_)
        critic.train_backward(c_saved, self.td_targets, B, losses_done=True)
        self._sync(critic)
        critic.apply_gradients(self._scale("critic"), with_norm=True, mix_rate=mix)
        self._loss_total = critic.loss[critic.T]                 # written by the loss launch: no reduction launch

    def _critic_device(self, b, mix=None):
        """mix: rate of the soft target update due after this update, applied by the critic's Adam pass itself (nothing
        reads the critic's target between its Adam step and the end of the update)."""
        fused = self._fused()
        if fused is not None:
            critic = self.networks["critic"]
            wg = self.dist is not None               # data parallel: gradients out, all-reduce, the flat Adam launch
            fused.critic_update(b, mix, write_grads=wg)
            if wg:
                self._sync(critic)
                critic.apply_gradients(self._scale("critic"), with_norm=True, mix_rate=mix)
            self._loss_total = critic.loss[critic.T]
            self._agrad_merged = None
            return
        obs2 = b._info.get("states_pair") if hasattr(b, "_info") else None
        if obs2 is not None and self.networks["critic"].T == 2:
            return self._critic_device_paired(b, obs2, mix)
        actor, critic = self.networks["actor"], self.networks["critic"]
        alg, B, s_ = self.ap.algorithm, self.batch_size, _rlx.current_stream()
        self._agrad_merged = None
        s, ns = b._states["observation"], b._next_states["observation"]
        next_actions, _ = actor.forward(ns, B, use_target=True, tag="next")
        self._actions_mean, self._a_saved = actor.forward(s, B, tag="train")
        # target-policy smoothing (:162-165)
        self.lib.td3_smooth_actions(next_actions, self.noise, float(alg.noise_clipping), self.d_low,
                                    self.d_high, B, self.A, self.smoothed, s_)
        q_next, _ = critic.forward(ns, self.smoothed, B, use_target=True, tag="next")
        self.lib.min_pair(q_next[0], q_next[1], self.q_min, None, None, 0.0, B, s_)   # output #2 (:168)
        self._td_targets(b, self.q_min)
        _, c_saved = critic.forward(s, b.actions(), B, tag="train")
        critic.train_backward(c_saved, self.td_targets, B)
        self._sync(critic)
        critic.apply_gradients(self._scale("critic"), with_norm=True, mix_rate=mix)
        self._loss_total = None

    def _actor_device(self, b, mix=None):
        actor, critic = self.networks["actor"], self.networks["critic"]
        fused = self._fused()
        if fused is not None:
            wg = self.dist is not None
            fused.actor_update(b, mix, write_grads=wg)
            if wg:
                self._sync(actor)
                actor.apply_gradients(self._scale("actor"), mix_rate=mix)
            return
        B = self.batch_size
        s = b._states["observation"]
        if self._agrad_merged is not None:
            # [mu(s) | s]: the observation columns were written with the critic inputs, the action columns now
            m = self._agrad_merged
            self.lib.copy_2d(self._actions_mean, self.A, m, critic.merged, B, self.A, 1.0, _rlx.current_stream())
            _, c_saved = critic.forward_merged(m, B, tag="agrad")              # :188-192, output #3
            # - d mean(Q1) / d action, scaled to the actor's tanh output, straight into the actor head's gradient
            critic.action_gradient(c_saved, B, actor.head_grad(self._a_saved), scale=-actor._uniform_scale)
            actor.backward(self._a_saved, None, B)
        else:
            _, c_saved = critic.forward(s, self._actions_mean, B, tag="agrad")
            critic.action_gradient(c_saved, B, self.neg_action_grad, scale=-1.0)
            actor.backward(self._a_saved, self.neg_action_grad, B)
        self._sync(actor)
        actor.apply_gradients(self._scale("actor"), mix_rate=mix)

    def _update_record_fields(self):
        return [("z", (self.batch_size, self.A), torch.float64)]

    def _draw_update_host(self):
        alg = self.ap.algorithm
        return {"z": np.random.normal(0, alg.policy_noise, (self.batch_size, self.A))}    # :162 (host stream)

    def learn_from_batch(self, batch):
        alg, B = self.ap.algorithm, self.batch_size
        if self._staged is not None:
            self.noise = self._staged["z"]               # shipped with the sampled rows (one record per update)
        else:
            # the staging buffer IS the noise operand (a static device tensor): no device-to-device copy behind the upload
            self.noise = self._to_device("td3_noise", self._draw_update_host()["z"], torch.float64)
        run = self._run
        mix = self._mix_rate
        staged = self._staged is not None          # (the captured graphs read the noise at ITS address)
        run(("critic", mix, staged), lambda: self._critic_device(batch, mix))
        if self.training_iteration % alg.update_policy_every_x_episode_steps == 0:   # :186
            run(("actor", mix, staged), lambda: self._actor_device(batch, mix))
            if mix is not None:
                self._mixed = self._mixed | {"actor"}
        if mix is not None:
            self._mixed = self._mixed | {"critic"}
        critic = self.networks["critic"]
        loss = self._loss_total if self._loss_total is not None else critic.loss[:2].sum()
        self.signals = {"Loss": loss, "Grads (unclipped)": critic.norm}
        return loss
