"""Soft Actor-Critic on one MI355X — host-side mirror of
rl_coach/agents/soft_actor_critic_agent.py (parameter classes :57-141,
SoftActorCriticAgent.learn_from_batch :168-280, choose_action :296-322) for the C5 workload
(Humanoid-like: obs 376, act 17, B = 256, 512 envs / GPU).

The reference runs ~9 separate sess.run passes per update; the order is kept because it is
observable: the policy gradient is applied BEFORE the V and Q targets are formed,
and V's target reuses the first pass's sampled actions / log-probs.  Every sess.run of the TF graph
re-samples the policy noise, so the three policy passes (outputs, d mean(logp)/d phi, d a/d phi
weighted by dQ/da) see three independent noise draws; `resample_noise_per_pass=False` fuses them
into one pass on one draw (1 forward + 1 backward instead of 3 + 2).  TF's own random stream
cannot be reproduced: the standard normals come from the host's np.random stream.
"""
from collections import OrderedDict

import numpy as np
import torch

from .. import _rlx
from ..core_types import EnvironmentSteps, RunPhase
from ..memories.non_episodic.experience_replay import ExperienceReplayParameters
from ..nn.actor_critic_nets import SACPolicyNet, SACQNet, SACValueNet
from ..architectures.scheme_views import SchemeViews
from .vector_agent import AlgorithmParameters, VectorOffPolicyAgent


class _SACNetParams(SchemeViews):
    _TUPLE_SCHEMES = True

    def __init__(self):
        self.activation_function = 'relu'
        self.optimizer_type = 'Adam'
        self.batch_size = 256
        self.learning_rate = 0.0003
        self.adam_optimizer_beta1 = 0.9                  # NetworkParameters defaults (base_parameters.py:297-299)
        self.adam_optimizer_beta2 = 0.99
        self.optimizer_epsilon = 0.0001
        self.scale_down_gradients_by_number_of_workers_for_sync_training = True


class SACValueNetworkParameters(_SACNetParams):          # :57-69
    def __init__(self):
        super().__init__()
        self.embedder_scheme = (256,)
        self.middleware_scheme = (256,)                  # Mujoco_SAC preset
        self.create_target_network = True


class SACCriticNetworkParameters(_SACNetParams):         # :72-84
    def __init__(self):
        super().__init__()
        self.network_layers_sizes = (256, 256)           # SACQHeadParameters
        self.create_target_network = False

    @property
    def heads_parameters(self):
        """[SACQHeadParameters]: presets set heads_parameters[0].network_layers_sizes (Mujoco_SAC.py)."""
        owner = self

        class _Head(object):
            @property
            def network_layers_sizes(self):
                return owner.network_layers_sizes

            @network_layers_sizes.setter
            def network_layers_sizes(self, value):
                owner.network_layers_sizes = tuple(value)
        return [_Head()]


class SACPolicyNetworkParameters(_SACNetParams):         # :87-100
    def __init__(self):
        super().__init__()
        self.embedder_scheme = (256,)
        self.middleware_scheme = (256,)
        self.create_target_network = False


class SoftActorCriticAlgorithmParameters(AlgorithmParameters):   # :103-126
    def __init__(self):
        super().__init__()
        self.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(1)
        self.rate_for_copying_weights_to_target = 0.005
        self.use_deterministic_for_evaluation = True
        self.reward_rescale = 5.0                        # Mujoco_SAC preset: RewardRescaleFilter(5)
        self.resample_noise_per_pass = True


class SoftActorCriticAgentParameters(object):            # :129-141
    def __init__(self):
        self.algorithm = SoftActorCriticAlgorithmParameters()
        self.memory = ExperienceReplayParameters()
        self.network_wrappers = OrderedDict([("policy", SACPolicyNetworkParameters()),
                                             ("q", SACCriticNetworkParameters()),
                                             ("v", SACValueNetworkParameters())])
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.soft_actor_critic_agent:SoftActorCriticAgent'


class SoftActorCriticAgent(VectorOffPolicyAgent):
    SIGNAL_NAMES = VectorOffPolicyAgent.SIGNAL_NAMES + [                  # soft_actor_critic_agent.py:151-162
        "Policy_mu_avg", "Policy_logsig", "Policy_logp_sampled", "Policy_grads_sumabs", "Q1", "TD err1", "Q2",
        "TD err2", "V_tgt_ns", "V_onl_ys", "actions"]
    continuous = True

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        super().__init__(agent_parameters, environment, device, dist, use_graphs)
        if self.image:
            raise ValueError("SAC works only for continuous control problems")
        ep = environment.p
        pn, qn, vn = (self.ap.network_wrappers[k] for k in ("policy", "q", "v"))
        self.obs_dim, self.A = int(ep.observation_shape[0]), int(ep.action_dim)
        self.batch_size = pn.batch_size
        self.low = np.broadcast_to(np.asarray(ep.action_low, dtype=np.float32), (self.A,)).copy()
        self.high = np.broadcast_to(np.asarray(ep.action_high, dtype=np.float32), (self.A,)).copy()
        seed, dev = self.ap.seed or 0, self.device
        adam = lambda n: (n.learning_rate, n.adam_optimizer_beta1, n.adam_optimizer_beta2, n.optimizer_epsilon)
        self.networks = OrderedDict([
            ("policy", SACPolicyNet(dev, self.obs_dim, self.A, pn.embedder_scheme, pn.middleware_scheme,
                                    *adam(pn), seed=seed)),
            ("q", SACQNet(dev, self.obs_dim, self.A, qn.network_layers_sizes, *adam(qn), seed=seed + 1)),
            ("v", SACValueNet(dev, self.obs_dim, vn.embedder_scheme, vn.middleware_scheme, *adam(vn),
                              seed=seed + 2))])
        self.memory = self._make_memory(action_dim=self.A)
        B = self.batch_size
        f32 = torch.float32
        self.actions = torch.zeros(self.n_env, self.A, dtype=f32, device=dev)
        self.act_normals = torch.zeros(self.n_env, self.A, dtype=torch.float64, device=dev)
        self.normals = torch.zeros(3, B, self.A, dtype=torch.float64, device=dev)
        self.log_target = torch.zeros(B, dtype=f32, device=dev)
        self.dq_da = torch.zeros(B, self.A, dtype=f32, device=dev)
        self.value_targets = torch.zeros(B, dtype=f32, device=dev)
        self.td_targets = torch.zeros(B, dtype=f32, device=dev)
        self.policy_grads_sumabs = torch.zeros(1, dtype=f32, device=dev)
        self._finish_init()

    # --------------------------------------------------------------------------------- acting
    def random_actions(self):
        a = np.random.uniform(self.low, self.high, (self.n_env, self.A)).astype(np.float32)
        self.actions.copy_(self._to_device("rand_act", a, torch.float32))
        return self.actions

    def choose_action(self, states):
        """choose_action (:296-322): the squashed sample, or the (un-squashed) mean in TEST."""
        alg = self.ap.algorithm
        det = self.phase == RunPhase.TEST and alg.use_deterministic_for_evaluation
        z = np.random.standard_normal((self.n_env, self.A))
        self.act_normals.copy_(self._to_device("act_z", z, torch.float64))
        self._run(("pi", det), lambda: self._pi_forward(states, det))
        return self.actions

    def _pi_forward(self, states, deterministic):
        o, _ = self.networks["policy"].forward(states, self.n_env, self.act_normals, tag="act")
        self.actions.copy_(o["mean"] if deterministic else o["actions"])

    # ------------------------------------------------------------------------------- training
    def _sync(self, net):
        self._allreduce(net.params.grads)

    def _scale(self, name):
        netp = self.ap.network_wrappers[name]
        return self.dist.grad_scale(netp.scale_down_gradients_by_number_of_workers_for_sync_training) \
            if self.dist else 1.0

    # The three networks' passes as parallel branches of the captured update (fork / join of side streams): the policy
    # chain, V's training pass and Q's training pass touch different networks until their losses meet.  Same calls,
    # same arithmetic, same results as the sequential order.  OFF: measured on C5 (profiles/r04_ab_sac_branches.txt) the
    # update's device time does not move (239-256 us with, 240-243 us without: the cross-stream edges of a hipGraph cost
    # what the overlap returns at these launch sizes, as round 2 found for the weight-gradient side stream) and the
    # multi-stream graph launch costs the host 11 % of the C5 rate.
    parallel_branches = False

    # The update as six launches (csrc/ac_fused.hip: every launch carries one level of ALL networks + one weight-gradient /
    # Adam launch for the three of them) instead of ~42 layer-by-layer ones, wherever the networks have the Mujoco_SAC
    # topology (nn/fused_updates.py FusedSAC.layers); the tests flip it to compare the two paths.
    FUSED_UPDATE = True

    def _fused(self):
        f = self.__dict__.get("_fused_sac", False)
        if f is False:
            from ..nn.fused_updates import FusedSAC
            f = FusedSAC(self) if FusedSAC.supported(self) else None
            self._fused_sac = f
        return f if self.FUSED_UPDATE else None

    def _v_mixes_in_update(self, batch):
        """Whether a soft target update due after this update rides in V's Adam pass: always in the fused chain (it
        evaluates V_target(s') first), otherwise only when the paired pass did."""
        return self._fused() is not None or (hasattr(batch, "_info") and batch._info.get("states_pair") is not None)

    def _learn_device(self, b, mix=None):
        pol, q, v = self.networks["policy"], self.networks["q"], self.networks["v"]
        fused = self._fused()
        if fused is not None:
            wg = self.dist is not None               # data parallel: gradients out, all-reduce, the flat Adam launches
            fused.update(b, mix, write_grads=wg)
            if wg:
                for name, net in (("policy", pol), ("q", q), ("v", v)):
                    self._sync(net)
                    net.apply_gradients(self._scale(name), **({"with_norm": True} if name == "q" else
                                                              {"mix_rate": mix} if name == "v" else {}))
            return
        alg, B, s_ = self.ap.algorithm, self.batch_size, _rlx.current_stream
        s, ns = b._states["observation"], b._next_states["observation"]
        resample = alg.resample_noise_per_pass
        obs2 = b._info.get("states_pair") if hasattr(b, "_info") else None
        br = None
        if self.parallel_branches and self.dist is None and obs2 is not None and torch.cuda.is_available():
            from .vector_agent import Branches
            br = getattr(self, "_branches", None)
            if br is None:
                br = self._branches = Branches(2)
                # V's and Q's backward passes run on side streams next to each other: their deferred split-K partial
                # sums must not share the per-device arena (both would start at offset 0)
                v.ctx.private_arena = q.ctx.private_arena = True
            br.fork()
        # ---- branch A (the stream of the caller): policy.  (1) policy outputs on the first noise draw (:186-190)
        o, p_saved = pol.forward(s, B, self.normals[0], tag="train0")
        # (2) Q(s, sampled actions): log_target = min(Q1, Q2) (:198-200)
        qv, q_saved_pi = q.forward(s, o["actions"], B, tag="pi")
        # min(Q1, Q2), the V targets of (4) = log_target - logp of this first pass (:244) and d mean(min) / d Q_i
        # for (3): one launch
        dqv = q_saved_pi[4].ensure_grad().view(2, B)
        self.lib.sac_min_targets(qv[0], qv[1], o["logprob"], 1.0 / B, B, self.log_target, self.value_targets,
                                 dqv[0], dqv[1], s_())
        targets_ready = br.mark() if br else None
        # (3) d mean(Q_min) / d a at the sampled actions (:216-217)
        q.action_gradient(q_saved_pi, B, self.dq_da, dq_done=True)
        q_weights_free = br.mark() if br else None           # nothing of branch A reads Q's weights after this
        if resample:
            # The three sess.run passes of the reference re-evaluate the SAME deterministic torso
            # (same weights, same states) and only re-sample the head's noise, so their gradients
            # w.r.t. the head output can be summed before ONE backward pass through the torso:
            # weighted_gradients[5] (mean log-prob, weight 1) on draw 1 (:210-213) minus
            # weighted_gradients[3] (actions, weights dq_da) on draw 2 (:221-227)
            pol.head_gradient(p_saved, B, self.normals[1], logprob_mean_weight=1.0)
            pol.head_gradient(p_saved, B, self.normals[2], action_weights=self.dq_da,
                              action_weight_scale=-1.0, accumulate=True)
            pol.backward_torso(p_saved)
        else:
            pol.backward(p_saved, B, logprob_mean_weight=1.0, action_weights=self.dq_da,
                         action_weight_scale=-1.0)
        self._sync(pol)
        pol.apply_gradients(self._scale("policy"))                               # :229

        # ---- branch B: (4) V, train_on_batch on the targets computed in (2) (:250)
        import contextlib
        with (br.on(0) if br else contextlib.nullcontext()):
            if obs2 is not None:
                # V_online(s) of this training pass and V_target(s') of the Q targets share their launches
                v2, v_saved = v.forward_pair(obs2, B)
                v_next = v2[1]
            else:
                _, v_saved = v.forward(s, B, tag="train")
            v_next_ready = br.mark() if br else None
            if br:
                br.after(targets_ready)
            v.train_backward(v_saved, self.value_targets, B)
            self._sync(v)
            # a soft update of V's target due after this update rides in V's Adam pass — only when V_target(s') was
            # already evaluated above (paired pass); otherwise it is still needed unmixed below
            v.apply_gradients(self._scale("v"), mix_rate=mix if obs2 is not None else None)

        # ---- branch C: (5) Q, y = r + (1 - done) gamma V_target(s') (:259-266), train_on_batch (:268)
        with (br.on(1) if br else contextlib.nullcontext()):
            if obs2 is None:
                v_next, _ = v.forward(ns, B, use_target=True, tag="next")
            qt, q_saved = q.forward(s, b.actions(), B, tag="train")
            if br:
                br.after(v_next_ready)
            # TD targets and both Q losses (0.5 * mse each, sac_q_head.py:91-95) + their sum: one launch
            self.lib.ac_critic_losses(v_next, None, b.rewards(), b.game_overs(), float(alg.discount), 0, 0, 0.0, 0.0,
                                      qt, 2, B, 0.5, None, self.td_targets, q_saved[4].ensure_grad(), q.loss, s_())
            q.train_backward(q_saved, None, B)
            self._sync(q)
            if br:
                br.after(q_weights_free)
            q.apply_gradients(self._scale("q"), with_norm=True)
        if br:
            br.join()

    def _update_record_fields(self):
        return [("z", (3, self.batch_size, self.A), torch.float64)]

    def _draw_update_host(self):
        return {"z": np.random.standard_normal((3, self.batch_size, self.A))}

    def learn_from_batch(self, batch):
        B = self.batch_size
        if self._staged is not None:
            self.normals = self._staged["z"]             # shipped with the sampled rows (one record per update)
        else:
            self.normals = self._to_device("sac_z", self._draw_update_host()["z"], torch.float64)     # the staging buffer is the operand
        mix = self._mix_rate
        self._run(("learn", mix, self._staged is not None), lambda: self._learn_device(batch, mix))
        if mix is not None and self._v_mixes_in_update(batch):     # (decided here, not in the captured body: replays skip that)
            self._mixed = self._mixed | {"v"}
        qn = self.networks["q"]
        self.signals = {"Loss": qn.loss[2], "Grads (unclipped)": qn.norm,
                        "V loss": self.networks["v"].loss}
        return qn.loss[2]
