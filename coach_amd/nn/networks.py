"""Device-resident networks of the hot-path agents, assembled from nn.graph layers.

Topologies follow rl_coach/architectures/tensorflow_components:
  embedders/image_embedder.py:57-63   Medium = conv 32x8x8/4, 64x4x4/2, 64x3x3/1
  embedders/vector_embedder.py:51-60  Medium = Dense 256, Shallow = Dense 128
  middlewares/fc_middleware.py:40-52  Medium = Dense 512, Shallow = Dense 64
  heads/q_head.py, v_head.py:43-48 (normalized-columns init, std 1.0), ppo_head.py:100-116
and the per-agent network parameters (agents/dqn_agent.py:43-56, clipped_ppo_agent.py:41-58).
"""
import numpy as np
import torch

from .. import _rlx
from . import graph as G

IMAGE_EMBEDDER = {"Medium": [(32, 8, 4), (64, 4, 2), (64, 3, 1)], "Shallow": [(32, 3, 1)]}
VECTOR_EMBEDDER = {"Medium": [256], "Shallow": [128], "Empty": []}
FC_MIDDLEWARE = {"Medium": [512], "Shallow": [64], "Empty": []}


def build_torso(params, prefix, obs_shape, activation, towers, embedder="Medium", middleware="Medium"):
    """input embedder + FC middleware.  obs_shape (H, W, C) -> image embedder on uint8 frames
    (input / 255, embedder_parameters.py:33-36), (D,) -> vector embedder."""
    layers = []
    if len(obs_shape) == 3:
        hwc = tuple(obs_shape)
        convs = IMAGE_EMBEDDER[embedder] if isinstance(embedder, str) else embedder
        for i, (f, k, s) in enumerate(convs):
            c = G.Conv2d(params, "%s/embedder/conv%d" % (prefix, i), hwc, f, k, s, activation, towers)
            layers.append(c)
            hwc = c.out_hwc
        feat = hwc[0] * hwc[1] * hwc[2]
    else:
        feat = int(obs_shape[0])
        dense = VECTOR_EMBEDDER[embedder] if isinstance(embedder, str) else embedder
        for i, u in enumerate(dense):
            layers.append(G.Dense(params, "%s/embedder/dense%d" % (prefix, i), feat, u, activation, towers))
            feat = u
    mids = FC_MIDDLEWARE[middleware] if isinstance(middleware, str) else middleware
    for i, u in enumerate(mids):
        layers.append(G.Dense(params, "%s/middleware/dense%d" % (prefix, i), feat, u, activation, towers))
        feat = u
    return G.Sequential(layers), feat


class _NetBase:
    def _finish(self, device, seed, lr, beta1, beta2, eps, has_target=True):
        self.device = device
        self.params.finalize(device)
        rng = np.random.RandomState(seed)
        for m in self.modules:
            m.initialize(rng)
        self.ctx = G.Context(device)
        self.lib = self.ctx.lib
        self.adam = G.AdamState(self.params, lr, beta1, beta2, eps)
        if has_target:
            self.target = self.params.target_weights
            self.target.copy_(self.params.weights)
        else:
            self.target = None
        self.norm = torch.zeros(1, dtype=torch.float32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)

    def obs_tensor(self, obs, B):
        if self.image:
            return G.input_tensor(obs, B, int(np.prod(self.obs_shape)), u8=True, div=255.0)
        return G.input_tensor(obs, B, int(self.obs_shape[0]))

    def update_target(self, rate=1.0):
        """NetworkWrapper.update_target_network (network_wrapper.py:109-116): w_t <- rate*w_o + (1-rate)*w_t."""
        self.lib.mix_weights(self.target, self.params.weights, self.params.size, float(rate), self.ctx.stream)

    def grad_norm(self):
        self.lib.global_norm(self.params.grads, self.params.size, self.norm, self.ctx.ws.small,
                             self.ctx.ws.small.numel(), self.ctx.stream)
        return self.norm

    def apply_gradients(self, grad_scale=1.0, with_norm=False, acc=None, mix_rate=None):
        """Adam; with_norm also refreshes self.norm = tf.global_norm(grads) in the same pass; mix_rate (a soft
        target update is due right after this step): the target weights are mixed in the same pass."""
        kw = {}
        if mix_rate is not None and self.target is not None and self.adam.one_launch:
            kw = dict(mix_target=self.target, mix_rate=float(mix_rate))
        elif mix_rate is not None and self.target is not None:
            kw = {}
        if with_norm:
            self.adam.step(grad_scale, norm_out=self.norm, workspace=self.ctx.ws.small, acc=acc, **kw)
        else:
            self.adam.step(grad_scale, **kw)
        if mix_rate is not None and self.target is not None and not kw:
            self.update_target(mix_rate)
        for bn in getattr(self, "bn_layers", ()):          # UPDATE_OPS of apply_gradients (architecture.py:273-277)
            bn.commit(self.ctx)

    def set_is_training(self, state):
        """NetworkWrapper.set_is_training (network_wrapper.py:215-224): batch-norm layers use batch statistics."""
        self.ctx.bn_training = bool(state)

    clip_gradients = None

    def clip_by_global_norm(self):
        """clip_gradients (ClipByGlobalNorm, architecture.py:196-200): self.norm <- tf.global_norm of
        the raw gradients, then the gradient buffer is rescaled in place.  False when clipping is off."""
        if not self.clip_gradients:
            return False
        self.grad_norm()
        self.lib.clip_by_global_norm(self.params.grads, self.params.size, self.norm,
                                     float(self.clip_gradients), self.ctx.stream)
        return True

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise ValueError("device kernel reported invalid input (status bits %d)" % s)


class ClippedPPONet(_NetBase):
    """ClippedPPONetworkParameters (agents/clipped_ppo_agent.py:41-58): two full copies of
    embedder + middleware (use_separate_networks_per_head), head 0 = VHead, head 1 = PPOHead
    (discrete).  tower 0 = value, tower 1 = policy."""
    HEADS_LOSS_BACKWARD_ONE_LAUNCH = True     # discrete heads: losses + heads' backward as one launch (tests flip it)
    HEADS_FORWARD_WITH_TORSO = True           # the heads' forward inside the last dense layer's split-K reduction (tests flip it)
    # the last dense layer of both towers + heads forward + both losses + heads backward as ONE launch (rlx_ppo_fc_heads,
    # csrc/ppo_fc_fused.hip) where the shape allows it (discrete heads, minibatch <= 64 rows).  OFF: measured on C2 it is
    # 34.8 us against 11.0 + 8.0 + 11.2 us for the three launches it replaces (1 MB of split-K partials instead of 6.5 MB, 10
    # launches per update instead of 12) and the update comes out 198.1 us against 193.8 us (same box, bench.py --fc-heads
    # 1 / 0, profiles/r06_ab_fc_heads.txt): its three in-launch hand-offs (8 K splits -> tile, 16 tiles -> tower, dy back to
    # the tiles) each cost a write-through + read round trip, which is what a launch boundary costs here.  Entry point,
    # parity test (tests/test_ppo_fc_fused.py) and A/B flag stay.
    FC_HEADS_ONE_LAUNCH = False
    # discrete heads, minibatch <= 256 rows: what of the heads' losses and backward pass is LOCAL TO A ROW (heads forward, the
    # row's loss terms, dV / dlogits, dz of the last dense layer) runs in the workgroup that finishes that row of the dense
    # layer's K-split reduction (rlx_ppo_fc_rows), and the rest — dW / db of the heads, the loss scalars, which nobody reads
    # before the Adam step — as extra workgroups of the backward pass's deferred-reduction launch
    # (rlx_splitk_reduce_jobs_ppo_tail): the heads' launch (11.3 us of the C2 update, tens of workgroups) disappears from
    # the chain.  Bit-identical to the three-launch path (tests/test_ppo_fc_rows.py); tests and bench.py --heads-row-local flip it.
    HEADS_ROW_LOCAL = True

    def __init__(self, device, obs_shape, n_actions, activation="tanh", embedder="Medium",
                 middleware="Medium", learning_rate=2.5e-4, adam_beta1=0.9, adam_beta2=0.99,
                 optimizer_epsilon=1e-4, clip_likelihood_ratio_using_epsilon=0.2, beta_entropy=0.01,
                 seed=0, continuous=False):
        """continuous: n_actions is the action dimension; the head is ppo_head.py:118-144 (policy_mean
        Dense with normalized-columns(0.01) init + one state-independent policy_log_std vector)."""
        self.obs_shape, self.image, self.A = tuple(obs_shape), len(obs_shape) == 3, n_actions
        self.clip_eps, self.beta = clip_likelihood_ratio_using_epsilon, beta_entropy
        self._clip_scale_dev, self._clip_scale_value = None, None
        self.continuous = continuous
        # (heads forward / losses / heads backward as ONE launch was tried twice and lost to these three small grids:
        # profiles/r02_ab_fused_heads.txt)
        self.params = G.FlatParams()
        self.torso, feat = build_torso(self.params, "main", obs_shape, activation, 2, embedder, middleware)
        self.v_head = G.Dense(self.params, "main/v_head/dense", feat, 1, None, 1,
                              init=G.normalized_columns(1.0))                   # v_head.py:43-48
        if continuous:
            self.pi_head = G.Dense(self.params, "main/ppo_head/policy_mean", feat, n_actions, None, 1,
                                   init=G.normalized_columns(0.01))
            self.params.add_group([("main/ppo_head/policy_log_std", (n_actions,))])   # zeros (:133-137)
        else:
            self.pi_head = G.Dense(self.params, "main/ppo_head/policy_fc", feat, n_actions, None, 1)
        self.modules = [self.torso, self.v_head, self.pi_head]
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon)
        # [surrogate, entropy, kl, policy-head total, value loss, grad norm]: one contiguous record
        self.scalars = torch.zeros(8, dtype=torch.float32, device=device)
        self.norm = self.scalars[5:6]

    # ---- inference -------------------------------------------------------------------------
    def policy_probs(self, obs, B, use_target=False, tag="act", out=None, sample=None):
        """softmax(policy_fc(policy tower(obs)))  — target weights = the frozen 'old policy'.
        sample=(uniforms [B] fp64, actions [B] int32 out): draw the actions in the same launch and return None."""
        w = self.target if use_target else None
        if self.HEADS_FORWARD_WITH_TORSO and self.pi_head.N <= G.SMALL_N and self.torso.layers[-1].N > G.SMALL_N:
            acts, (logits,) = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, weights=w, t0=1, nt=1,
                                                 row_heads=[(self.pi_head, 0)])
        else:
            acts = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, weights=w, t0=1, nt=1)
            logits = self.pi_head.forward(self.ctx, acts[-1], tag=tag, weights=w)
        if sample is not None:
            # acting: the categorical draw in the softmax launch (rlx_softmax_categorical_sample); the probabilities
            # themselves are not needed by anyone
            uniforms, actions = sample
            self.lib.softmax_categorical_sample(logits.data, self.A, uniforms, B, self.A, None, 0, actions, self.ctx.stream)
            return None
        probs = out if out is not None else self.ctx.buffer("probs", (B, self.A), tag=tag)
        self.lib.softmax(logits.data, self.A, B, self.A, probs, self.A, self.ctx.stream)
        return probs

    def act_and_record(self, obs, B, uniforms, actions, value_out, probs_out, tag="actrec"):
        """One acting step of the discrete agent that also leaves what the training phase needs from these states:
        BOTH towers run (the forward of forward_backward, launch for launch), the categorical draw happens in the softmax
        launch as in policy_probs(sample=...), V(s) goes to value_out [B] and the action probabilities to probs_out
        [B, A] — rows of the rollout's own columns.  The weights do not change between a rollout's steps and its training
        phase, and the old policy of that phase IS the acting policy (networks['main'].sync() at its start,
        clipped_ppo_agent.py:326), so fill_advantages' pass over the whole dataset (:161-170) and the old-policy pass
        (:238-241) recompute exactly these numbers: with them recorded here both passes disappear."""
        ctx = self.ctx
        fused_heads = self.pi_head.N <= G.SMALL_N and B * self.pi_head.N <= 1024
        if fused_heads and self.HEADS_FORWARD_WITH_TORSO and self.torso.layers[-1].N > G.SMALL_N:
            acts, (v, logits) = self.torso.forward(ctx, self.obs_tensor(obs, B), tag=tag,
                                                   row_heads=[(self.v_head, 0, value_out), (self.pi_head, 1)])
        else:
            acts = self.torso.forward(ctx, self.obs_tensor(obs, B), tag=tag)
            mid = acts[-1]
            if fused_heads:
                v, logits = G.small_dense_forward_multi(ctx, [(self.v_head, mid.tower_view(0)), (self.pi_head, mid.tower_view(1))], tag=tag)
            else:
                v = self.v_head.forward(ctx, mid.tower_view(0), tag=tag)
                logits = self.pi_head.forward(ctx, mid.tower_view(1), tag=tag)
            value_out.copy_(v.data.view(-1))
        self.lib.softmax_categorical_sample(logits.data, self.A, uniforms, B, self.A, probs_out, self.A, actions, ctx.stream)

    def policy_mean_std(self, obs, B, use_target=False, tag="act", out_mean=None, out_std=None):
        """Continuous head outputs [policy_mean, policy_std] (ppo_head.py:139-144): std = exp(log_std)
        tiled over the batch (the +eps is added where the distribution is built)."""
        w = self.target if use_target else None
        acts = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, weights=w, t0=1, nt=1)
        mean = self.pi_head.forward(self.ctx, acts[-1], tag=tag, weights=w).data.view(B, self.A)
        std = out_std if out_std is not None else self.ctx.buffer("policy_std", (B, self.A), tag=tag)
        ls = self.params.w("main/ppo_head/policy_log_std", 0, w)
        self.lib.exp_rows(ls, std, B, self.A, self.ctx.stream)
        if out_mean is not None:
            out_mean.copy_(mean)
            mean = out_mean
        return mean, std

    def values(self, obs, B, tag="val", out=None):
        """V(s) from the value tower of the online network (fill_advantages :161-170)."""
        if self.HEADS_FORWARD_WITH_TORSO and self.torso.layers[-1].N > G.SMALL_N:
            acts, (v,) = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, t0=0, nt=1,
                                            row_heads=[(self.v_head, 0)])
        else:
            acts = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, t0=0, nt=1)
            v = self.v_head.forward(self.ctx, acts[-1], tag=tag)
        if out is not None:
            out.copy_(v.data.view(-1))
            return out
        return v.data.view(-1)

    # ---- one minibatch of ClippedPPOAgent.train_network (:226-266) ---------------------------
    def late_gradient_offset(self):
        """Offset in the flat buffers where the parameters whose gradients are produced FIRST in the
        backward pass start (FC middleware + heads: 95 % of the bytes).  [offset, size) can be
        all-reduced while the convolution gradients [0, offset) are still being computed."""
        return self.params.entries[self.torso.layers[self._split_layer()].kname][0]

    def _split_layer(self):
        from .graph import Conv2d
        for i, l in enumerate(self.torso.layers):
            if not isinstance(l, Conv2d):
                return i
        return 0

    def set_clip_rescaler(self, value):
        """clip_param_rescaler (clipped_ppo_agent.py:266-268) as a DEVICE scalar: forward_backward(clip_rescaler=None)
        then multiplies the clip range by it inside the loss kernel, so a captured graph serves every value of a
        decaying clipping_decay_schedule."""
        if self._clip_scale_dev is None:
            self._clip_scale_dev = torch.ones(1, dtype=torch.float32, device=self.device)
        if self._clip_scale_value != float(value):
            self._clip_scale_dev.fill_(float(value))
            self._clip_scale_value = float(value)

    def forward_backward(self, obs, B, actions, advantages, value_targets, old_probs,
                         clip_rescaler=1.0, ratio_out=None, clipped_out=None, stop_after_dense=False):
        """accumulate_gradients (tensorflow_components/architecture.py:312-385): forward both towers,
        head losses, backward; leaves d total_loss / d theta in params.grads.
        stop_after_dense: stop once the dense layers' gradients are final (backward_rest() resumes).
        clip_rescaler None: the device scalar of set_clip_rescaler."""
        ctx = self.ctx
        clip_dev = self._clip_scale_dev if clip_rescaler is None else None
        if clip_rescaler is None:
            assert clip_dev is not None, "set_clip_rescaler first"
            clip_rescaler = 1.0
        fused_heads = self.pi_head.N <= G.SMALL_N and B * self.pi_head.N <= 1024
        heads_with_torso = fused_heads and self.HEADS_FORWARD_WITH_TORSO and self.torso.layers[-1].N > G.SMALL_N
        discrete_fused = heads_with_torso and not self.continuous and self.HEADS_LOSS_BACKWARD_ONE_LAUNCH
        if (self.FC_HEADS_ONE_LAUNCH or (self.HEADS_ROW_LOCAL and B <= 256)) and discrete_fused:
            acts = self.torso.forward(ctx, self.obs_tensor(obs, B), tag="train", skip_last=True)
            if self.HEADS_ROW_LOCAL and B <= 256 and not self.FC_HEADS_ONE_LAUNCH and len(acts) == len(self.torso.layers):
                res = G.ppo_fc_rows(ctx, self.torso.layers[-1], acts[-1], self.v_head, self.pi_head, value_targets, actions,
                                    advantages, old_probs, self.A, self.clip_eps * clip_rescaler, clip_dev, self.beta,
                                    self.scalars, ratio_out, clipped_out, self.status, tag="train",
                                    tail_now=stop_after_dense)
                if res is not None:
                    acts.append(res[0])
                    return self._backward_torso(acts, stop_after_dense)
            if self.FC_HEADS_ONE_LAUNCH and len(acts) == len(self.torso.layers) and \
                    G.ppo_fc_heads_supported(ctx, self.torso.layers[-1], acts[-1], self.v_head, self.pi_head):
                mid, v, logits = G.ppo_fc_heads(ctx, self.torso.layers[-1], acts[-1], self.v_head, self.pi_head, value_targets,
                                                actions, advantages, old_probs, self.A, self.clip_eps * clip_rescaler, clip_dev,
                                                self.beta, self.scalars, ratio_out, clipped_out, self.status, tag="train")
                acts.append(mid)
                return self._backward_torso(acts, stop_after_dense)
            # (not this shape: the layer-by-layer path below recomputes nothing — acts holds the layers in front of the last)
            last = self.torso.layers[-1]
            if heads_with_torso:
                y, (v, logits) = last.forward(ctx, acts[-1], tag="train", row_heads=[(self.v_head, 0), (self.pi_head, 1)])
            else:
                y = last.forward(ctx, acts[-1], tag="train")
            acts.append(y)
        elif heads_with_torso:   # the heads' forward rides on the last dense layer's launch (rlx_gemm_desc.row_heads)
            acts, (v, logits) = self.torso.forward(ctx, self.obs_tensor(obs, B), tag="train",
                                                   row_heads=[(self.v_head, 0), (self.pi_head, 1)])
        else:
            acts = self.torso.forward(ctx, self.obs_tensor(obs, B), tag="train")
        mid = acts[-1]
        mid.ensure_grad()
        xv, xp = mid.tower(0), mid.tower(1)
        if heads_with_torso:
            pass
        elif fused_heads:      # value + policy head in one launch (forward here, backward below)
            v, logits = G.small_dense_forward_multi(ctx, [(self.v_head, xv), (self.pi_head, xp)], tag="train")
        else:
            v = self.v_head.forward(ctx, xv, tag="train")
            logits = self.pi_head.forward(ctx, xp, tag="train")
        one_launch = fused_heads and not self.continuous and B <= 256 and self.HEADS_LOSS_BACKWARD_ONE_LAUNCH
        if one_launch:         # both head losses + the heads' backward pass: rlx_ppo_heads_loss_backward
            G.ppo_heads_loss_backward(ctx, (self.v_head, xv, v), (self.pi_head, xp, logits), value_targets, actions,
                                      advantages, old_probs, self.A, self.clip_eps * clip_rescaler, clip_dev, self.beta,
                                      self.scalars, ratio_out, clipped_out, self.status)
        dv, dlogits = v.ensure_grad(), logits.ensure_grad()
        # head 0: VHead, MSE(target, V), loss weight 1 (head.py:172-181)
        if one_launch:
            pass                                                   # (done above)
        elif not self.continuous:
            # ... and head 1, the discrete PPOHead clipped surrogate (+ entropy bonus), in one launch
            self.lib.ppo_discrete_value_losses(logits.data, self.A, actions, advantages, old_probs, self.A, B,
                                               self.A, self.clip_eps * clip_rescaler, self.beta, 1.0, dlogits,
                                               self.A, self.scalars[0:4], ratio_out, clipped_out, self.status,
                                               v.data, value_targets, dv, self.scalars[4:5], clip_dev, ctx.stream)
        else:
            self.lib.regression_loss(v.data, 1, value_targets, 1, None, B, 1, 0, 1.0, 1.0, dv, 1,
                                     self.scalars[4:5], ctx.stream)
        # head 1: PPOHead clipped surrogate (+ entropy bonus)
        if self.continuous:
            old_mean, old_std = old_probs                          # [policy_mean, policy_std] of the old policy
            self.lib.ppo_continuous_loss(logits.data, self.A, self.params.w("main/ppo_head/policy_log_std"),
                                         actions, advantages, old_mean, old_std, self.A, B, self.A,
                                         self.clip_eps * clip_rescaler, self.beta, 1.0, dlogits, self.A,
                                         self.params.g("main/ppo_head/policy_log_std"), self.scalars[0:4],
                                         ratio_out, clipped_out, clip_dev, ctx.stream)

        if one_launch:
            pass
        elif fused_heads:
            G.small_dense_backward_multi(ctx, [(self.v_head, xv, v), (self.pi_head, xp, logits)])
        else:
            self.v_head.backward(ctx, xv, v)
            self.pi_head.backward(ctx, xp, logits)
        mid.grad_is_dz = xv.grad_is_dz and xp.grad_is_dz     # the heads wrote dz of the middleware
        self._backward_torso(acts, stop_after_dense)

    def _backward_torso(self, acts, stop_after_dense):
        ctx = self.ctx
        if stop_after_dense:
            k = self._split_layer()
            self.torso.backward(ctx, acts, layers=(k, len(self.torso.layers)))
            self._resume = (acts, k)
        else:
            self.torso.backward(ctx, acts)
        ctx.flush_ppo_tail()        # (no deferred-reduction launch took the heads' all-rows part along)

    def backward_rest(self):
        acts, k = self._resume
        if k > 0:
            self.torso.backward(self.ctx, acts, layers=(0, k))

    def finish_update(self, grad_scale=1.0, signal_acc=None):
        """apply_gradients (architecture.py:469-521): global norm fetch + Adam (+ the running sums of
        [surrogate, entropy, kl, policy total, value loss, grad norm] into signal_acc)."""
        self.apply_gradients(grad_scale, with_norm=True,
                             acc=(self.scalars, signal_acc, 6) if signal_acc is not None else None)

    def train_minibatch(self, obs, B, actions, advantages, value_targets, old_probs,
                        clip_rescaler=1.0, grad_scale=1.0, ratio_out=None, clipped_out=None):
        self.forward_backward(obs, B, actions, advantages, value_targets, old_probs, clip_rescaler,
                              ratio_out, clipped_out)
        self.finish_update(grad_scale)
        # scalars: [surrogate, entropy, kl, policy head total, value loss]
        return self.scalars


class DQNNet(_NetBase):
    """DQNNetworkParameters (agents/dqn_agent.py:43-56): embedder -> FC middleware -> QHead;
    MSE or Huber loss, importance weights from prioritized replay."""
    HEAD_FORWARD_WITH_TORSO = True     # image networks: the Q head's forward rides on the last dense layer's launch
    HEAD_LOSS_BACKWARD_ONE_LAUNCH = True   # the head's loss (TD targets, dQ) and its backward pass as one launch
    FUSED_MLP = True        # small MLPs: the whole update / the acting step as one launch (tests flip these to cross-check)
    FUSED_ACT = True

    def __init__(self, device, obs_shape, n_actions, activation="relu", embedder="Medium",
                 middleware="Medium", learning_rate=2.5e-4, adam_beta1=0.9, adam_beta2=0.99,
                 optimizer_epsilon=1e-4, replace_mse_with_huber_loss=True, seed=0, dueling=False,
                 head_activation="relu", head_gradient_rescale=1.0, clip_gradients=None):
        self.obs_shape, self.image, self.A = tuple(obs_shape), len(obs_shape) == 3, n_actions
        self.huber = replace_mse_with_huber_loss
        self.dueling = dueling
        self.head_gradient_rescale = float(head_gradient_rescale)
        self.clip_gradients = clip_gradients
        self.params = G.FlatParams()
        self.torso, feat = build_torso(self.params, "main", obs_shape, activation, 1, embedder, middleware)
        if dueling:
            # DuelingQHead (heads/dueling_q_head.py:33-48): state-value and action-advantage streams,
            # each Dense(512, act) -> Dense(1 | A); the two fc1 layers are towers of one launch
            hn = "main/dueling_q_values_head"
            self.stream_fc = G.Dense(self.params, hn + "/fc1", feat, 512, head_activation, 2)
            self.v_out = G.Dense(self.params, hn + "/state_value/fc2", 512, 1, None, 1)
            self.a_out = G.Dense(self.params, hn + "/action_advantage/fc2", 512, n_actions, None, 1)
            self.modules = [self.torso, self.stream_fc, self.v_out, self.a_out]
        else:
            self.q_head = G.Dense(self.params, "main/q_head/dense", feat, n_actions, None, 1)
            self.modules = [self.torso, self.q_head]
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon)
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        self._fused = self._fused_mlp_setup()
        self._act = self._act_setup()

    # ---------------------------------------------------------------- fused small-MLP update (one launch)
    def _fused_mlp_setup(self):
        """obs -> Dense(relu) -> Dense(relu) -> Dense(A) Q networks (CartPole_DQN's shape) qualify for
        rlx_mlp_dqn_update: the whole learn_from_batch in ONE launch (csrc/mlp_fused.hip).
        DQNNet.FUSED_MLP = False keeps the layer-by-layer path (the tests cross-check the two)."""
        if not self.FUSED_MLP or self.image or self.dueling:
            return None
        ls = self.torso.layers
        if len(ls) != 2 or any(not isinstance(l, G.Dense) or l.act != "relu" or l.T != 1 for l in ls):
            return None
        if self.head_gradient_rescale != 1.0 or self.clip_gradients:
            return None
        d0, h1, h2 = ls[0].K, ls[0].N, ls[1].N
        if not self.lib.rlx_mlp_dqn_supported(1, d0, h1, h2, self.A):
            return None
        import ctypes
        n = ctypes.c_longlong()
        self.lib.mlp_dqn_workspace_floats(h1, h2, self.A, ctypes.byref(n))
        off = lambda name: self.params.entries[name][0]
        return dict(dims=(d0, h1, h2), ws=torch.zeros(n.value, dtype=torch.float32, device=self.device),
                    sync=torch.zeros(4, dtype=torch.int32, device=self.device),
                    offs=(off(ls[0].kname), off(ls[0].bname), off(ls[1].kname), off(ls[1].bname),
                          off(self.q_head.kname), off(self.q_head.bname)))

    def _act_setup(self):
        """the same network shape qualifies for rlx_mlp_q_act: Q(s) of a few envs + the epsilon-greedy choice as one
        launch (DQNNet.FUSED_ACT = False keeps the layer launches + rlx_egreedy)."""
        if not self.FUSED_ACT or self.image or self.dueling:
            return None
        ls = self.torso.layers
        if len(ls) != 2 or any(not isinstance(l, G.Dense) or l.act != "relu" or l.T != 1 for l in ls):
            return None
        if getattr(self.q_head, "act", None) is not None:
            return None
        off = lambda name: self.params.entries[name][0]
        return dict(dims=(ls[0].K, ls[0].N, ls[1].N),
                    offs=(off(ls[0].kname), off(ls[0].bname), off(ls[1].kname), off(ls[1].bname),
                          off(self.q_head.kname), off(self.q_head.bname)))

    def can_act_fused(self, n_env):
        a = self._act
        return a is not None and bool(self.lib.rlx_mlp_q_act_supported(int(n_env), *a["dims"], self.A))

    def q_act(self, states, n_env, explore_u, random_act, tie_rand, epsilon, q_out, actions, use_target=False):
        """q_out [n_env, A] = Q(states); actions = rlx_egreedy on them (actions None: values only)."""
        a = self._act
        w = self.target if use_target else self.params.weights
        self.lib.mlp_q_act(w, *a["offs"], states, int(n_env), *a["dims"], self.A, explore_u, random_act, tie_rand,
                           float(epsilon), q_out, actions, self.ctx.stream)

    def _fused_learn(self, obs, next_obs, B, actions, rewards, game_overs, discount, w, td_errors, double_dqn,
                     grad_scale):
        f = self._fused
        d = _rlx.MlpDqnDesc()
        d.weights, d.target_weights = self.params.weights.data_ptr(), self.target.data_ptr()
        d.adam_m, d.adam_v, d.adam_state = self.adam.m.data_ptr(), self.adam.v.data_ptr(), self.adam.state.data_ptr()
        d.states, d.next_states = obs.data_ptr(), next_obs.data_ptr()
        d.actions, d.rewards, d.game_overs = actions.data_ptr(), rewards.data_ptr(), game_overs.data_ptr()
        d.importance_weights = None if w is None else w.data_ptr()
        d.workspace, d.workspace_floats, d.sync_words = f["ws"].data_ptr(), f["ws"].numel(), f["sync"].data_ptr()
        d.loss_out, d.norm_out = self.loss.data_ptr(), self.norm.data_ptr()
        d.td_errors = None if td_errors is None else td_errors.data_ptr()
        d.status = self.status.data_ptr()
        d.off_w1, d.off_b1, d.off_w2, d.off_b2, d.off_w3, d.off_b3 = f["offs"]
        d.discount = float(discount)
        d.batch, (d.obs_dim, d.h1, d.h2), d.n_actions = int(B), f["dims"], self.A
        d.huber, d.double_dqn = int(self.huber), int(bool(double_dqn))
        a = self.adam
        d.learning_rate, d.beta1, d.beta2, d.epsilon, d.grad_scale = a.lr, a.beta1, a.beta2, a.eps, float(grad_scale)
        import ctypes
        self.lib.mlp_dqn_update(ctypes.byref(d), self.ctx.stream)
        return self.loss

    def _dueling_forward(self, feat, B, tag, weights=None, train=False):
        """-> (q Tensor [1, B, A], saved) ; saved = tensors the backward pass needs."""
        ctx = self.ctx
        shared = G.Tensor(feat.data, feat.rows, feat.cols, 0, act=feat.act)      # one input, two streams
        h = self.stream_fc.forward(ctx, shared, tag=tag, weights=weights)
        hv, ha = (h.tower(0), h.tower(1)) if train else (h.tower_view(0), h.tower_view(1))
        if self.A <= G.SMALL_N:
            v, adv = G.small_dense_forward_multi(ctx, [(self.v_out, hv), (self.a_out, ha)], tag=tag,
                                                 weights=weights)
        else:
            v = self.v_out.forward(ctx, hv, tag=tag, weights=weights)
            adv = self.a_out.forward(ctx, ha, tag=tag, weights=weights)
        qbuf = ctx.buffer("main/dueling_q_values_head/output", (1, B, self.A), tag=tag)
        self.lib.dueling_combine(v.data, adv.data, B, self.A, qbuf, ctx.stream)
        q = G.Tensor(qbuf, B, self.A, 1, grad_key=(ctx, "main/dueling_q_values_head/output", tag))
        return q, (shared, h, hv, ha, v, adv)

    def _dueling_backward(self, feat, q, saved, B):
        ctx = self.ctx
        shared, h, hv, ha, v, adv = saved
        self.lib.dueling_combine_backward(q.grad, B, self.A, v.ensure_grad(), adv.ensure_grad(), ctx.stream)
        if self.A <= G.SMALL_N and B * self.A <= 1024:
            G.small_dense_backward_multi(ctx, [(self.v_out, hv, v), (self.a_out, ha, adv)])
        else:
            self.v_out.backward(ctx, hv, v)
            self.a_out.backward(ctx, ha, adv)
        h.grad_is_dz = hv.grad_is_dz and ha.grad_is_dz
        shared.grad = feat.ensure_grad()
        self.stream_fc.backward(ctx, shared, h)
        feat.grad_is_dz = shared.grad_is_dz

    def q_values(self, obs, B, use_target=False, tag="q"):
        w = self.target if use_target else None
        acts = self.torso.forward(self.ctx, self.obs_tensor(obs, B), tag=tag, weights=w)
        if self.dueling:
            return self._dueling_forward(acts[-1], B, tag, w)[0]
        return self.q_head.forward(self.ctx, acts[-1], tag=tag, weights=w)

    def _backward_from_q(self, acts, q, saved, B):
        ctx = self.ctx
        if self.dueling:
            self._dueling_backward(acts[-1], q, saved, B)
        else:
            self.q_head.backward(ctx, acts[-1], q)
        if self.head_gradient_rescale != 1.0:         # rescale_gradient_from_head_by_factor
            g = acts[-1].grad
            self.lib.axpby(g, self.head_gradient_rescale, g, 0.0, None, g.numel(), ctx.stream)
        self.torso.backward(ctx, acts)

    def accumulate_regression(self, obs, B, targets, importance_weights=None):
        """Architecture.accumulate_gradients for this network: forward, the head loss of head.py:143-186
        against explicit [B, A] targets, backward; leaves the gradients in params.grads (clipped when
        clip_gradients is set), the loss in self.loss and tf.global_norm of the raw gradients in self.norm."""
        ctx = self.ctx
        acts = self.torso.forward(ctx, self.obs_tensor(obs, B), tag="train")
        saved = None
        if self.dueling:
            q, saved = self._dueling_forward(acts[-1], B, "train", train=True)
        else:
            q = self.q_head.forward(ctx, acts[-1], tag="train")
        self.lib.regression_loss(q.data, self.A, targets, self.A, importance_weights, B, self.A, int(self.huber),
                                 1.0, 1.0, q.ensure_grad(), self.A, self.loss, ctx.stream)
        self._backward_from_q(acts, q, saved, B)
        if not self.clip_by_global_norm():
            self.grad_norm()
        return self.loss

    def learn_from_batch(self, obs, next_obs, B, actions, rewards, game_overs, discount,
                         importance_weights=None, td_errors=None, double_dqn=False, grad_scale=1.0,
                         sync=None, states_pair=None):
        """DQNAgent.learn_from_batch (agents/dqn_agent.py:81-113), all on device."""
        ctx = self.ctx
        if self._fused is not None and sync is None and B <= 32 and obs.is_contiguous() and next_obs.is_contiguous():
            w = importance_weights
            if w is not None and w.dtype != torch.float64:
                w = w.double()
            return self._fused_learn(obs, next_obs, B, actions, rewards, game_overs, discount, w, td_errors,
                                     double_dqn, grad_scale)
        sel = self.q_values(next_obs, B, tag="next_o").data.view(B, self.A) if double_dqn else None
        if states_pair is not None and not self.dueling:
            # parallel_prediction (dqn_agent.py:86-89): online(s) and target(s') as two towers of the
            # same launches — the replay collates states / next_states into one [2, B, ...] buffer
            cols = int(np.prod(self.obs_shape))
            both = states_pair.view(2, B, cols)
            x = G.Tensor(both, B, cols, 2, u8=self.image, div=255.0 if self.image else 1.0)
            if self.HEAD_FORWARD_WITH_TORSO and self.A <= G.SMALL_N and self.q_head.T == 1 and \
                    self.torso.layers[-1].N > G.SMALL_N:
                # the Q head of both copies inside the last dense layer's split-K reduction (rlx_gemm_desc.row_heads)
                acts2, (q2,) = self.torso.forward(ctx, x, tag="pair", pair=True, row_heads=[(self.q_head, "pair")])
            else:
                acts2 = self.torso.forward(ctx, x, tag="pair", pair=True)
                q2 = self.q_head.forward(ctx, acts2[-1], tag="pair", pair=True)
            q_next = q2.data[1].view(B, self.A)
            acts = [x.tower_view(0)] + [a.tower(0) for a in acts2[1:]]
            q = q2.tower(0)
        else:
            q_next = self.q_values(next_obs, B, use_target=True, tag="next_t").data.view(B, self.A)
            acts = self.torso.forward(ctx, self.obs_tensor(obs, B), tag="train")
            if self.dueling:
                q, saved = self._dueling_forward(acts[-1], B, "train", train=True)
            else:
                q = self.q_head.forward(ctx, acts[-1], tag="train")
        dq = q.ensure_grad()
        # TD targets, |TD errors|, QHead loss and its gradient in one launch (importance weights are
        # the fp64 weights of the prioritized replay, or fp32 -> converted, or None)
        w = importance_weights
        if w is not None and w.dtype != torch.float64:
            w = w.double()
        feat = acts[-1]
        if self.HEAD_LOSS_BACKWARD_ONE_LAUNCH and not self.dueling and self.head_gradient_rescale == 1.0 and \
                self.q_head.T == 1 and self.A <= G.SMALL_N and B * self.A <= 1024 and B <= 256 and not feat.u8 and \
                feat.towers == 1:
            # TD targets, |TD errors|, loss, dQ AND the Q head's backward pass (dW, db, dz of the last dense layer) in one
            # launch (rlx_dqn_head_loss_backward): what the two calls of the else branch compute, bit for bit
            import ctypes
            prob = _rlx.SmallDenseProblem()
            G._small_backward_problem(prob, self.q_head, feat, q)
            self.lib.dqn_head_loss_backward(ctypes.byref(prob), q.data, self.A, q_next, sel, self.A, actions, rewards,
                                            game_overs, w, float(discount), B, self.A, int(self.huber), 1.0, td_errors,
                                            self.loss, self.status, ctx.stream)
            self.torso.backward(ctx, acts)
        else:
            self.lib.dqn_head_loss(q.data, self.A, q_next, sel, self.A, actions, rewards, game_overs, w,
                                   float(discount), B, self.A, int(self.huber), 1.0, dq, self.A, td_errors,
                                   None, self.A, self.loss, self.status, ctx.stream)
            self._backward_from_q(acts, q, saved if self.dueling else None, B)
        clipped = self.clip_by_global_norm()          # this worker's gradient, before it is shared
        if sync is not None:                          # data-parallel: ONE all-reduce of the flat buffer
            sync.all_reduce_sum(self.params.grads)
        self.apply_gradients(grad_scale, with_norm=not clipped)
        return self.loss
