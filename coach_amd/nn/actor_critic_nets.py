"""Device-resident networks of the continuous-control agents (DDPG, TD3, SAC).

Topologies (paths under rl_coach/):
  DDPG actor   agents/ddpg_agent.py:55-70    obs -> embedder -> FC middleware -> DDPGActorHead
               (heads/ddpg_actor_head.py:48-56: Dense(A) -> tanh -> * max_abs_range)
  DDPG critic  agents/ddpg_agent.py:36-52    merged = concat(sorted inputs: action, observation
               embedding) (general_network.py:251,270-277) -> FC middleware -> DDPGVHead Dense(1)
               (normalized-columns init, heads/ddpg_v_head.py + v_head.py:43-48)
  TD3 critic   agents/td3_agent.py:36-51     num_streams = 2 FC middlewares on the same merged
               input, TD3VHead (heads/td3_v_head.py:40-60): Dense(1) per stream, outputs
               [Q1, Q2, min(Q1,Q2), mean(Q1)], loss = sum_i mean((target - Q_i)^2)
  SAC policy   agents/soft_actor_critic_agent.py:87-100 + heads/sac_head.py:60-97
  SAC V        :57-69 + heads/v_head.py (xavier init)
  SAC Q        :72-84 + heads/sac_q_head.py:46-96: per Q_i relu(W_o s) + relu(W_a a) -> Dense -> Dense(1)
The twin streams / twin Q networks run as two towers of ONE batched GEMM launch per layer.
"""
import numpy as np
import torch

from .. import _rlx
from . import graph as G
from .networks import _NetBase


def _mlp(params, prefix, in_features, sizes, activation, towers=1, batchnorm=False):
    """batchnorm: Dense -> BatchnormActivationDropout(batchnorm, activation) per layer (embedder.py:70-78,
    fc_middleware.py with FCMiddlewareParameters(batchnorm=True))."""
    layers, feat = [], in_features
    for i, u in enumerate(sizes):
        layers.append(G.Dense(params, "%s/dense%d" % (prefix, i), feat, u, None if batchnorm else activation, towers))
        if batchnorm:
            if towers != 1:
                raise NotImplementedError("batch normalisation under several middleware streams")
            layers.append(G.BatchNorm(params, "%s/batchnorm%d" % (prefix, i), u, activation))
        feat = u
    return G.Sequential(layers), feat


def _bn_layers(*sequences):
    return [l for s in sequences for l in (s.layers if hasattr(s, "layers") else [s]) if isinstance(l, G.BatchNorm)]


class _ACBase(_NetBase):
    image = False

    def _obs(self, obs, B):
        return G.input_tensor(obs, B, self.obs_dim)


class ActorNet(_ACBase):
    """DDPG / TD3 actor: obs -> embedder -> middleware -> Dense(A) -> tanh -> * output_scale."""

    def __init__(self, device, obs_dim, action_dim, output_scale, embedder=(400,), middleware=(300,),
                 activation="relu", learning_rate=1e-4, adam_beta1=0.9, adam_beta2=0.999,
                 optimizer_epsilon=1e-8, seed=0, batchnorm=False):
        """batchnorm: DDPGActorNetworkParameters(use_batchnorm=True) (ddpg_agent.py:55-60): batch normalisation after
        the embedder's and the middleware's dense layers and between fc_mean and its tanh (ddpg_actor_head.py:48-56)."""
        self.obs_dim, self.A = obs_dim, action_dim
        self.params = G.FlatParams()
        self.emb, feat = _mlp(self.params, "actor/embedder", obs_dim, embedder, activation, batchnorm=batchnorm)
        self.mid, feat = _mlp(self.params, "actor/middleware", feat, middleware, activation, batchnorm=batchnorm)
        self.head = G.Dense(self.params, "actor/ddpg_actor_head/fc_mean", feat, action_dim,
                            None if batchnorm else "tanh")
        self.head_bn = G.BatchNorm(self.params, "actor/ddpg_actor_head/batchnorm0", action_dim, "tanh") \
            if batchnorm else None
        self.modules = [self.emb, self.mid, self.head] + ([self.head_bn] if batchnorm else [])
        self.bn_layers = _bn_layers(self.emb, self.mid) + ([self.head_bn] if batchnorm else [])
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon)
        self.scale = torch.as_tensor(np.broadcast_to(np.asarray(output_scale, dtype=np.float32),
                                                     (action_dim,)).copy(), device=device)
        self._uniform_scale = float(self.scale[0].item()) if bool((self.scale == self.scale[0]).all()) \
            else None
        if self._uniform_scale is None:
            raise NotImplementedError("per-dimension output scale is not implemented")

    def _torso(self, ctx, x, tag, w, pair):
        """embedder then middleware."""
        kw = {"pair": True} if pair else {"weights": w}
        acts = self.emb.forward(ctx, x, tag=tag, **kw)
        return acts, self.mid.forward(ctx, acts[-1], tag=tag, **kw)

    def forward(self, obs, B, use_target=False, tag="a", out=None):
        w = self.target if use_target else None
        ctx = self.ctx
        acts, acts2 = self._torso(ctx, self._obs(obs, B), tag, w, False)
        y = self.head.forward(ctx, acts2[-1], tag=tag, weights=w)
        if self.head_bn is not None:
            z = y
            y = self.head_bn.forward(ctx, z, tag=tag, weights=w)
            y.pre_bn = z
        if out is None and self._uniform_scale == 1.0:
            return y.data.view(B, self.A), (acts, acts2, y)        # tanh output IS the action: no scaling launch
        out = out if out is not None else ctx.buffer("actions", (B, self.A), tag=tag)
        self.lib.copy_2d(y.data, self.A, out, self.A, B, self.A, self._uniform_scale, ctx.stream)
        return out, (acts, acts2, y)

    def forward_pair(self, obs2, B, tag="pair"):
        """actor.parallel_prediction([(target, next_states), (online, states)]) (ddpg_agent.py:146-149,
        td3_agent.py:157-160) as ONE pass: obs2 [2, B, D] = (states, next_states); returns actions
        [2, B, A] (row 0: online mu(s), row 1: target mu(s')) and the online activations for backward."""
        if self.bn_layers:
            raise NotImplementedError("paired online / target pass with batch normalisation")
        ctx = self.ctx
        x = G.Tensor(obs2.view(2, B, self.obs_dim), B, self.obs_dim, 2)
        acts, acts2 = self._torso(ctx, x, tag, None, True)
        y = self.head.forward(ctx, acts2[-1], tag=tag, pair=True)
        if self._uniform_scale == 1.0:
            out = y.data.view(2, B, self.A)
        else:
            out = ctx.buffer("actions2", (2, B, self.A), tag=tag)
            self.lib.copy_2d(y.data, self.A, out, self.A, 2 * B, self.A, self._uniform_scale, ctx.stream)
        on = lambda ts: [ts[0].slice_towers(0, 1, with_grad=False)] + [t.slice_towers(0, 1) for t in ts[1:]]
        acts_o = on(acts)
        acts2_o = [acts_o[-1]] + [t.slice_towers(0, 1) for t in acts2[1:]]
        return out, (acts_o, acts2_o, y.slice_towers(0, 1))

    def backward(self, saved, action_grad, B, grad_scale=1.0):
        """weighted_gradients[0] with gradients_weights_ph = action_grad (ddpg_agent.py:183-186):
        d sum(action * action_grad) / d theta -> params.grads."""
        acts, acts2, y = saved
        ctx = self.ctx
        dy = y.ensure_grad()
        if action_grad is not None:          # None: the caller wrote d loss / d tanh-output into head_grad(saved)
            self.lib.copy_2d(action_grad, self.A, dy, self.A, B, self.A, self._uniform_scale * grad_scale,
                             ctx.stream)
        if self.head_bn is not None:
            self.head_bn.backward(ctx, y.pre_bn, y)
            y = y.pre_bn
        self.head.backward(ctx, acts2[-1], y)
        self.mid.backward(ctx, acts2, need_input_grad=len(self.emb.layers) > 0)
        self.emb.backward(ctx, acts)


    def head_grad(self, saved):
        """the buffer backward() reads d(loss)/d(head output) from — [B, A], gradient w.r.t. the UNSCALED tanh output."""
        return saved[2].ensure_grad()


class CriticNet(_ACBase):
    """DDPG critic (streams = 1) / TD3 twin critic (streams = 2)."""

    def __init__(self, device, obs_dim, action_dim, obs_embedder=(400,), middleware=(300,), streams=1,
                 activation="relu", head_init="normalized_columns", learning_rate=1e-3, adam_beta1=0.9,
                 adam_beta2=0.999, optimizer_epsilon=1e-8, seed=0, batchnorm=False):
        """batchnorm: DDPGCriticNetworkParameters(use_batchnorm=True) (ddpg_agent.py:36-45): batch normalisation after
        the observation embedder's and the middleware's dense layers (the Empty action embedder and the V head have
        none)."""
        self.obs_dim, self.A, self.T = obs_dim, action_dim, streams
        self.params = G.FlatParams()
        self.emb, efeat = _mlp(self.params, "critic/embedder", obs_dim, obs_embedder, activation, batchnorm=batchnorm)
        if streams > 1 and len(obs_embedder) > 0:
            raise NotImplementedError("a trainable observation embedder below several middleware "
                                      "streams needs summed input gradients")
        self.efeat = efeat
        self.merged = action_dim + efeat                    # sorted inputs: 'action' < 'observation'
        self.mid, feat = _mlp(self.params, "critic/middleware", self.merged, middleware, activation,
                              towers=streams, batchnorm=batchnorm)
        init = G.normalized_columns(1.0) if head_init == "normalized_columns" else None
        self.head = G.Dense(self.params, "critic/v_head/output", feat, 1, None, streams, init=init)
        self.modules = [self.emb, self.mid, self.head]
        self.bn_layers = _bn_layers(self.emb, self.mid)
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon)
        self.loss = torch.zeros(4, dtype=torch.float32, device=device)

    def forward(self, obs, actions, B, use_target=False, tag="q"):
        """-> q [streams, B] and the saved activations."""
        w = self.target if use_target else None
        ctx = self.ctx
        eacts = self.emb.forward(ctx, self._obs(obs, B), tag=tag, weights=w)
        merged = ctx.buffer("critic/merged", (1, B, self.merged), tag=tag)
        m2 = merged.view(B, self.merged)
        self.lib.copy_2d(actions, self.A, m2, self.merged, B, self.A, 1.0, ctx.stream)
        self.lib.copy_2d(eacts[-1].data, self.efeat, m2.data_ptr() + 4 * self.A, self.merged, B,
                         self.efeat, 1.0, ctx.stream)
        x = G.Tensor(merged, B, self.merged, 0 if self.T > 1 else 1, grad_key=(ctx, "critic/merged", tag))
        macts = self.mid.forward(ctx, x, tag=tag, weights=w)
        q = self.head.forward(ctx, macts[-1], tag=tag, weights=w)
        return q.data.view(self.T, B), (eacts, macts, q, merged)

    def forward_pair(self, obs2, actions2, B, tag="pair"):
        """Online critic on (s, a) and target critic on (s', a') in one pass: obs2 [2, B, D],
        actions2 [2, B, A].  Returns q [2, streams, B] (row 0 online, row 1 target) and the ONLINE
        activations for train_backward."""
        ctx, T = self.ctx, self.T
        x = G.Tensor(obs2.view(2, B, self.obs_dim), B, self.obs_dim, 2)
        eacts = self.emb.forward(ctx, x, tag=tag, pair=True)
        merged = ctx.buffer("critic/merged2", (2, B, self.merged), tag=tag)
        m2 = merged.view(2 * B, self.merged)
        self.lib.copy_2d(actions2, self.A, m2, self.merged, 2 * B, self.A, 1.0, ctx.stream)
        self.lib.copy_2d(eacts[-1].data, self.efeat, m2.data_ptr() + 4 * self.A, self.merged, 2 * B,
                         self.efeat, 1.0, ctx.stream)
        xm = G.Tensor(merged, B, self.merged, 2, grad_key=(ctx, "critic/merged2", tag))
        macts = self.mid.forward(ctx, xm, tag=tag, pair=True)
        q = self.head.forward(ctx, macts[-1], tag=tag, pair=True)
        # online halves: merged tower 0 (one tower per copy), activations towers [0, T)
        m_on = xm.slice_towers(0, 1)
        m_on.towers = 0 if T > 1 else 1                       # shared by the T streams, as in forward()
        macts_o = [m_on] + [t.slice_towers(0, T) for t in macts[1:]]
        eacts_o = [eacts[0].slice_towers(0, 1, with_grad=False)] + [t.slice_towers(0, 1) for t in eacts[1:]]
        return q.data.view(2, T, B), (eacts_o, macts_o, q.slice_towers(0, T), merged[0:1])

    def merged_pair_buffer(self, B, tag="pair"):
        """[2, B, A + D]: (s, a) rows for the online pass, (s', a') rows for the target pass (no observation
        embedder: the merged input is concat(action, observation))."""
        assert len(self.emb.layers) == 0
        return self.ctx.buffer("critic/merged2", (2, B, self.merged), tag=tag)

    def merged_buffer(self, B, tag):
        return self.ctx.buffer("critic/merged", (1, B, self.merged), tag=tag)

    def forward_pair_merged(self, merged, B, tag="pair"):
        """forward_pair on inputs already merged by rlx_ac_merge_inputs (merged_pair_buffer)."""
        ctx, T = self.ctx, self.T
        xm = G.Tensor(merged, B, self.merged, 2, grad_key=(ctx, "critic/merged2", tag))
        macts = self.mid.forward(ctx, xm, tag=tag, pair=True)
        q = self.head.forward(ctx, macts[-1], tag=tag, pair=True)
        m_on = xm.slice_towers(0, 1)
        m_on.towers = 0 if T > 1 else 1
        macts_o = [m_on] + [t.slice_towers(0, T) for t in macts[1:]]
        return q.data.view(2, T, B), ([], macts_o, q.slice_towers(0, T), merged[0:1])

    def forward_merged(self, merged, B, use_target=False, tag="q"):
        """forward on an input already merged: merged [1, B, A + D] from merged_buffer(B, tag)."""
        w = self.target if use_target else None
        ctx = self.ctx
        x = G.Tensor(merged, B, self.merged, 0 if self.T > 1 else 1, grad_key=(ctx, "critic/merged", tag))
        macts = self.mid.forward(ctx, x, tag=tag, weights=w)
        q = self.head.forward(ctx, macts[-1], tag=tag, weights=w)
        return q.data.view(self.T, B), ([], macts, q, merged)

    def train_backward(self, saved, targets, B, losses_done=False):
        """accumulate_gradients for the critic loss sum_i mean((target - Q_i)^2).
        losses_done: rlx_ac_critic_losses already wrote the loss and d loss / d Q_i into q's gradient."""
        eacts, macts, q, merged = saved
        ctx = self.ctx
        dq = q.ensure_grad()
        if not losses_done:
            for t in range(self.T):
                self.lib.regression_loss(q.data[t], 1, targets, 1, None, B, 1, 0, 1.0, 1.0, dq[t], 1,
                                         self.loss[t:t + 1], ctx.stream)
        self.head.backward(ctx, macts[-1], q)
        need_in = len(self.emb.layers) > 0
        self.mid.backward(ctx, macts, need_input_grad=need_in)
        if need_in:
            g = macts[0].grad.view(B, self.merged)
            eg = eacts[-1].ensure_grad().view(B, self.efeat)
            self.lib.copy_2d(g.data_ptr() + 4 * self.A, self.merged, eg, self.efeat, B, self.efeat, 1.0,
                             ctx.stream)
            self.emb.backward(ctx, eacts)

    def action_gradient(self, saved, B, out, scale=1.0):
        """gradients_wrt_inputs[...]['action'] of mean_b(Q_1) (ddpg_agent.py:171-173,
        td3_agent.py:194-198): backward through stream 0 only, no weight gradients."""
        eacts, macts, q, merged = saved
        ctx = self.ctx
        dq = q.ensure_grad()
        # d mean_b(Q_1) / d Q_1 = 1/B: a constant the backward pass only reads (the head has no activation), so the
        # cached gradient buffer of this pass is filled once per batch size
        key = ("agrad_filled", dq.data_ptr(), B)
        if ctx.cache.get(key) is None:
            dq[0].fill_(1.0 / B)
            ctx.cache[key] = True
        x1 = G.Tensor(merged, B, self.merged, 1, grad_key=(ctx, "critic/merged", "agrad"))
        acts = [x1] + [a.tower(0) for a in macts[1:]]
        self.head.backward(ctx, acts[-1], q.tower(0), need_dx=True, t0=0, nt=1, need_dw=False)
        self.mid.backward(ctx, acts, need_input_grad=True, t0=0, nt=1, need_dw=False)
        g = x1.grad.view(B, self.merged)
        self.lib.copy_2d(g, self.merged, out, self.A, B, self.A, float(scale), ctx.stream)
        return out


class SACPolicyNet(_ACBase):
    def __init__(self, device, obs_dim, action_dim, embedder=(256,), middleware=(256,),
                 learning_rate=3e-4, adam_beta1=0.9, adam_beta2=0.99, optimizer_epsilon=1e-4, seed=0):
        self.obs_dim, self.A = obs_dim, action_dim
        self.params = G.FlatParams()
        self.emb, feat = _mlp(self.params, "policy/embedder", obs_dim, embedder, "relu")
        self.mid, feat = _mlp(self.params, "policy/middleware", feat, middleware, "relu")
        self.head = G.Dense(self.params, "policy/sac_policy_head/policy_mu_logsig", feat, 2 * action_dim, None)
        self.modules = [self.emb, self.mid, self.head]
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon,
                     has_target=False)

    def forward(self, obs, B, normals, tag="pi", want=("actions", "logprob")):
        ctx = self.ctx
        acts = self.emb.forward(ctx, self._obs(obs, B), tag=tag)
        acts2 = self.mid.forward(ctx, acts[-1], tag=tag)
        y = self.head.forward(ctx, acts2[-1], tag=tag)
        o = {k: ctx.buffer("sac/" + k, (B, self.A) if k != "logprob" else (B,), tag=tag)
             for k in ("mean", "log_std", "raw_actions", "actions", "logprob")}
        self.lib.sac_policy_head(y.data, 2 * self.A, normals, B, self.A, o["mean"], o["log_std"],
                                 o["raw_actions"], o["actions"], o["logprob"], ctx.stream)
        return o, (acts, acts2, y, normals)

    def resample(self, saved, B, normals, tag):
        """Outputs of another sess.run of the head on the SAME torso output with fresh noise."""
        y, ctx = saved[2], self.ctx
        o = {k: ctx.buffer("sac/" + k, (B, self.A) if k != "logprob" else (B,), tag=tag)
             for k in ("mean", "log_std", "raw_actions", "actions", "logprob")}
        self.lib.sac_policy_head(y.data, 2 * self.A, normals, B, self.A, o["mean"], o["log_std"],
                                 o["raw_actions"], o["actions"], o["logprob"], ctx.stream)
        return o

    def head_gradient(self, saved, B, normals=None, logprob_mean_weight=0.0, action_weights=None,
                      action_weight_scale=1.0, accumulate=False):
        """d(...)/d mu_logsig of one pass (its own noise draw) into the head's gradient buffer."""
        y = saved[2]
        dy = y.ensure_grad()
        self.lib.sac_policy_head_backward(y.data, 2 * self.A, normals if normals is not None else saved[3],
                                          B, self.A, float(logprob_mean_weight), action_weights,
                                          float(action_weight_scale), dy, 2 * self.A, int(accumulate),
                                          self.ctx.stream)

    def backward_torso(self, saved):
        acts, acts2, y, _ = saved
        ctx = self.ctx
        self.head.backward(ctx, acts2[-1], y)
        self.mid.backward(ctx, acts2, need_input_grad=len(self.emb.layers) > 0)
        self.emb.backward(ctx, acts)

    def backward(self, saved, B, logprob_mean_weight=0.0, action_weights=None, action_weight_scale=1.0):
        acts, acts2, y, normals = saved
        ctx = self.ctx
        self.head_gradient(saved, B, None, logprob_mean_weight, action_weights, action_weight_scale)
        self.head.backward(ctx, acts2[-1], y)
        self.mid.backward(ctx, acts2, need_input_grad=len(self.emb.layers) > 0)
        self.emb.backward(ctx, acts)


class SACValueNet(_ACBase):
    def __init__(self, device, obs_dim, embedder=(256,), middleware=(256,), learning_rate=3e-4,
                 adam_beta1=0.9, adam_beta2=0.99, optimizer_epsilon=1e-4, seed=0):
        self.obs_dim = obs_dim
        self.params = G.FlatParams()
        self.emb, feat = _mlp(self.params, "v/embedder", obs_dim, embedder, "relu")
        self.mid, feat = _mlp(self.params, "v/middleware", feat, middleware, "relu")
        self.head = G.Dense(self.params, "v/v_values_head/output", feat, 1, None)      # xavier
        self.modules = [self.emb, self.mid, self.head]
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon)
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)

    def forward(self, obs, B, use_target=False, tag="v"):
        w = self.target if use_target else None
        ctx = self.ctx
        acts = self.emb.forward(ctx, self._obs(obs, B), tag=tag, weights=w)
        acts2 = self.mid.forward(ctx, acts[-1], tag=tag, weights=w)
        v = self.head.forward(ctx, acts2[-1], tag=tag, weights=w)
        return v.data.view(B), (acts, acts2, v)

    def forward_pair(self, obs2, B, tag="pair"):
        """V_online(s) (for the training pass) and V_target(s') (for the Q targets) in one pass."""
        ctx = self.ctx
        x = G.Tensor(obs2.view(2, B, self.obs_dim), B, self.obs_dim, 2)
        acts = self.emb.forward(ctx, x, tag=tag, pair=True)
        acts2 = self.mid.forward(ctx, acts[-1], tag=tag, pair=True)
        v = self.head.forward(ctx, acts2[-1], tag=tag, pair=True)
        acts_o = [acts[0].slice_towers(0, 1, with_grad=False)] + [t.slice_towers(0, 1) for t in acts[1:]]
        acts2_o = [acts_o[-1]] + [t.slice_towers(0, 1) for t in acts2[1:]]
        return v.data.view(2, B), (acts_o, acts2_o, v.slice_towers(0, 1))

    def train_backward(self, saved, targets, B):
        acts, acts2, v = saved
        ctx = self.ctx
        dv = v.ensure_grad()
        self.lib.regression_loss(v.data, 1, targets, 1, None, B, 1, 0, 1.0, 1.0, dv, 1, self.loss, ctx.stream)
        self.head.backward(ctx, acts2[-1], v)
        self.mid.backward(ctx, acts2, need_input_grad=len(self.emb.layers) > 0)
        self.emb.backward(ctx, acts)


class SACQNet(_ACBase):
    """Twin Q head (heads/sac_q_head.py:46-96) as two towers; empty embedder / middleware."""

    def __init__(self, device, obs_dim, action_dim, layers=(256, 256), learning_rate=3e-4,
                 adam_beta1=0.9, adam_beta2=0.99, optimizer_epsilon=1e-4, seed=0):
        self.obs_dim, self.A = obs_dim, action_dim
        self.params = G.FlatParams()
        P = self.params
        self.obs_fc = G.Dense(P, "q/q_head/obs_fc", obs_dim, layers[0], "relu", 2)
        self.act_fc = G.Dense(P, "q/q_head/act_fc", action_dim, layers[0], "relu", 2)
        self.fcs, feat = [], layers[0]
        for i, u in enumerate(layers[1:]):
            self.fcs.append(G.Dense(P, "q/q_head/fc%d" % (i + 1), feat, u, "relu", 2))
            feat = u
        self.out = G.Dense(P, "q/q_head/q_output", feat, 1, None, 2)
        self.modules = [self.obs_fc, self.act_fc] + self.fcs + [self.out]
        self.h0 = layers[0]
        self._finish(device, seed, learning_rate, adam_beta1, adam_beta2, optimizer_epsilon,
                     has_target=False)
        self.loss = torch.zeros(3, dtype=torch.float32, device=device)       # q1 loss, q2 loss, their sum

    def forward(self, obs, actions, B, tag="q"):
        ctx = self.ctx
        # both Q towers read the SAME action batch: a shared (tower-less) input, like the observation
        xa = G.Tensor(actions.view(1, B, self.A), B, self.A, 0, grad_key=(ctx, "q/actions", tag))
        ho = self.obs_fc.forward(ctx, self._obs(obs, B), tag=tag)
        ha = self.act_fc.forward(ctx, xa, tag=tag)
        h = ctx.buffer("q/sum", (2, B, self.h0), tag=tag)
        self.lib.axpby(h, 1.0, ho.data, 1.0, ha.data, h.numel(), ctx.stream)
        acts = [G.Tensor(h, B, self.h0, 2, grad_key=(ctx, "q/sum", tag))]
        for l in self.fcs:
            acts.append(l.forward(ctx, acts[-1], tag=tag))
        q = self.out.forward(ctx, acts[-1], tag=tag)
        return q.data.view(2, B), (ho, ha, xa, acts, q, obs)

    def _backward(self, saved, B, need_dw, need_action_grad):
        ho, ha, xa, acts, q, obs = saved
        ctx = self.ctx
        kw = {"need_dw": need_dw}
        self.out.backward(ctx, acts[-1], q, **kw)
        for i in reversed(range(len(self.fcs))):
            self.fcs[i].backward(ctx, acts[i], acts[i + 1], **kw)
        dh = acts[0].grad
        # h = relu(obs_fc) + relu(act_fc): each branch multiplies dh by ITS relu mask in place, so the two need
        # separate buffers only when both are propagated (training pass)
        if need_dw:
            ho.ensure_grad().copy_(dh)
            self.obs_fc.backward(ctx, self._obs(obs, B), ho, need_dx=False)
        ha.grad = dh
        self.act_fc.backward(ctx, xa, ha, need_dx=need_action_grad, **kw)

    def train_backward(self, saved, targets, B):
        """loss = 0.5*mean((q1 - y)^2) + 0.5*mean((q2 - y)^2)  (sac_q_head.py:91-95)."""
        q = saved[4]
        dq = q.ensure_grad()
        if targets is not None:        # None: rlx_ac_critic_losses already wrote the losses and dq
            for t in range(2):
                self.lib.regression_loss(q.data[t], 1, targets, 1, None, B, 1, 0, 0.5, 1.0, dq[t], 1,
                                         self.loss[t:t + 1], self.ctx.stream)
        self._backward(saved, B, True, False)

    def action_gradient(self, saved, B, out, dq_done=False):
        """gradients_wrt_inputs[1]['output_0_0'] = d mean(min(q1,q2)) / d action (:216-217) -> out [B, A].
        dq_done: the caller (rlx_sac_min_targets) already wrote d mean(min) / d q_i into q's gradient."""
        xa, q = saved[2], saved[4]
        ctx = self.ctx
        dq = q.ensure_grad()
        if not dq_done:
            qv = q.data.view(2, B)
            self.lib.min_pair(qv[0], qv[1], None, dq.view(2, B)[0], dq.view(2, B)[1], 1.0 / B, B, ctx.stream)
        xa.grad = out.view(1, B, self.A)            # the shared input's gradient sums the towers (Dense.backward)
        self._backward(saved, B, False, True)
        return out
