"""Continuous-control networks and update steps on the CPU — numpy fp32 restatement of
rl_coach/agents/{ddpg,td3,soft_actor_critic}_agent.py learn_from_batch around the TF heads
(heads/ddpg_actor_head.py:48-56, ddpg_v_head.py, td3_v_head.py:40-60, sac_head.py:60-97,
sac_q_head.py:46-96, v_head.py:43-52).  PARITY UNPINNED for TF's op-level rounding (TensorFlow is
absent); cross-checked against torch autograd in tests/test_ac_nets.py.  The update LOGIC of
ddpg_update / td3_update / sac_update (pass order, targets, gradient signs and weights, actor cadence) is
pinned: tests/test_update_pins.py compares them with fixtures made by running the reference's own
learn_from_batch on oracle-backed stand-ins (tests/golden/_oracle_backend.py).  TEST INFRASTRUCTURE ONLY.
"""
import copy

import numpy as np

from . import nn as N
from . import targets as T

F32 = np.float32
EPS32 = np.finfo(np.float32).eps


def _batchnorm(arrays, name, activation):
    return N.BatchNorm(*(arrays["%s/%s" % (name, k)][0].copy() for k in ("gamma", "beta", "moving_mean", "moving_variance")),
                       activation=activation)


def _dense_list(arrays, prefix, tower, activation):
    """[(name, layer)]; a "<prefix>/batchnorm<i>/gamma" entry marks Dense -> BatchNorm -> activation (the
    use_batchnorm=True networks of agents/ddpg_agent.py:37-60)."""
    out, i = [], 0
    while "%s/dense%d/kernel" % (prefix, i) in arrays:
        n, bn = "%s/dense%d" % (prefix, i), "%s/batchnorm%d" % (prefix, i)
        has_bn = bn + "/gamma" in arrays
        out.append((n, N.Dense(arrays[n + "/kernel"][tower].copy(), arrays[n + "/bias"][tower].copy(),
                               None if has_bn else activation)))
        if has_bn:
            out.append((bn, _batchnorm(arrays, bn, activation)))
        i += 1
    return out


def _tensors(l):
    """(suffix, weight attribute, gradient attribute or None) of a layer's variables."""
    if isinstance(l, N.BatchNorm):
        return [("gamma", "gamma", "dgamma"), ("beta", "beta", "dbeta"), ("moving_mean", "moving_mean", None),
                ("moving_variance", "moving_var", None)]
    return [("kernel", "W", "dW"), ("bias", "b", "db")]


class _Net:
    """named layers + per-tensor TF1 Adam; `layers` = [(name, tower, Dense)]."""

    def _setup(self, lr, adam):
        self.adam = N.PerTensorAdam(lr, *adam)

    def apply(self, grad_scale=1.0):
        for name, tower, l in self.layers:
            if isinstance(l, N.BatchNorm):
                l.commit()                      # UPDATE_OPS run before the optimizer step, on the pre-update forward pass
            for suffix, w, g in _tensors(l):
                if g is not None:
                    self.adam.step((name, tower, suffix[0]), getattr(l, w), getattr(l, g), grad_scale)

    def weights(self):
        out = {}
        for name, tower, l in self.layers:
            for suffix, w, _ in _tensors(l):
                out.setdefault("%s/%s" % (name, suffix), {})[tower] = getattr(l, w)
        return out

    def grads(self):
        out = {}
        for name, tower, l in self.layers:
            for suffix, _, g in _tensors(l):
                if g is not None:
                    out.setdefault("%s/%s" % (name, suffix), {})[tower] = getattr(l, g)
        return out

    def global_norm(self):
        s = 0.0
        for _, _, l in self.layers:
            for _, _, g in _tensors(l):
                if g is not None:
                    s += float(np.sum(getattr(l, g).astype(np.float64) ** 2))
        return np.sqrt(s)

    def clone_target(self):
        self.target_layers = copy.deepcopy([l for _, _, l in self.layers])

    def mix_target(self, rate):
        """NetworkWrapper.update_target_network copies the TRAINABLE variables (architecture.py: self.weights =
        tf.trainable_variables): a target network's moving statistics stay at their initial values."""
        from .optim import mix_weights
        for lt, (_, _, lo) in zip(self.target_layers, self.layers):
            for _, w, g in _tensors(lo):
                if g is not None:
                    getattr(lt, w)[...] = mix_weights(getattr(lt, w), getattr(lo, w), F32(rate))

    def set_is_training(self, state):
        """is_training of the online AND the target graph (network_wrapper.py:215-224)."""
        for l in [l for _, _, l in self.layers] + list(getattr(self, "target_layers", [])):
            if isinstance(l, N.BatchNorm):
                l.training = bool(state)


class ActorOracle(_Net):
    def __init__(self, arrays, scale, activation="relu", lr=1e-4, adam=(0.9, 0.999, 1e-8)):
        emb = _dense_list(arrays, "actor/embedder", 0, activation)
        mid = _dense_list(arrays, "actor/middleware", 0, activation)
        hn, hbn = "actor/ddpg_actor_head/fc_mean", "actor/ddpg_actor_head/batchnorm0"
        has_bn = hbn + "/gamma" in arrays
        head = N.Dense(arrays[hn + "/kernel"][0].copy(), arrays[hn + "/bias"][0].copy(), None if has_bn else "tanh")
        self.layers = [(n, 0, l) for n, l in emb + mid] + [(hn, 0, head)]
        if has_bn:                               # fc_mean -> batchnorm -> tanh (ddpg_actor_head.py:48-56)
            self.layers.append((hbn, 0, _batchnorm(arrays, hbn, "tanh")))
        self.scale = F32(scale)
        self._setup(lr, adam)
        self.clone_target()

    def forward(self, obs, target=False):
        x = np.asarray(obs, dtype=F32)
        ls = self.target_layers if target else [l for _, _, l in self.layers]
        for l in ls:
            x = l.forward(x)
        return x * self.scale

    def backward(self, action_grad):
        """d sum(action * action_grad) / d theta (run right after forward(obs) on the online net)."""
        dy = np.asarray(action_grad, dtype=F32) * self.scale
        for _, _, l in reversed(self.layers):
            dy = l.backward(dy)


class CriticOracle(_Net):
    """DDPG (streams=1) / TD3 (streams=2) critic; merged input = concat(action, obs embedding)."""

    def __init__(self, arrays, streams=1, activation="relu", lr=1e-3, adam=(0.9, 0.999, 1e-8)):
        self.T = streams
        self.emb = _dense_list(arrays, "critic/embedder", 0, activation)
        self.mid = [_dense_list(arrays, "critic/middleware", t, activation) for t in range(streams)]
        hn = "critic/v_head/output"
        self.heads = [N.Dense(arrays[hn + "/kernel"][t].copy(), arrays[hn + "/bias"][t].copy()) for t in range(streams)]
        self.layers = [(n, 0, l) for n, l in self.emb]
        for t in range(streams):
            self.layers += [(n, t, l) for n, l in self.mid[t]] + [(hn, t, self.heads[t])]
        self._setup(lr, adam)
        self.clone_target()

    def _split(self, layers):
        ne = len(self.emb)
        emb, rest, per = layers[:ne], layers[ne:], len(self.mid[0]) + 1
        return emb, [rest[t * per:(t + 1) * per] for t in range(self.T)]

    def forward(self, obs, actions, target=False):
        ls = self.target_layers if target else [l for _, _, l in self.layers]
        emb, streams = self._split(ls)
        x = np.asarray(obs, dtype=F32)
        for l in emb:
            x = l.forward(x)
        self.merged = np.concatenate([np.asarray(actions, dtype=F32), x], axis=1)
        qs = []
        for st in streams:
            h = self.merged
            for l in st:
                h = l.forward(h)
            qs.append(h[:, 0])
        return np.stack(qs)                     # [T, B]

    def train_backward(self, targets):
        """after forward(obs, actions) on the online net: loss = sum_i mean((y - Q_i)^2)."""
        emb, streams = self._split([l for _, _, l in self.layers])
        y = np.asarray(targets, dtype=F32).reshape(-1)
        B = y.shape[0]
        losses, dmerged = [], 0
        for st in streams:
            q = st[-1].y[:, 0]
            losses.append(F32(np.mean((y - q) ** 2, dtype=F32)))
            d = (F32(2.0) * (q - y) / F32(B)).reshape(-1, 1).astype(F32)
            for l in reversed(st):
                d = l.backward(d)
            dmerged = dmerged + d
        if emb:
            d = dmerged[:, self.merged.shape[1] - emb[-1].y.shape[1]:]
            for l in reversed(emb):
                d = l.backward(d)
        return losses

    def action_gradient(self, n_act):
        """d mean_b(Q_1) / d action after forward(obs, actions) on the online net."""
        _, streams = self._split([l for _, _, l in self.layers])
        B = self.merged.shape[0]
        d = np.full((B, 1), 1.0 / B, dtype=F32)
        for l in reversed(streams[0]):
            if isinstance(l, N.BatchNorm):
                d = l.backward(d, weights=False)
            else:
                d = (d * N.act_grad(l.y, l.act)) @ l.W.T       # no weight gradients
        return d[:, :n_act]


def ddpg_update(actor, critic, batch, discount=0.99, clip=None, non_zero_terminal=False):
    """DDPGAgent.learn_from_batch (agents/ddpg_agent.py:137-195)."""
    s, a, r, done, ns = batch
    actor.set_is_training(True)                # Agent.train brackets the update (agent.py:716,779): batch statistics in
    critic.set_is_training(True)               # every pass below, target networks included
    next_actions = actor.forward(ns, target=True)
    q_next = critic.forward(ns, next_actions, target=True)[0]
    y = T.ac_td_targets(r, done, q_next[:, None], discount, non_zero_terminal, clip).astype(F32)[:, 0]
    actions_mean = actor.forward(s)
    critic.forward(s, actions_mean)
    g = critic.action_gradient(a.shape[1])
    critic.forward(s, a)                       # the batch the critic's moving averages see (:178-180)
    losses = critic.train_backward(y)
    norm = critic.global_norm()
    critic.apply()
    actor.forward(s)                           # ... and the actor's (additional_inputs = the states, :188-193)
    actor.backward(-g)
    actor.apply()
    actor.set_is_training(False)
    critic.set_is_training(False)
    return dict(loss=float(sum(losses)), targets=y, norm=norm, action_grad=g)


def td3_update(actor, critic, batch, noise, training_iteration, low, high, discount=0.99,
               noise_clipping=0.5, policy_every=2):
    """TD3Agent.learn_from_batch (agents/td3_agent.py:148-209); `noise` = np.random.normal(0, policy_noise, shape)."""
    s, a, r, done, ns = batch
    next_actions = actor.forward(ns, target=True)
    actions_mean = actor.forward(s)
    sm = T.td3_smooth_actions(next_actions, noise, noise_clipping, low, high).astype(F32)
    q_next = critic.forward(ns, sm, target=True)
    q_min = np.minimum(q_next[0], q_next[1])
    y = T.ac_td_targets(r, done, q_min[:, None], discount).astype(F32)[:, 0]
    critic.forward(s, a)
    losses = critic.train_backward(y)
    norm = critic.global_norm()
    critic.apply()
    if training_iteration % policy_every == 0:
        critic.forward(s, actions_mean)
        g = critic.action_gradient(a.shape[1])
        actor.forward(s)
        actor.backward(-g)
        actor.apply()
    return dict(loss=float(sum(losses)), targets=y, norm=norm)


# ------------------------------------------------------------------------------------------ SAC
class SACPolicyOracle(_Net):
    def __init__(self, arrays, lr=3e-4, adam=(0.9, 0.99, 1e-4)):
        emb = _dense_list(arrays, "policy/embedder", 0, "relu")
        mid = _dense_list(arrays, "policy/middleware", 0, "relu")
        hn = "policy/sac_policy_head/policy_mu_logsig"
        head = N.Dense(arrays[hn + "/kernel"][0].copy(), arrays[hn + "/bias"][0].copy())
        self.layers = [(n, 0, l) for n, l in emb + mid] + [(hn, 0, head)]
        self._setup(lr, adam)

    def forward(self, obs, normals):
        x = np.asarray(obs, dtype=F32)
        for _, _, l in self.layers:
            x = l.forward(x)
        A = x.shape[1] // 2
        self.A, self.eps = A, np.asarray(normals).astype(F32)
        mu, ls_raw = x[:, :A], x[:, A:]
        ls = np.clip(ls_raw, F32(-20), F32(2))
        sd = np.exp(ls)
        raw = mu + sd * self.eps
        t = np.tanh(raw)
        z = (raw - mu) / sd
        lp = np.sum(F32(-0.5) * z * z - ls - F32(0.91893853320467274178), axis=1) - \
            np.sum(np.log(F32(1) - t * t + EPS32), axis=1)
        self.cache = (mu, ls_raw, ls, sd, raw, t)
        return dict(mean=mu, log_std=ls, raw_actions=raw, actions=t, logprob=lp.astype(F32))

    def backward(self, logp_weight=0.0, action_weights=None, action_weight_scale=1.0):
        mu, ls_raw, ls, sd, raw, t = self.cache
        B = mu.shape[0]
        one_m = F32(1) - t * t
        w = F32(logp_weight / B)
        g_raw = w * (F32(2) * t * one_m / (one_m + EPS32))
        if action_weights is not None:
            g_raw = g_raw + F32(action_weight_scale) * np.asarray(action_weights, dtype=F32) * one_m
        d_mu = g_raw
        d_ls = g_raw * sd * self.eps - w
        d_ls = np.where((ls_raw >= -20) & (ls_raw <= 2), d_ls, F32(0))
        dy = np.concatenate([d_mu, d_ls], axis=1).astype(F32)
        for _, _, l in reversed(self.layers):
            dy = l.backward(dy)


class SACValueOracle(_Net):
    def __init__(self, arrays, lr=3e-4, adam=(0.9, 0.99, 1e-4)):
        emb = _dense_list(arrays, "v/embedder", 0, "relu")
        mid = _dense_list(arrays, "v/middleware", 0, "relu")
        hn = "v/v_values_head/output"
        head = N.Dense(arrays[hn + "/kernel"][0].copy(), arrays[hn + "/bias"][0].copy())
        self.layers = [(n, 0, l) for n, l in emb + mid] + [(hn, 0, head)]
        self._setup(lr, adam)
        self.clone_target()

    def forward(self, obs, target=False):
        x = np.asarray(obs, dtype=F32)
        for l in (self.target_layers if target else [l for _, _, l in self.layers]):
            x = l.forward(x)
        return x[:, 0]

    def train(self, obs, targets):
        v = self.forward(obs)
        y = np.asarray(targets, dtype=F32)
        loss = F32(np.mean((y - v) ** 2, dtype=F32))
        d = (F32(2) * (v - y) / F32(len(y))).reshape(-1, 1).astype(F32)
        for _, _, l in reversed(self.layers):
            d = l.backward(d)
        self.apply()
        return loss


class SACQOracle(_Net):
    def __init__(self, arrays, lr=3e-4, adam=(0.9, 0.99, 1e-4)):
        P = "q/q_head/"
        self.towers = []
        self.layers = []
        for t in range(2):
            mk = lambda n, act: N.Dense(arrays[P + n + "/kernel"][t].copy(), arrays[P + n + "/bias"][t].copy(), act)
            fcs, i = [], 1
            while P + "fc%d/kernel" % i in arrays:
                fcs.append(("fc%d" % i, mk("fc%d" % i, "relu"))); i += 1
            tw = dict(obs=mk("obs_fc", "relu"), act=mk("act_fc", "relu"), fcs=fcs, out=mk("q_output", None))
            self.towers.append(tw)
            self.layers += [(P + "obs_fc", t, tw["obs"]), (P + "act_fc", t, tw["act"])] + \
                [(P + n, t, l) for n, l in fcs] + [(P + "q_output", t, tw["out"])]
        self._setup(lr, adam)

    def forward(self, obs, actions):
        s, a = np.asarray(obs, dtype=F32), np.asarray(actions, dtype=F32)
        qs = []
        for tw in self.towers:
            h = tw["obs"].forward(s) + tw["act"].forward(a)
            for _, l in tw["fcs"]:
                h = l.forward(h)
            qs.append(tw["out"].forward(h)[:, 0])
        return np.stack(qs)

    def _backward(self, dq):                        # dq [2, B]
        da = 0
        for tw, d in zip(self.towers, dq):
            d = tw["out"].backward(d.reshape(-1, 1).astype(F32))
            for _, l in reversed(tw["fcs"]):
                d = l.backward(d)
            tw["obs"].backward(d)
            da = da + tw["act"].backward(d)
        return da

    def action_gradient(self, q):
        B = q.shape[1]
        first = q[0] <= q[1]
        dq = np.stack([np.where(first, F32(1.0 / B), F32(0)), np.where(first, F32(0), F32(1.0 / B))]).astype(F32)
        return self._backward(dq)

    def train(self, obs, actions, targets):
        q = self.forward(obs, actions)
        y = np.asarray(targets, dtype=F32)
        B = len(y)
        losses = [F32(0.5) * F32(np.mean((q[t] - y) ** 2, dtype=F32)) for t in range(2)]
        self._backward(np.stack([(q[t] - y) / F32(B) for t in range(2)]).astype(F32))
        norm = self.global_norm()
        self.apply()
        return losses, norm


def sac_update(pol, q, v, batch, normals, discount=0.99, resample=True):
    """SoftActorCriticAgent.learn_from_batch (agents/soft_actor_critic_agent.py:168-280);
    normals [3, B, A] = the noise of the three policy sess.run passes."""
    s, a, r, done, ns = batch
    o = pol.forward(s, normals[0])
    qv = q.forward(s, o["actions"])
    log_target = np.minimum(qv[0], qv[1])
    dq_da = q.action_gradient(qv)
    if resample:
        pol.forward(s, normals[1])
        pol.backward(logp_weight=1.0)
        g_a = {k: {t: g.copy() for t, g in d.items()} for k, d in pol.grads().items()}
        pol.forward(s, normals[2])
        pol.backward(action_weights=dq_da)
        for name, tower, l in pol.layers:
            l.dW = g_a[name + "/kernel"][tower] - l.dW
            l.db = g_a[name + "/bias"][tower] - l.db
    else:
        pol.backward(logp_weight=1.0, action_weights=dq_da, action_weight_scale=-1.0)
    pol.apply()
    value_targets = (log_target - o["logprob"]).astype(F32)
    v_loss = v.train(s, value_targets)
    v_next = v.forward(ns, target=True)
    y = T.ac_td_targets(r, done, v_next[:, None], discount).astype(F32)[:, 0]
    q_losses, norm = q.train(s, a, y)
    return dict(loss=float(sum(q_losses)), v_loss=float(v_loss), value_targets=value_targets, td_targets=y,
                dq_da=dq_da, logprob=o["logprob"], norm=norm, actions=o["actions"])
