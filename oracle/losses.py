"""Head losses — numpy restatement of the TensorFlow head graphs (fp32 like the reference).

PARITY UNPINNED for TF's op-level rounding: TensorFlow 1.x is a pip dependency of the reference
(setup.py:69), absent from /root/reference and not installable here.  Formulas follow
rl_coach/architectures/tensorflow_components/heads/{head.py:143-186, ppo_head.py:52-116, q_head.py,
v_head.py:43-52}; they ARE pinned by the known-answer values of the reference's MXNet twin tests
(rl_coach/tests/architectures/mxnet_components/heads/test_ppo_head.py:141-183,363-376), see
tests/test_losses.py.
"""
import numpy as np


def mse(target, out):
    return (out - target) ** 2                       # tf.losses.mean_squared_error, Reduction.NONE


def huber(target, out, delta=1.0):
    e = out - target                                 # tf.losses.huber_loss
    a = np.abs(e)
    quad = np.minimum(a, delta)
    return 0.5 * quad ** 2 + delta * (a - quad)


def regression_head_loss(out, target, weights=None, kind="mse", loss_weight=1.0):
    """Head.set_loss (head.py:172-181): mean_b(loss_weight * w_b * sum_dims l).  Returns
    (loss, d loss / d out)."""
    out = np.asarray(out, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    B = out.shape[0]
    w = np.full(B, loss_weight, dtype=np.float32) if weights is None else \
        (loss_weight * np.asarray(weights, dtype=np.float32))
    l = mse(target, out) if kind == "mse" else huber(target, out)
    loss = np.mean(w * l.reshape(B, -1).sum(1))
    e = out - target
    g = 2 * e if kind == "mse" else np.clip(e, -1, 1)
    return np.float32(loss), (w.reshape((B,) + (1,) * (out.ndim - 1)) * g / B).astype(np.float32)


def log_softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def categorical_log_prob(probs, actions):
    """tf Categorical(probs=p).log_prob(a): logits = log p, renormalised by log_softmax."""
    lp = log_softmax(np.log(probs))
    return np.take_along_axis(lp, np.asarray(actions)[:, None], axis=1)[:, 0]


def categorical_entropy(probs):
    lp = log_softmax(np.log(probs))
    return -(np.exp(lp) * lp).sum(-1)


def categorical_kl(p, q):
    lp, lq = log_softmax(np.log(p)), log_softmax(np.log(q))
    return (np.exp(lp) * (lp - lq)).sum(-1)


def ppo_discrete_loss(logits, actions, advantages, old_probs, clip_eps, beta):
    """PPOHead (ppo_head.py:52-116) from the policy_fc logits.  Returns dict with the loss terms,
    the fetches (kl, entropy, ratios) and d total / d logits."""
    logits = np.asarray(logits, dtype=np.float32)
    B, A = logits.shape
    lp_all = log_softmax(logits)
    p = np.exp(lp_all)
    lpo_all = log_softmax(np.log(np.asarray(old_probs, dtype=np.float32)))
    a = np.asarray(actions)
    logp = lp_all[np.arange(B), a]
    logp_old = lpo_all[np.arange(B), a]
    ratio = np.exp(logp - logp_old)                                  # :79
    lo, hi = 1 - clip_eps, 1 + clip_eps
    clipped = np.clip(ratio, lo, hi)                                  # :85
    adv = np.asarray(advantages, dtype=np.float32)
    s1, s2 = ratio * adv, clipped * adv
    surrogate = -np.mean(np.minimum(s1, s2))                          # :86-91
    ent = -(p * lp_all).sum(-1)
    kl = (np.exp(lpo_all) * (lpo_all - lp_all)).sum(-1)
    total = surrogate - beta * ent.mean()
    # gradient
    use1 = s1 <= s2
    inside = (ratio >= lo) & (ratio <= hi)
    g_logp = np.where(use1 | inside, -adv * ratio, 0.0) / B
    onehot = np.eye(A, dtype=np.float32)[a]
    g = g_logp[:, None] * (onehot - p)
    g = g + (beta / B) * p * (lp_all + ent[:, None])
    return dict(surrogate=np.float32(surrogate), entropy=np.float32(ent.mean()), kl=np.float32(kl.mean()),
                total=np.float32(total), ratio=ratio, clipped=clipped, dlogits=g.astype(np.float32))


def ppo_continuous_loss(mean, log_std, actions, advantages, old_mean, old_std, clip_eps, beta):
    """Continuous PPOHead (ppo_head.py:118-144 + :58-98): MultivariateNormalDiag(mean, exp(log_std) + eps)
    against MVN(old_mean, old_std + eps).  Returns loss terms, fetches and the gradients w.r.t. the
    policy_mean output [B, A] and the log_std variable [A]."""
    F = np.float32
    eps = np.finfo(np.float32).eps
    mu, x = np.asarray(mean, dtype=F), np.asarray(actions, dtype=F)
    mo, so = np.asarray(old_mean, dtype=F), np.asarray(old_std, dtype=F) + eps
    ls = np.asarray(log_std, dtype=F).reshape(-1)
    B, A = mu.shape
    e = np.exp(ls)
    sd = e + eps
    c = F(0.91893853320467274178)
    logp = (-0.5 * ((x - mu) / sd) ** 2 - np.log(sd) - c).sum(1)
    logp_old = (-0.5 * ((x - mo) / so) ** 2 - np.log(so) - c).sum(1)
    ratio = np.exp(logp - logp_old)
    lo, hi = 1 - clip_eps, 1 + clip_eps
    clipped = np.clip(ratio, lo, hi)
    adv = np.asarray(advantages, dtype=F)
    s1, s2 = ratio * adv, clipped * adv
    surrogate = -np.mean(np.minimum(s1, s2))
    ent = (0.5 + c + np.log(sd)).sum()
    kl = (np.log(sd / so) + (so ** 2 + (mo - mu) ** 2) / (2 * sd ** 2) - 0.5).sum(1)
    total = surrogate - beta * ent
    use = (s1 <= s2) | ((ratio >= lo) & (ratio <= hi))
    g_logp = np.where(use, -adv * ratio, 0.0) / B
    d = x - mu
    dmean = g_logp[:, None] * d / sd ** 2
    dls = (g_logp[:, None] * (d ** 2 / sd ** 3 - 1 / sd) * e).sum(0) - beta * e / sd
    return dict(surrogate=F(surrogate), entropy=F(ent), kl=F(kl.mean()), total=F(total), ratio=ratio,
                clipped=clipped, dmean=dmean.astype(F), dlog_std=dls.astype(F))
