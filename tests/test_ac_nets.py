"""Continuous-control updates (DDPG / TD3 / SAC): oracle vs torch autograd on the CPU (the oracle's
hand-written backward passes are checked against an independent implementation), and the HIP
networks vs the oracle on the GPU — a few consecutive learn_from_batch steps from identical
weights and identical batches / noise.  Tolerances: fp32 accumulation order only (rtol 2e-4 on
losses and targets, atol 2e-5 on weights after 4 updates)."""
import numpy as np
import pytest

from oracle import ac_nets as O

F32 = np.float32


def _rand_arrays(rng, spec):
    """{name: (in, out, towers)} -> named_arrays-like dict with xavier-ish weights."""
    out = {}
    for name, (i, o, t) in spec.items():
        out[name + "/kernel"] = [rng.uniform(-1, 1, (i, o)).astype(F32) * F32(np.sqrt(3.0 / i)) for _ in range(t)]
        out[name + "/bias"] = [rng.uniform(-0.1, 0.1, (o,)).astype(F32) for _ in range(t)]
    return out


def _batch(rng, B, D, A):
    s = rng.randn(B, D).astype(F32)
    a = rng.uniform(-1, 1, (B, A)).astype(F32)
    r = rng.randn(B).astype(F32)
    done = (rng.rand(B) < 0.2)
    ns = rng.randn(B, D).astype(F32)
    return s, a, r, done, ns


# ------------------------------------------------------------------ oracle vs torch autograd (CPU)
def test_oracle_sac_policy_backward_matches_torch():
    import torch
    rng = np.random.RandomState(0)
    B, D, A = 6, 5, 3
    arrays = _rand_arrays(rng, {"policy/embedder/dense0": (D, 8, 1), "policy/middleware/dense0": (8, 8, 1),
                                "policy/sac_policy_head/policy_mu_logsig": (8, 2 * A, 1)})
    pol = O.SACPolicyOracle(arrays)
    s = rng.randn(B, D).astype(F32)
    eps = rng.randn(B, A)
    w = rng.randn(B, A).astype(F32)
    o = pol.forward(s, eps)
    pol.backward(logp_weight=1.0, action_weights=w, action_weight_scale=-1.0)
    # torch: same graph through the reparameterised sample
    W = {k: torch.tensor(v[0], requires_grad=True) for k, v in arrays.items()}
    x = torch.tensor(s)
    for n in ("policy/embedder/dense0", "policy/middleware/dense0"):
        x = torch.relu(x @ W[n + "/kernel"] + W[n + "/bias"])
    y = x @ W["policy/sac_policy_head/policy_mu_logsig/kernel"] + W["policy/sac_policy_head/policy_mu_logsig/bias"]
    mu, ls = y[:, :A], torch.clamp(y[:, A:], -20, 2)
    raw = mu + torch.exp(ls) * torch.tensor(eps, dtype=torch.float32)
    act = torch.tanh(raw)
    dist = torch.distributions.Normal(mu, torch.exp(ls))
    logp = dist.log_prob(raw).sum(1) - torch.log(1 - act ** 2 + float(O.EPS32)).sum(1)
    np.testing.assert_allclose(o["logprob"], logp.detach().numpy(), rtol=2e-5, atol=2e-5)
    obj = logp.mean() - (act * torch.tensor(w)).sum()
    obj.backward()
    g = pol.grads()
    for name, d in g.items():
        np.testing.assert_allclose(d[0], W[name].grad.numpy(), rtol=2e-4, atol=2e-6, err_msg=name)


def test_oracle_td3_critic_and_action_gradient_match_torch():
    import torch
    rng = np.random.RandomState(1)
    B, D, A = 7, 4, 2
    arrays = _rand_arrays(rng, {"critic/middleware/dense0": (D + A, 8, 2), "critic/middleware/dense1": (8, 6, 2),
                                "critic/v_head/output": (6, 1, 2)})
    c = O.CriticOracle(arrays, streams=2)
    s, a, r, done, ns = _batch(rng, B, D, A)
    y = rng.randn(B).astype(F32)
    q = c.forward(s, a)
    losses = c.train_backward(y)
    W = {k: [torch.tensor(x, requires_grad=True) for x in v] for k, v in arrays.items()}
    at = torch.tensor(a, requires_grad=True)

    def tq(t):
        h = torch.cat([at, torch.tensor(s)], 1)
        for n in ("critic/middleware/dense0", "critic/middleware/dense1"):
            h = torch.relu(h @ W[n + "/kernel"][t] + W[n + "/bias"][t])
        return (h @ W["critic/v_head/output/kernel"][t] + W["critic/v_head/output/bias"][t])[:, 0]
    loss = sum(((torch.tensor(y) - tq(t)) ** 2).mean() for t in range(2))
    loss.backward()
    np.testing.assert_allclose(float(sum(losses)), float(loss), rtol=1e-5)
    for name, d in c.grads().items():
        for t in range(2):
            np.testing.assert_allclose(d[t], W[name][t].grad.numpy(), rtol=2e-4, atol=2e-6, err_msg=name)
    c.forward(s, a)
    g = c.action_gradient(A)
    at.grad = None
    tq(0).mean().backward()
    np.testing.assert_allclose(g, at.grad.numpy(), rtol=2e-4, atol=2e-7)


def test_oracle_sac_q_action_gradient_matches_torch():
    import torch
    rng = np.random.RandomState(2)
    B, D, A = 9, 5, 3
    arrays = _rand_arrays(rng, {"q/q_head/obs_fc": (D, 8, 2), "q/q_head/act_fc": (A, 8, 2),
                                "q/q_head/fc1": (8, 8, 2), "q/q_head/q_output": (8, 1, 2)})
    qn = O.SACQOracle(arrays)
    s, a, r, done, ns = _batch(rng, B, D, A)
    q = qn.forward(s, a)
    g = qn.action_gradient(q)
    W = {k: [torch.tensor(x) for x in v] for k, v in arrays.items()}
    at = torch.tensor(a, requires_grad=True)
    qs = []
    for t in range(2):
        h = torch.relu(torch.tensor(s) @ W["q/q_head/obs_fc/kernel"][t] + W["q/q_head/obs_fc/bias"][t]) + \
            torch.relu(at @ W["q/q_head/act_fc/kernel"][t] + W["q/q_head/act_fc/bias"][t])
        h = torch.relu(h @ W["q/q_head/fc1/kernel"][t] + W["q/q_head/fc1/bias"][t])
        qs.append((h @ W["q/q_head/q_output/kernel"][t] + W["q/q_head/q_output/bias"][t])[:, 0])
    torch.minimum(qs[0], qs[1]).mean().backward()
    np.testing.assert_allclose(g, at.grad.numpy(), rtol=2e-4, atol=2e-7)


def _bn_arrays(rng, names_channels):
    out = {}
    for n, c in names_channels:
        out[n + "/gamma"] = [rng.uniform(0.5, 1.5, c).astype(F32)]
        out[n + "/beta"] = [rng.uniform(-0.3, 0.3, c).astype(F32)]
        out[n + "/moving_mean"] = [rng.uniform(-0.2, 0.2, c).astype(F32)]
        out[n + "/moving_variance"] = [rng.uniform(0.5, 1.5, c).astype(F32)]
    return out


def _ddpg_bn_arrays(rng, D, A, H1=10, H2=7):
    a = _rand_arrays(rng, {"actor/embedder/dense0": (D, H1, 1), "actor/middleware/dense0": (H1, H2, 1),
                           "actor/ddpg_actor_head/fc_mean": (H2, A, 1)})
    a.update(_bn_arrays(rng, [("actor/embedder/batchnorm0", H1), ("actor/middleware/batchnorm0", H2),
                              ("actor/ddpg_actor_head/batchnorm0", A)]))
    c = _rand_arrays(rng, {"critic/embedder/dense0": (D, H1, 1), "critic/middleware/dense0": (H1 + A, H2, 1),
                           "critic/v_head/output": (H2, 1, 1)})
    c.update(_bn_arrays(rng, [("critic/embedder/batchnorm0", H1), ("critic/middleware/batchnorm0", H2)]))
    return a, c


def test_oracle_batchnorm_networks_match_torch():
    """The use_batchnorm=True DDPG networks of the oracle (Dense -> batch norm -> activation: embedder, middleware, and
    fc_mean -> batch norm -> tanh in the actor; agents/ddpg_agent.py:37-60) against torch: training-mode forward and
    every gradient through the batch statistics (torch.nn.functional.batch_norm, eps 1e-3), the inference-mode forward
    on the moving statistics, and the moving-average step (population variance, m -= (m - batch) * 0.01)."""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.RandomState(5)
    D, A, B = 6, 3, 16
    aa, ca = _ddpg_bn_arrays(rng, D, A)
    actor, critic = O.ActorOracle(aa, 1.0), O.CriticOracle(ca, streams=1)
    s, a, r, done, ns = _batch(rng, B, D, A)
    tw = {k: torch.tensor(v[0], requires_grad=True) for k, v in list(aa.items()) + list(ca.items())}

    def bn(x, name, training):
        return Fn.batch_norm(x, tw[name + "/moving_mean"].detach().clone(), tw[name + "/moving_variance"].detach().clone(),
                             tw[name + "/gamma"], tw[name + "/beta"], training=training, momentum=0.0, eps=1e-3)

    def t_actor(x, training):
        h = torch.relu(bn(x @ tw["actor/embedder/dense0/kernel"] + tw["actor/embedder/dense0/bias"],
                          "actor/embedder/batchnorm0", training))
        h = torch.relu(bn(h @ tw["actor/middleware/dense0/kernel"] + tw["actor/middleware/dense0/bias"],
                          "actor/middleware/batchnorm0", training))
        return torch.tanh(bn(h @ tw["actor/ddpg_actor_head/fc_mean/kernel"] + tw["actor/ddpg_actor_head/fc_mean/bias"],
                             "actor/ddpg_actor_head/batchnorm0", training))

    def t_critic(x, act, training):
        h = torch.relu(bn(x @ tw["critic/embedder/dense0/kernel"] + tw["critic/embedder/dense0/bias"],
                          "critic/embedder/batchnorm0", training))
        h = torch.cat([act, h], dim=1)
        h = torch.relu(bn(h @ tw["critic/middleware/dense0/kernel"] + tw["critic/middleware/dense0/bias"],
                          "critic/middleware/batchnorm0", training))
        return (h @ tw["critic/v_head/output/kernel"] + tw["critic/v_head/output/bias"])[:, 0]
    # inference mode: the moving statistics
    np.testing.assert_allclose(actor.forward(s), t_actor(torch.tensor(s), False).detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(critic.forward(s, a)[0], t_critic(torch.tensor(s), torch.tensor(a), False).detach().numpy(),
                               rtol=2e-5, atol=2e-6)
    # training mode: forward, the critic loss gradients, the action gradient, the actor gradients
    actor.set_is_training(True); critic.set_is_training(True)
    y = rng.randn(B).astype(F32)
    q = critic.forward(s, a)[0]
    tq = t_critic(torch.tensor(s), torch.tensor(a), True)
    np.testing.assert_allclose(q, tq.detach().numpy(), rtol=2e-5, atol=2e-6)
    critic.train_backward(y)
    ((torch.tensor(y) - tq) ** 2).mean().backward()
    for name, towers in critic.grads().items():
        np.testing.assert_allclose(towers[0], tw[name].grad.numpy(), rtol=3e-4, atol=3e-6, err_msg=name)
    pending = {n: l.pending for n, _, l in critic.layers if hasattr(l, "pending")}
    before = {n: (l.moving_mean.copy(), l.moving_var.copy()) for n, _, l in critic.layers if hasattr(l, "pending")}
    critic.apply()
    for n, _, l in critic.layers:
        if hasattr(l, "pending"):
            mean, var = pending[n]
            np.testing.assert_allclose(l.moving_mean, before[n][0] * 0.99 + mean * 0.01, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(l.moving_var, before[n][1] * 0.99 + var * 0.01, rtol=1e-6, atol=1e-7)
    aa2, ca2 = _ddpg_bn_arrays(np.random.RandomState(5), D, A)       # fresh copies (apply() moved the critic)
    critic = O.CriticOracle(ca2, streams=1)
    critic.set_is_training(True)
    mu = actor.forward(s)
    critic.forward(s, mu)
    g = critic.action_gradient(A)
    at = torch.tensor(mu, requires_grad=True)
    t_critic(torch.tensor(s), at, True).mean().backward()
    np.testing.assert_allclose(g, at.grad.numpy(), rtol=3e-4, atol=3e-7)
    for v in tw.values():
        v.grad = None
    w = rng.randn(B, A).astype(F32)
    actor.forward(s)
    actor.backward(w)
    (t_actor(torch.tensor(s), True) * torch.tensor(w)).sum().backward()
    for name, towers in actor.grads().items():
        np.testing.assert_allclose(towers[0], tw[name].grad.numpy(), rtol=3e-4, atol=3e-6, err_msg=name)


# ----------------------------------------------------------------------------- HIP vs oracle (GPU)
def _t(x, dev, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), device=dev) if dtype is None else \
        torch.as_tensor(np.ascontiguousarray(x).astype(dtype), device=dev)


def _check_weights(net, onet, atol=2e-5):
    w = net.params.named_arrays()
    for name, towers in onet.weights().items():
        for t, arr in towers.items():
            np.testing.assert_allclose(w[name][t], arr, rtol=0, atol=atol, err_msg="%s[%d]" % (name, t))


class _B:
    """minimal DeviceBatch stand-in"""
    def __init__(self, dev, batch, paired=False):
        import torch
        s, a, r, done, ns = batch
        self._info = {}
        if paired:
            both = _t(np.stack([s, ns]), dev)
            self._info["states_pair"] = both
            self._states, self._next_states = {"observation": both[0]}, {"observation": both[1]}
        else:
            self._states = {"observation": _t(s, dev)}
            self._next_states = {"observation": _t(ns, dev)}
        self._a, self._r, self._d = _t(a, dev), _t(r, dev), _t(done.astype(np.uint8), dev)

    def actions(self): return self._a
    def rewards(self): return self._r
    def game_overs(self): return self._d


def _agent(dev, cls, params, D, A, B, n_env=4):
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    ep = SyntheticVectorEnvironmentParameters("vector", n_env, (D,), None, action_dim=A, episode_length=8, seed=5)
    env = SyntheticVectorEnvironment(ep, dev)
    for n in params.network_wrappers.values():
        n.batch_size = B
    params.memory.max_size = (MemoryGranularity.Transitions, 256)
    return cls(params, env, dev, use_graphs=False)


@pytest.mark.gpu
def test_ddpg_update_matches_oracle(dev):
    from coach_amd.agents.ddpg_agent import DDPGAgent, DDPGAgentParameters
    D, A, B = 11, 3, 32
    ag = _agent(dev, DDPGAgent, DDPGAgentParameters(), D, A, B)
    actor, critic = ag.networks["actor"], ag.networks["critic"]
    oa = O.ActorOracle(actor.params.named_arrays(), 1.0, lr=1e-4)
    oc = O.CriticOracle(critic.params.named_arrays(), streams=1, lr=1e-3)
    rng = np.random.RandomState(3)
    for it in range(4):
        batch = _batch(rng, B, D, A)
        r = O.ddpg_update(oa, oc, batch)
        ag._learn_device(_B(dev, batch))
        np.testing.assert_allclose(ag.td_targets.cpu().numpy(), r["targets"], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(-ag.neg_action_grad.cpu().numpy(), r["action_grad"], rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(float(critic.loss[0]), r["loss"], rtol=2e-4)
        np.testing.assert_allclose(float(critic.norm), r["norm"], rtol=2e-4)
        ag.update_target_networks(0.001); oa.mix_target(0.001); oc.mix_target(0.001)
    _check_weights(actor, oa)
    _check_weights(critic, oc)
    ag.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("paired,fused", [(False, False), (True, False), (True, True)])
def test_td3_update_matches_oracle(dev, paired, fused, monkeypatch):
    """fused: the five-launch update of csrc/ac_fused.hip (the default wherever the topology allows it); otherwise the
    layer-by-layer launch chain."""
    import torch
    from coach_amd.agents.td3_agent import TD3Agent, TD3AgentParameters
    monkeypatch.setattr(TD3Agent, "FUSED_UPDATE", fused)
    D, A, B = 17, 6, 100
    ag = _agent(dev, TD3Agent, TD3AgentParameters(), D, A, B)
    assert (ag._fused() is not None) == fused
    actor, critic = ag.networks["actor"], ag.networks["critic"]
    oa = O.ActorOracle(actor.params.named_arrays(), 1.0, lr=1e-3)
    oc = O.CriticOracle(critic.params.named_arrays(), streams=2, lr=1e-3)
    rng = np.random.RandomState(4)
    for it in range(1, 5):
        batch = _batch(rng, B, D, A)
        noise = rng.normal(0, 0.2, (B, A))
        r = O.td3_update(oa, oc, batch, noise, it, ag.low, ag.high)
        db = _B(dev, batch, paired)
        ag.noise.copy_(_t(noise, dev))
        ag._critic_device(db)
        if it % 2 == 0:
            ag._actor_device(db)
            ag.update_target_networks(0.005); oa.mix_target(0.005); oc.mix_target(0.005)
        np.testing.assert_allclose(ag.td_targets.cpu().numpy(), r["targets"], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(float(critic.loss[:2].sum()), r["loss"], rtol=2e-4)
    _check_weights(actor, oa)
    _check_weights(critic, oc)
    ag.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("resample,paired,dims", [(True, False, (23, 5, 64)), (False, False, (23, 5, 64)),
                                                  (True, True, (23, 5, 64)),
                                                  (True, False, (11, 3, 37)),         # ragged last row block
                                                  (True, True, (376, 17, 256))])      # BASELINE C5: Humanoid SAC
def test_sac_update_matches_oracle(dev, resample, paired, dims, fused, monkeypatch):
    """fused: the six-launch update of csrc/ac_fused.hip (the default wherever the topology allows it); otherwise the
    layer-by-layer launch chain."""
    from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent, SoftActorCriticAgentParameters
    monkeypatch.setattr(SoftActorCriticAgent, "FUSED_UPDATE", fused)
    D, A, B = dims
    p = SoftActorCriticAgentParameters()
    p.algorithm.resample_noise_per_pass = resample
    ag = _agent(dev, SoftActorCriticAgent, p, D, A, B)
    assert (ag._fused() is not None) == fused
    pol, qn, vn = ag.networks["policy"], ag.networks["q"], ag.networks["v"]
    op = O.SACPolicyOracle(pol.params.named_arrays())
    oq = O.SACQOracle(qn.params.named_arrays())
    ov = O.SACValueOracle(vn.params.named_arrays())
    rng = np.random.RandomState(5)
    for it in range(4):
        batch = _batch(rng, B, D, A)
        if D > 100:
            # 376 unit-variance inputs through a freshly initialised policy give log-std outputs of +-2 and
            # pre-tanh actions of +-20: log(1 - tanh^2 + 1e-6) is then decided by the last bit of tanh.  Normalised
            # observations (what the reference's Humanoid preset feeds, ObservationNormalizationFilter) keep it sane.
            batch = (batch[0] * np.float32(0.05),) + tuple(batch[1:4]) + (batch[4] * np.float32(0.05),)
        z = rng.standard_normal((3, B, A))
        r = O.sac_update(op, oq, ov, batch, z, resample=resample)
        ag.normals.copy_(_t(z, dev))
        ag._learn_device(_B(dev, batch, paired))
        np.testing.assert_allclose(ag.dq_da.cpu().numpy(), r["dq_da"], rtol=2e-3, atol=1e-7)
        # the log-probability's conditioning: d log(1 - t^2 + eps) = 2 |t| dt / (1 - t^2 + eps), dt = 2 ulp of tanh
        t = r["actions"].astype(np.float64)
        cond = (2 * np.abs(t) * 2.4e-7 / (1 - t * t + 1e-6)).sum(axis=1)
        vt = ag.value_targets.cpu().numpy()
        assert np.all(np.abs(vt - r["value_targets"]) <= 2e-4 * np.abs(r["value_targets"]) + 2e-5 + cond), \
            np.abs(vt - r["value_targets"]).max()
        np.testing.assert_allclose(ag.td_targets.cpu().numpy(), r["td_targets"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(float(qn.loss[:2].sum()), r["loss"], rtol=3e-4)
        assert float(qn.loss[2]) == float(qn.loss[0] + qn.loss[1])
        np.testing.assert_allclose(float(vn.loss), r["v_loss"], rtol=3e-4)
        ag.update_target_networks(0.005); ov.mix_target(0.005)
    _check_weights(pol, op, atol=5e-5)
    _check_weights(qn, oq, atol=5e-5)
    _check_weights(vn, ov, atol=5e-5)
    ag.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("B,C", [(16, 7), (64, 400), (100, 300)])
def test_batchnorm_kernels_match_oracle(dev, B, C):
    """rlx_bn_forward (batch and moving statistics), rlx_bn_backward (with and without the activation derivative and the
    weight gradients) and rlx_bn_update_moving against oracle.nn.BatchNorm."""
    import torch
    from coach_amd import _rlx
    from oracle import nn as N
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    rng = np.random.RandomState(B + C)
    x = (rng.randn(B, C) * 2 + 0.5).astype(F32)
    dy = rng.randn(B, C).astype(F32)
    arr = {k: v[0] for k, v in _bn_arrays(rng, [("bn", C)]).items()}
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    for act, code in (("relu", 1), ("tanh", 2), (None, 0)):
        o = N.BatchNorm(arr["bn/gamma"], arr["bn/beta"], arr["bn/moving_mean"], arr["bn/moving_variance"], act)
        d = {k: _t(v, dev) for k, v in arr.items()}
        y, mean, var, dx, dg, db = z(B, C), z(C), z(C), z(B, C), z(C), z(C)
        ref = o.forward(x)                                                     # inference
        lib.bn_forward(_t(x, dev), d["bn/gamma"], d["bn/beta"], d["bn/moving_mean"], d["bn/moving_variance"], B, C, 1e-3,
                       0, code, y, None, None, s_)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
        o.training = True
        ref = o.forward(x)                                                     # training
        lib.bn_forward(_t(x, dev), d["bn/gamma"], d["bn/beta"], None, None, B, C, 1e-3, 1, code, y, mean, var, s_)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(mean.cpu().numpy(), o.mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var.cpu().numpy(), o.var, rtol=1e-5, atol=1e-6)
        rdx = o.backward(dy)
        lib.bn_backward(_t(dy, dev), y, _t(x, dev), d["bn/gamma"], mean, var, B, C, 1e-3, code, dx, dg, db, s_)
        np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=3e-4, atol=3e-6)
        np.testing.assert_allclose(dg.cpu().numpy(), o.dgamma, rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(db.cpu().numpy(), o.dbeta, rtol=3e-4, atol=3e-5)
        dx.zero_()                                                             # input gradient only
        lib.bn_backward(_t(dy, dev), y, _t(x, dev), d["bn/gamma"], mean, var, B, C, 1e-3, code, dx, None, None, s_)
        np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=3e-4, atol=3e-6)
        o.commit()
        lib.bn_update_moving(d["bn/moving_mean"], d["bn/moving_variance"], mean, var, C, 0.99, s_)
        np.testing.assert_allclose(d["bn/moving_mean"].cpu().numpy(), o.moving_mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(d["bn/moving_variance"].cpu().numpy(), o.moving_var, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_ddpg_batchnorm_update_matches_oracle(dev):
    """DDPGAgentParameters(use_batchnorm=True) (agents/ddpg_agent.py:111-122): four learn_from_batch steps of the device
    agent against oracle.ac_nets.ddpg_update on batch-normalised networks — batch statistics in every pass of the update
    (target networks included), moving averages stepped at apply time from the critic's (s, a) pass and the actor's s pass
    — then acting on the moving statistics."""
    from coach_amd.agents.ddpg_agent import DDPGAgent, DDPGAgentParameters
    D, A, B = 11, 3, 32
    p = DDPGAgentParameters(use_batchnorm=True)
    p.network_wrappers["actor"].observation_embedder_scheme, p.network_wrappers["actor"].middleware_scheme = (24,), (16,)
    p.network_wrappers["critic"].observation_embedder_scheme, p.network_wrappers["critic"].middleware_scheme = (24,), (16,)
    ag = _agent(dev, DDPGAgent, p, D, A, B)
    actor, critic = ag.networks["actor"], ag.networks["critic"]
    assert len(actor.bn_layers) == 3 and len(critic.bn_layers) == 2
    oa = O.ActorOracle(actor.params.named_arrays(), 1.0, lr=1e-4)
    oc = O.CriticOracle(critic.params.named_arrays(), streams=1, lr=1e-3)
    rng = np.random.RandomState(3)
    for it in range(4):
        batch = _batch(rng, B, D, A)
        r = O.ddpg_update(oa, oc, batch)
        ag.learn_from_batch(_B(dev, batch))
        np.testing.assert_allclose(ag.td_targets.cpu().numpy(), r["targets"], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(-ag.neg_action_grad.cpu().numpy(), r["action_grad"], rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(float(critic.loss[0]), r["loss"], rtol=2e-4)
        np.testing.assert_allclose(float(critic.norm), r["norm"], rtol=5e-4)
        ag.update_target_networks(0.001); oa.mix_target(0.001); oc.mix_target(0.001)
    # kernels, gamma / beta and the moving statistics.  The bias of a dense layer that feeds a batch norm has a gradient
    # that is zero up to rounding (the batch mean is subtracted right after it), and Adam turns rounding noise of either
    # sign into steps of up to lr: such a bias is checked to 4 updates x lr, nothing reads it through the normalisation
    for net, onet, lr in ((actor, oa, 1e-4), (critic, oc, 1e-3)):
        w = net.params.named_arrays()
        for name, towers in onet.weights().items():
            dead = name.endswith("/bias") and name.replace("dense", "batchnorm").replace("/bias", "/gamma") in w or \
                name == "actor/ddpg_actor_head/fc_mean/bias"
            np.testing.assert_allclose(w[name][0], towers[0], rtol=0, atol=4.5 * lr if dead else 3e-5, err_msg=name)
    s = rng.randn(4, D).astype(F32)                       # acting: is_training is off again
    mu, _ = actor.forward(_t(s, dev), 4, tag="act")
    np.testing.assert_allclose(mu.cpu().numpy().reshape(4, A), oa.forward(s), rtol=2e-4, atol=2e-5)
    ag.check_status()
