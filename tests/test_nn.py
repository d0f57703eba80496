"""Networks: numpy oracle vs an independent torch-CPU implementation (CPU), HIP nets vs the oracle
(GPU): forward, gradients, TF1-Adam step, target mixing.  fp32 tolerances are stated inline."""
import numpy as np
import pytest

from oracle import losses as L
from oracle import nn as N
from oracle.optim import AdamTF1, mix_weights


# ------------------------------------------------------------------------------------ CPU
def test_oracle_conv_dense_match_torch_autograd():
    import torch
    import torch.nn.functional as Fn
    rng = np.random.RandomState(0)
    B = 3
    x = rng.randint(0, 256, size=(B, 20, 20, 4)).astype(np.uint8)
    Wc = (rng.randn(4 * 4 * 4, 8) * 0.2).astype(np.float32)
    bc = rng.randn(8).astype(np.float32) * 0.1
    Wd = (rng.randn(5 * 5 * 8, 6) * 0.1).astype(np.float32)
    bd = rng.randn(6).astype(np.float32) * 0.1
    chain = N.Chain([N.Conv(Wc, bc, (20, 20, 4), 4, 4, "tanh"), N.Dense(Wd, bd, "relu")])
    y = chain.forward(N.prep_obs(x, True))
    dy = rng.randn(*y.shape).astype(np.float32)
    chain.backward(dy)
    # torch: NCHW conv with the same [KH,KW,Cin,Cout] kernel
    xt = torch.tensor(x.astype(np.float32) / 255.0).permute(0, 3, 1, 2)
    wc = torch.tensor(Wc.reshape(4, 4, 4, 8)).permute(3, 2, 0, 1).clone().requires_grad_(True)
    bct = torch.tensor(bc, requires_grad=True)
    wd = torch.tensor(Wd, requires_grad=True)
    bdt = torch.tensor(bd, requires_grad=True)
    h = torch.tanh(Fn.conv2d(xt, wc, bct, stride=4)).permute(0, 2, 3, 1).reshape(B, -1)   # NHWC flatten
    yt = torch.relu(h @ wd + bdt)
    yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(chain.layers[1].dW, wd.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(chain.layers[0].dW.reshape(4, 4, 4, 8),
                               wc.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(chain.layers[0].db, bct.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_oracle_dueling_head_matches_torch_autograd():
    """DuelingQHead restatement (dueling_q_head.py:33-48) + rescale_gradient_from_head_by_factor +
    tf.clip_by_global_norm against torch autograd on the same weights."""
    import torch
    rng = np.random.RandomState(5)
    B, F_, A = 6, 10, 4
    names = {"main/dueling_q_values_head/fc1": ((F_, 512), 2),
             "main/dueling_q_values_head/state_value/fc2": ((512, 1), 1),
             "main/dueling_q_values_head/action_advantage/fc2": ((512, A), 1)}
    arrays = {}
    for n, (shape, towers) in names.items():
        arrays[n + "/kernel"] = [(rng.randn(*shape) / np.sqrt(shape[0])).astype(np.float32) for _ in range(towers)]
        arrays[n + "/bias"] = [(rng.randn(shape[1]) * 0.1).astype(np.float32) for _ in range(towers)]
    head = N.DuelingHead(arrays, "relu")
    x = rng.randn(B, F_).astype(np.float32)
    q = head.forward(x)
    dq = rng.randn(B, A).astype(np.float32)
    f = 1 / np.sqrt(2)
    dx = head.backward(dq) * np.float32(f)
    t = {k: [torch.tensor(a, requires_grad=True) for a in v] for k, v in arrays.items()}
    xt = torch.tensor(x, requires_grad=True)
    xin = (1 - f) * xt.detach() + f * xt                            # general_network.py:296-303
    hn = "main/dueling_q_values_head"
    hv = torch.relu(xin @ t[hn + "/fc1/kernel"][0] + t[hn + "/fc1/bias"][0])
    ha = torch.relu(xin @ t[hn + "/fc1/kernel"][1] + t[hn + "/fc1/bias"][1])
    v = hv @ t[hn + "/state_value/fc2/kernel"][0] + t[hn + "/state_value/fc2/bias"][0]
    a = ha @ t[hn + "/action_advantage/fc2/kernel"][0] + t[hn + "/action_advantage/fc2/bias"][0]
    qt = v + (a - a.mean(1, keepdim=True))
    qt.backward(torch.tensor(dq))
    np.testing.assert_allclose(q, qt.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(head.a1.dW, t[hn + "/fc1/kernel"][1].grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(head.v2.dW, t[hn + "/state_value/fc2/kernel"][0].grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(head.a2.db, t[hn + "/action_advantage/fc2/bias"][0].grad.numpy(), rtol=1e-4, atol=1e-6)
    # clip_by_global_norm as the oracle agent applies it
    params = [p for v in t.values() for p in v]
    norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)))
    torch.nn.utils.clip_grad_norm_(params, 0.5 * norm)               # forces a scale of 0.5
    c = np.float32(0.5 * norm)
    scale = c * min(np.float32(1) / np.float32(norm), np.float32(1) / c)
    np.testing.assert_allclose(head.a1.dW * scale, t[hn + "/fc1/kernel"][1].grad.numpy(), rtol=1e-4, atol=1e-6)


def test_oracle_ppo_loss_known_answers():
    """Reference KATs: rl_coach/tests/architectures/mxnet_components/heads/test_ppo_head.py
    :141-156 (log-prob, entropy), :160-170 (KL), :363-376 (clipped surrogate = -0.142857153)."""
    probs = np.array([[0.8, 0.2], [0.7, 0.3], [0.5, 0.5]], dtype=np.float32)
    np.testing.assert_allclose(L.categorical_log_prob(probs, [0, 1, 0]),
                               [-0.22314353, -1.20397282, -0.69314718], rtol=1e-6)
    ent = [-(p * np.log(p)).sum() for p in probs.astype(np.float64)]
    np.testing.assert_allclose(L.categorical_entropy(probs), ent, rtol=1e-6)
    a, b, c = (np.array([x], dtype=np.float32) for x in ([.4, .2, .4], [.3, .4, .3], [.2, .6, .2]))
    np.testing.assert_allclose(L.categorical_kl(a, b), [0.09151624], rtol=1e-5)
    np.testing.assert_allclose(L.categorical_kl(a, c), [0.33479536], rtol=1e-5)
    np.testing.assert_allclose(L.categorical_kl(c, a), [0.38190854], rtol=1e-5)
    new = np.array([[0.9, 0.1], [0.2, 0.8], [0.4, 0.6]], dtype=np.float32)
    old = np.array([[0.7, 0.3], [0.2, 0.8], [0.4, 0.6]], dtype=np.float32)
    r = L.ppo_discrete_loss(np.log(new), [0, 1, 0], [-2, 2, 1], old, 0.2, 0.0)
    np.testing.assert_allclose(r["surrogate"], -0.142857153, rtol=2e-6)


def test_oracle_ppo_loss_gradient_matches_torch_autograd():
    import torch
    rng = np.random.RandomState(1)
    B, A = 16, 6
    logits = rng.randn(B, A).astype(np.float32)
    old = N.softmax(logits + 0.3 * rng.randn(B, A).astype(np.float32))
    actions = rng.randint(0, A, size=B)
    adv = rng.randn(B).astype(np.float32)
    r = L.ppo_discrete_loss(logits, actions, adv, old, 0.2, 0.01)
    z = torch.tensor(logits, requires_grad=True)
    lp = torch.log_softmax(z, -1)
    lpo = torch.log_softmax(torch.log(torch.tensor(old)), -1)
    idx = torch.tensor(actions)[:, None]
    ratio = torch.exp(lp.gather(1, idx)[:, 0] - lpo.gather(1, idx)[:, 0])
    advt = torch.tensor(adv)
    sur = -torch.min(ratio * advt, torch.clamp(ratio, 0.8, 1.2) * advt).mean()
    ent = -(lp.exp() * lp).sum(-1).mean()
    (sur - 0.01 * ent).backward()
    np.testing.assert_allclose(r["total"], float(sur - 0.01 * ent), rtol=1e-5)
    np.testing.assert_allclose(r["dlogits"], z.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_oracle_adam_tf1_closed_form():
    """First step of ApplyAdam: m = (1-b1) g, v = (1-b2) g^2, alpha = lr*sqrt(1-b2)/(1-b1)."""
    g = np.array([0.5, -2.0, 1e-3], dtype=np.float32)
    w = np.ones(3, dtype=np.float32)
    opt = AdamTF1(3, 1e-3, 0.9, 0.99, 1e-4)
    opt.step(w, g)
    alpha = 1e-3 * np.sqrt(1 - 0.99) / (1 - 0.9)
    ref = 1 - alpha * (0.1 * g) / (np.sqrt(0.01 * g * g) + 1e-4)
    np.testing.assert_allclose(w, ref, rtol=1e-6)
    np.testing.assert_allclose(mix_weights(np.float32([1, 2]), np.float32([3, 6]), 0.25), [1.5, 3.0])


# ------------------------------------------------------------------------------------ GPU
def _cmp_named(hip_arrays, oracle_dict, rtol, atol):
    for name, per_tower in oracle_dict.items():
        for t, ref in per_tower.items():
            got = hip_arrays[name][t]
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg="%s tower %d" % (name, t))


def _grads_named(net):
    p = net.params
    return {name: [p.g(name, t).cpu().numpy().copy() for t in range(e[2])] for name, e in p.entries.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("obs_shape,B", [((84, 84, 4), 8), ((17,), 64)])
def test_hip_clipped_ppo_net_vs_oracle(rlx, dev, obs_shape, B):
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    from oracle.agents import ClippedPPOOracle
    from tests.util import dev_tensor
    np.random.seed(5)
    rng = np.random.RandomState(7)
    A = 6
    net = ClippedPPONet(dev, obs_shape, A, seed=3)
    o = ClippedPPOOracle(net.params.named_arrays(), obs_shape, A)
    image = len(obs_shape) == 3
    obs = rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8) if image else \
        rng.randn(B, *obs_shape).astype(np.float32)
    obs_d = dev_tensor(obs, dev)
    # inference paths
    probs = net.policy_probs(obs_d, B).cpu().numpy()
    np.testing.assert_allclose(probs, o.policy_probs(obs), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(net.values(obs_d, B).cpu().numpy(), o.values(obs), rtol=2e-4, atol=2e-5)
    # three minibatch updates from identical weights and inputs
    frozen = o.clone_policy()
    net.update_target(1.0)
    for step in range(3):
        actions = rng.randint(0, A, size=B)
        adv = rng.randn(B).astype(np.float32)
        vt = rng.randn(B).astype(np.float32)
        old = net.policy_probs(obs_d, B, use_target=True, tag="old")
        np.testing.assert_allclose(old.cpu().numpy(), o.policy_probs(obs, frozen), rtol=2e-4, atol=1e-6)
        ratio = torch.empty(B, dtype=torch.float32, device=dev)
        clipped = torch.empty(B, dtype=torch.float32, device=dev)
        sc = net.train_minibatch(obs_d, B, dev_tensor(actions, dev, np.int32), dev_tensor(adv, dev),
                                 dev_tensor(vt, dev), old, ratio_out=ratio, clipped_out=clipped).cpu().numpy()
        ref = o.train_minibatch(obs, actions, adv, vt, old.cpu().numpy())
        net.check_status()
        np.testing.assert_allclose(sc[:5], [ref["surrogate"], ref["entropy"], ref["kl"], ref["total"],
                                            ref["value_loss"]], rtol=5e-4, atol=2e-6)
        np.testing.assert_allclose(ratio.cpu().numpy(), ref["ratio"], rtol=5e-4)
        np.testing.assert_allclose(net.norm.item(), ref["norm"], rtol=1e-3)
        _cmp_named(_grads_named(net), o.grads(), rtol=2e-3, atol=2e-5 * max(1.0, ref["norm"]))
        _cmp_named(net.params.named_arrays(), o.weights(), rtol=1e-3, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("obs_shape,B,huber,ddqn", [((4,), 32, False, False), ((84, 84, 4), 8, True, True)])
def test_hip_dqn_net_vs_oracle(rlx, dev, obs_shape, B, huber, ddqn):
    import torch
    from coach_amd.nn.networks import DQNNet
    from oracle.agents import DQNOracle
    from tests.util import dev_tensor
    rng = np.random.RandomState(11)
    A = 4
    net = DQNNet(dev, obs_shape, A, replace_mse_with_huber_loss=huber, seed=1)
    net._fused = None        # this test reads the materialised gradients: the layer-by-layer path (the one-launch
    #                          update never writes them; tests/test_mlp_fused.py compares the two paths)
    o = DQNOracle(net.params.named_arrays(), obs_shape, A, huber=huber)
    image = len(obs_shape) == 3
    gen = (lambda: rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8)) if image else \
        (lambda: rng.randn(B, *obs_shape).astype(np.float32))
    for step in range(3):
        obs, nxt = gen(), gen()
        actions = rng.randint(0, A, size=B)
        rewards = rng.choice([-1.0, 0.0, 1.0], size=B).astype(np.float32)
        go = rng.rand(B) < 0.2
        w = rng.rand(B).astype(np.float32) if step == 1 else None
        err = torch.empty(B, dtype=torch.float64, device=dev)
        loss = net.learn_from_batch(dev_tensor(obs, dev), dev_tensor(nxt, dev), B,
                                    dev_tensor(actions, dev, np.int32), dev_tensor(rewards, dev),
                                    dev_tensor(go, dev, np.uint8), 0.99,
                                    importance_weights=None if w is None else dev_tensor(w, dev),
                                    td_errors=err, double_dqn=ddqn)
        ref = o.learn_from_batch(obs, nxt, actions, rewards, go, 0.99, w, ddqn)
        net.check_status()
        np.testing.assert_allclose(loss.item(), ref["loss"], rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(err.cpu().numpy(), ref["td_errors"], rtol=1e-3, atol=1e-5)
        _cmp_named(_grads_named(net), o.grads(), rtol=2e-3, atol=2e-5 * max(1.0, ref["norm"]))
        _cmp_named(net.params.named_arrays(), o.weights(), rtol=1e-3, atol=2e-5)
        if step == 1:
            net.update_target(0.25)
            o.update_target(0.25)
    tgt = net.params.named_arrays(net.target)
    np.testing.assert_allclose(tgt["main/q_head/dense/kernel"][0], o.target[1].W, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("obs_shape,B,middleware,clip", [((6,), 32, "Medium", None), ((44, 44, 4), 8, "Empty", 0.05)])
def test_hip_dueling_dqn_net_vs_oracle(rlx, dev, obs_shape, B, middleware, clip):
    """Dueling DDQN as presets/Atari_Dueling_DDQN.py configures it: DuelingQHead on an Empty middleware,
    head gradient rescaled by 1/sqrt(2), gradients clipped by global norm — HIP network vs the oracle."""
    import torch
    from coach_amd.nn.networks import DQNNet
    from oracle.agents import DQNOracle
    from tests.util import dev_tensor
    rng = np.random.RandomState(13)
    A, f = 5, 1 / np.sqrt(2)
    net = DQNNet(dev, obs_shape, A, middleware=middleware, seed=2, dueling=True, head_gradient_rescale=f,
                 clip_gradients=clip)
    o = DQNOracle(net.params.named_arrays(), obs_shape, A, dueling=True, head_gradient_rescale=f,
                  clip_gradients=clip)
    image = len(obs_shape) == 3
    gen = (lambda: rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8)) if image else \
        (lambda: rng.randn(B, *obs_shape).astype(np.float32))
    probe = gen()
    np.testing.assert_allclose(net.q_values(dev_tensor(probe, dev), B).data.view(B, A).cpu().numpy(),
                               o.q(probe), rtol=1e-4, atol=1e-5)
    for step in range(3):
        obs, nxt = gen(), gen()
        actions = rng.randint(0, A, size=B)
        rewards = rng.choice([-1.0, 0.0, 1.0], size=B).astype(np.float32)
        go = rng.rand(B) < 0.2
        err = torch.empty(B, dtype=torch.float64, device=dev)
        loss = net.learn_from_batch(dev_tensor(obs, dev), dev_tensor(nxt, dev), B,
                                    dev_tensor(actions, dev, np.int32), dev_tensor(rewards, dev),
                                    dev_tensor(go, dev, np.uint8), 0.99, td_errors=err, double_dqn=True)
        ref = o.learn_from_batch(obs, nxt, actions, rewards, go, 0.99, None, True)
        net.check_status()
        np.testing.assert_allclose(loss.item(), ref["loss"], rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(err.cpu().numpy(), ref["td_errors"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(net.norm.item(), ref["norm"], rtol=1e-3)
        if clip:
            assert ref["norm"] > clip                        # the clip is active in this case
        _cmp_named(_grads_named(net), o.grads(), rtol=2e-3, atol=2e-5 * max(1.0, min(ref["norm"], clip or 1e9)))
        _cmp_named(net.params.named_arrays(), o.weights(), rtol=1e-3, atol=2e-5)
        if step == 1:
            net.update_target(0.25)
            o.update_target(0.25)
    tgt = net.params.named_arrays(net.target)
    np.testing.assert_allclose(tgt["main/dueling_q_values_head/fc1/kernel"][1], o.target[1].a1.W, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_adam_mix_norm_vs_oracle(rlx, dev):
    import torch
    from tests.util import dev_tensor
    rng = np.random.RandomState(2)
    n = 100003
    w = rng.randn(n).astype(np.float32)
    g = rng.randn(n).astype(np.float32)
    # pad to a 16-byte multiple like FlatParams does
    wd, gd = dev_tensor(w, dev), dev_tensor(g, dev)
    m = torch.empty(n, dtype=torch.float32, device=dev)
    v = torch.empty(n, dtype=torch.float32, device=dev)
    st = torch.empty(2, dtype=torch.float32, device=dev)
    rlx.adam_init(m, v, n, st, 0.9, 0.999, 0)
    opt = AdamTF1(n, 3e-4, 0.9, 0.999, 1e-5)
    wo = w.copy()
    for step in range(4):
        rlx.adam_tf1(wd, gd, m, v, n, 3e-4, 0.9, 0.999, 1e-5, st, 0.5, 0)
        opt.step(wo, g, 0.5)
    np.testing.assert_allclose(wd.cpu().numpy(), wo, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(st.cpu().numpy(), [opt.b1p, opt.b2p], rtol=1e-6)
    norm = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(4096, dtype=torch.float32, device=dev)
    rlx.global_norm(gd, n, norm, ws, 4096, 0)
    np.testing.assert_allclose(norm.item(), np.sqrt((g.astype(np.float64) ** 2).sum()), rtol=1e-5)
    t = dev_tensor(w, dev)
    rlx.mix_weights(t, gd, n, 0.005, 0)
    assert np.array_equal(t.cpu().numpy(), np.float32(0.005) * g + np.float32(1 - 0.005) * w)
    rlx.mix_weights(t, gd, n, 1.0, 0)
    assert np.array_equal(t.cpu().numpy(), g)


@pytest.mark.gpu
def test_hip_clipped_ppo_continuous_net_vs_oracle(rlx, dev):
    """Mujoco_ClippedPPO head (ppo_head.py:118-144): policy_mean Dense + state-independent log_std."""
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    from oracle.agents import ClippedPPOOracle
    from tests.util import dev_tensor
    np.random.seed(9)
    rng = np.random.RandomState(2)
    D, A, B = 17, 6, 64
    net = ClippedPPONet(dev, (D,), A, seed=3, continuous=True, beta_entropy=0.01)
    net.params.w("main/ppo_head/policy_log_std").copy_(dev_tensor(rng.randn(A) * 0.2, dev, np.float32))
    o = ClippedPPOOracle(net.params.named_arrays(), (D,), A, continuous=True)
    obs = rng.randn(B, D).astype(np.float32)
    obs_d = dev_tensor(obs, dev)
    frozen = o.clone_policy_continuous()
    net.update_target(1.0)
    for step in range(3):
        actions = rng.randn(B, A).astype(np.float32)
        adv = rng.randn(B).astype(np.float32)
        vt = rng.randn(B).astype(np.float32)
        om, os_ = net.policy_mean_std(obs_d, B, use_target=True, tag="old")
        rm, rs = o.policy_mean_std(obs, frozen)
        np.testing.assert_allclose(om.cpu().numpy(), rm, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(os_.cpu().numpy(), rs, rtol=1e-6)
        sc = net.train_minibatch(obs_d, B, dev_tensor(actions, dev), dev_tensor(adv, dev), dev_tensor(vt, dev),
                                 (om, os_)).cpu().numpy()
        ref = o.train_minibatch(obs, actions, adv, vt, (rm, rs))
        np.testing.assert_allclose(sc[:5], [ref["surrogate"], ref["entropy"], ref["kl"], ref["total"],
                                            ref["value_loss"]], rtol=5e-4, atol=2e-6)
        np.testing.assert_allclose(net.norm.item(), ref["norm"], rtol=1e-3)
        np.testing.assert_allclose(net.params.w("main/ppo_head/policy_log_std").cpu().numpy(), o.log_std,
                                   rtol=1e-4, atol=2e-6)
        w, wo = net.params.named_arrays(), o.weights()
        for name, towers in wo.items():
            for t, arr in towers.items():
                np.testing.assert_allclose(w[name][t], arr, rtol=1e-3, atol=2e-5, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("obs_shape,B,A", [((84, 84, 4), 64, 6), ((84, 84, 4), 16, 3), ((17,), 200, 3)])
def test_ppo_heads_forward_inside_the_dense_reduction_equals_the_separate_launch(rlx, dev, obs_shape, B, A):
    """rlx_gemm_desc.row_heads: the split-K reduction of the last dense layer finishes whole rows and computes the value /
    policy head outputs on them (or, where the product is not split that way, rlx_gemm launches the heads behind it):
    head outputs, dense activations, the losses and every gradient equal the separate heads launch bit for bit."""
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    from tests.util import dev_tensor
    rng = np.random.RandomState(B + A)
    image = len(obs_shape) == 3
    obs = rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8) if image else \
        rng.randn(B, *obs_shape).astype(np.float32)
    obs_d = dev_tensor(obs, dev)
    actions = dev_tensor(rng.randint(0, A, size=B), dev, np.int32)
    adv, vt = dev_tensor(rng.randn(B).astype(np.float32), dev), dev_tensor(rng.randn(B).astype(np.float32), dev)
    got = []
    saved = ClippedPPONet.HEADS_FORWARD_WITH_TORSO
    try:
        for fused in (False, True):
            ClippedPPONet.HEADS_FORWARD_WITH_TORSO = fused
            np.random.seed(5)
            net = ClippedPPONet(dev, obs_shape, A, seed=3)
            net.update_target(1.0)
            w = net.params.w(net.pi_head.kname)
            w.add_(dev_tensor((np.random.RandomState(9).randn(*w.shape) * 0.05).astype(np.float32), dev))
            old = net.policy_probs(obs_d, B, use_target=True, tag="old")
            net.forward_backward(obs_d, B, actions, adv, vt, old)
            net.check_status()
            ctx = net.ctx
            got.append([net.params.grads.cpu().numpy().copy(), net.scalars[:5].cpu().numpy().copy(),
                        ctx.buffer(net.v_head.name, (1, B, 1), tag="train").cpu().numpy().copy(),
                        ctx.buffer(net.pi_head.name, (1, B, A), tag="train").cpu().numpy().copy()])
    finally:
        ClippedPPONet.HEADS_FORWARD_WITH_TORSO = saved
    assert np.abs(got[0][0]).max() > 0 and np.abs(got[0][3]).max() > 0
    for a, b in zip(*got):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(32, 6), (32, 4)])
def test_dqn_head_forward_inside_the_dense_reduction_equals_the_separate_launch(rlx, dev, B, A):
    """The image DQN update with online(s) and target(s') as two towers of the same launches: the Q head of both copies
    computed by the last dense layer's split-K reduction (rlx_gemm_desc.row_heads, batch_inner addressing) — loss, TD
    errors, every gradient and every weight after the Adam step equal the separate head launch bit for bit."""
    import torch
    from coach_amd.nn.networks import DQNNet
    from tests.util import dev_tensor
    rng = np.random.RandomState(B + A)
    shape = (84, 84, 4)
    both = dev_tensor(rng.randint(0, 256, size=(2, B) + shape).astype(np.uint8), dev)
    actions = dev_tensor(rng.randint(0, A, size=B), dev, np.int32)
    rewards = dev_tensor(rng.randn(B).astype(np.float32), dev)
    overs = dev_tensor((rng.rand(B) < 0.1).astype(np.uint8), dev, np.uint8)
    got = []
    saved = DQNNet.HEAD_FORWARD_WITH_TORSO
    try:
        for fused in (False, True):
            DQNNet.HEAD_FORWARD_WITH_TORSO = fused
            np.random.seed(5)
            net = DQNNet(dev, shape, A, seed=3)
            net.update_target(1.0)
            w = net.params.w(net.q_head.kname)
            w.add_(dev_tensor((np.random.RandomState(9).randn(*w.shape) * 0.05).astype(np.float32), dev))
            td = torch.zeros(B, dtype=torch.float64, device=dev)
            net.learn_from_batch(both[0], both[1], B, actions, rewards, overs, 0.99, td_errors=td, states_pair=both)
            net.check_status()
            got.append([net.params.grads.cpu().numpy().copy(), net.params.weights.cpu().numpy().copy(),
                        net.loss.cpu().numpy().copy(), td.cpu().numpy().copy()])
    finally:
        DQNNet.HEAD_FORWARD_WITH_TORSO = saved
    assert np.abs(got[0][0]).max() > 0 and np.abs(got[0][3]).max() > 0
    for a, b in zip(*got):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("obs_shape,B,A", [((84, 84, 4), 64, 6), ((17,), 200, 3), ((11,), 7, 16)])
def test_ppo_heads_losses_and_backward_as_one_launch_equal_the_two_launches(rlx, dev, obs_shape, B, A):
    """rlx_ppo_heads_loss_backward recomputes the loss-gradient rows in every workgroup of the heads' backward instead of
    reading them behind a launch boundary: gradients of the whole network, the five loss scalars and the likelihood ratios
    must equal rlx_ppo_discrete_value_losses + rlx_dense_small_backward_multi bit for bit."""
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    from tests.util import dev_tensor
    rng = np.random.RandomState(B + A)
    image = len(obs_shape) == 3
    obs = rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8) if image else \
        rng.randn(B, *obs_shape).astype(np.float32)
    obs_d = dev_tensor(obs, dev)
    actions = dev_tensor(rng.randint(0, A, size=B), dev, np.int32)
    adv, vt = dev_tensor(rng.randn(B).astype(np.float32), dev), dev_tensor(rng.randn(B).astype(np.float32), dev)
    got = []
    saved = ClippedPPONet.HEADS_LOSS_BACKWARD_ONE_LAUNCH
    try:
        for one in (False, True):
            ClippedPPONet.HEADS_LOSS_BACKWARD_ONE_LAUNCH = one
            np.random.seed(5)
            net = ClippedPPONet(dev, obs_shape, A, seed=3)
            net.update_target(1.0)
            # an old policy that differs from the new one: perturb the online policy head
            w = net.params.w(net.pi_head.kname)
            w.add_(dev_tensor((np.random.RandomState(9).randn(*w.shape) * 0.05).astype(np.float32), dev))
            old = net.policy_probs(obs_d, B, use_target=True, tag="old")
            ratio = torch.zeros(B, dtype=torch.float32, device=dev)
            clipped = torch.zeros(B, dtype=torch.float32, device=dev)
            net.forward_backward(obs_d, B, actions, adv, vt, old, ratio_out=ratio, clipped_out=clipped)
            net.check_status()
            got.append([net.params.grads.cpu().numpy().copy(), net.scalars[:5].cpu().numpy().copy(),
                        ratio.cpu().numpy(), clipped.cpu().numpy()])
    finally:
        ClippedPPONet.HEADS_LOSS_BACKWARD_ONE_LAUNCH = saved
    assert np.abs(got[0][0]).max() > 0 and not np.allclose(got[0][2], 1.0)
    for a, b in zip(*got):
        assert np.array_equal(a, b)
