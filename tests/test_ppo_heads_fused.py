"""rlx_ppo_discrete_heads_fused (csrc/ppo_heads_fused.hip): value + policy head forward, both head losses and the
heads' backward pass in ONE launch, against the three launches it replaces (rlx_dense_small_forward_multi,
rlx_ppo_discrete_value_losses, rlx_dense_small_backward_multi).  The default form (one workgroup per head, input in
LDS) sums its dot products in another order than the dense_small kernels: the gradients of an update agree to fp32
accumulation noise, the loss scalars (same reduction tree, inputs differing in the last bits) to 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obs_shape,A,B,emb,mid", [((12,), 6, 64, [48], [512]), ((9,), 3, 20, [32], [96]),
                                                   ((84, 84, 4), 6, 64, "Medium", "Medium")])
def test_one_launch_heads_match_three_launches(dev, obs_shape, A, B, emb, mid):
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    nets = []
    for one in (True, False):
        net = ClippedPPONet(dev, obs_shape, A, embedder=emb, middleware=mid, seed=2)
        net._heads_one_launch = one
        nets.append(net)
    nets[1].params.weights.copy_(nets[0].params.weights)            # (the head initialiser draws from np.random)
    nets[1].update_target(1.0); nets[0].update_target(1.0)
    rng = np.random.RandomState(3)
    image = len(obs_shape) == 3
    for it in range(3):
        obs = rng.randint(0, 256, size=(B,) + obs_shape).astype(np.uint8) if image else \
            rng.randn(B, *obs_shape).astype(np.float32)
        actions = rng.randint(0, A, B).astype(np.int32)
        adv = rng.randn(B).astype(np.float32)
        vt = rng.randn(B).astype(np.float32)
        old = rng.rand(B, A).astype(np.float32) + 0.05
        old /= old.sum(1, keepdims=True)
        outs = []
        for net in nets:
            t = lambda x: torch.from_numpy(x).to(dev)
            ratio = torch.zeros(B, device=dev)
            sc = net.train_minibatch(t(obs), B, t(actions), t(adv), t(vt), t(old), ratio_out=ratio).clone()
            net.check_status()
            outs.append((sc, net.norm.clone(), ratio, net.params.grads.clone()))
        (sc_a, norm_a, ratio_a, g_a), (sc_b, norm_b, ratio_b, g_b) = outs
        np.testing.assert_allclose(sc_a.cpu().numpy(), sc_b.cpu().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(float(norm_a), float(norm_b), rtol=1e-5)
        np.testing.assert_allclose(ratio_a.cpu().numpy(), ratio_b.cpu().numpy(), rtol=1e-5)
        ga, gb = g_a.cpu().numpy(), g_b.cpu().numpy()
        np.testing.assert_allclose(ga, gb, rtol=2e-4, atol=2e-6 * max(1.0, float(np.abs(gb).max())))
        # keep the two nets on one trajectory: the comparison is per update, not of a chaotic 3-step history
        nets[1].params.weights.copy_(nets[0].params.weights)
        nets[1].adam.m.copy_(nets[0].adam.m); nets[1].adam.v.copy_(nets[0].adam.v)
    assert torch.isfinite(nets[0].params.weights).all()
