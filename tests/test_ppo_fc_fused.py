"""rlx_ppo_fc_heads (csrc/ppo_fc_fused.hip): the last dense layer of both towers + heads + losses + heads' backward as one
launch, against the three-launch path it replaces (rlx_gemm with row_heads + rlx_ppo_heads_loss_backward) on identical
inputs: scalars, head outputs, every gradient.  Another fp32 summation order (8 K splits of 392 instead of 25 tiles), so
the bounds are tests/tolerances.py's, not bit equality.  The oracle comparisons at the C2 size run through this launch by
default (tests/test_ppo_full_size.py, tests/test_ppo_long_episodes.py)."""
import numpy as np
import pytest

from tolerances import LOSS, OUT


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _run(net, fused, obs, B, actions, adv, vt, old):
    net.FC_HEADS_ONE_LAUNCH = fused          # (instance attribute: the class default is off)
    net.HEADS_ROW_LOCAL = False              # the reference here is the three-launch path (tests/test_ppo_fc_rows.py: the row-local one)
    net.params.grads.zero_()
    net.scalars.zero_()
    net.forward_backward(obs, B, actions, adv, vt, old)
    import torch
    torch.cuda.synchronize()
    return net.scalars[:5].cpu().numpy().copy(), net.params.grads.cpu().numpy().copy()


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(64, 6), (37, 4), (8, 16)])
def test_fc_heads_one_launch_matches_the_three_launch_path(dev, B, A):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn.networks import ClippedPPONet
    np.random.seed(11)
    shape = (84, 84, 4)
    net = ClippedPPONet(dev, shape, A, seed=5)
    assert _rlx.lib().ppo_fc_heads_supported(B, 3136, 512, A)
    rng = np.random.RandomState(B)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    actions = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    net.update_target(1.0)
    # an old policy that differs from the current one (ratios off 1, some clipped)
    net.params.weights.add_(torch.from_numpy((rng.randn(net.params.size) * 2e-3).astype(np.float32)).to(dev))
    old = net.policy_probs(obs, B, use_target=True, tag="old").clone()
    s0, g0 = _run(net, False, obs, B, actions, adv, vt, old)
    s1, g1 = _run(net, True, obs, B, actions, adv, vt, old)
    assert np.abs(s0).max() > 0 and np.abs(g0).max() > 0
    np.testing.assert_allclose(s1, s0, **LOSS)
    # gradients: relative to each tensor's own scale (sums of 64 x O(1e-2) products: another order moves the last bits)
    for name, (off, shp, towers, stride) in net.params.entries.items():
        n = int(np.prod(shp))
        for t in range(towers):
            a, b = g1[off + t * stride: off + t * stride + n], g0[off + t * stride: off + t * stride + n]
            scale = max(np.abs(b).max(), 1e-12)
            assert np.abs(a - b).max() <= 2e-5 * scale + 1e-9, (name, t, np.abs(a - b).max(), scale)
    # head outputs of the fused launch against a plain fp64 evaluation of its own h
    ctx = net.ctx
    h = ctx.buffer(net.torso.layers[-1].name, (2, B, 512), tag="train").cpu().numpy().astype(np.float64)
    w = net.params.named_arrays()
    v = h[0] @ w["main/v_head/dense/kernel"][0].astype(np.float64) + w["main/v_head/dense/bias"][0]
    lg = h[1] @ w["main/ppo_head/policy_fc/kernel"][0].astype(np.float64) + w["main/ppo_head/policy_fc/bias"][0]
    np.testing.assert_allclose(ctx.buffer("main/v_head/dense", (1, B, 1), tag="train").cpu().numpy().reshape(B), v[:, 0], **OUT)
    np.testing.assert_allclose(ctx.buffer("main/ppo_head/policy_fc", (1, B, A), tag="train").cpu().numpy().reshape(B, A), lg, **OUT)
    net.check_status()
    # twice in a row: the tickets re-arm themselves
    s2, g2 = _run(net, True, obs, B, actions, adv, vt, old)
    np.testing.assert_array_equal(s2, s1)
    np.testing.assert_array_equal(g2, g1)
