"""The convolution layers' weight gradients of a backward pass as ONE launch behind the input-gradient chain
(rlx_gemm_multi_defer, nn.graph.MULTI_DW) against one launch per layer inside its dW + dX pair (tf.gradients of the
three tf.layers.conv2d, architectures/tensorflow_components/architecture.py:187-220).  Every product keeps the tiling and
K split rlx_gemm gives it alone, so every gradient must be BIT-IDENTICAL."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,B,A", [((84, 84, 4), 64, 6), ((84, 84, 4), 32, 4), ((44, 44, 4), 16, 4)])
def test_multi_dw_equals_per_layer_launches_bit_for_bit(rlx, dev, shape, B, A):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    res = {}
    # (the products as rlx_gemm tiles them on both sides: the round-6 kernels that replaced them by default — rlx_conv_dw_u8,
    # rlx_conv_dw_f32 behind rlx_conv32_input_grad — have tests of their own)
    saved = G.CONV_DW_U8, G.CONV_DW_F32, G.FUSE_CONV_INPUT_GRADS
    for multi in (False, True):
        G.MULTI_DW = multi
        G.CONV_DW_U8 = G.CONV_DW_F32 = G.FUSE_CONV_INPUT_GRADS = False
        try:
            np.random.seed(1)
            net = ClippedPPONet(dev, shape, A, seed=2)
            net.update_target(1.0)
            old = net.policy_probs(obs, B, use_target=True, tag="old")
            net.forward_backward(obs, B, acts, adv, vt, old)            # (first pass: offset tables are built)
            net.finish_update(1.0)
            with _rlx.KernelTimer(128) as timer:
                net.forward_backward(obs, B, acts, adv, vt, old)
            grads = net.params.grads.clone()
            net.finish_update(1.0)
            net.check_status()
            res[multi] = (grads, net.params.weights.clone(), [n for n, _ in timer.records])
        finally:
            G.MULTI_DW = False
            G.CONV_DW_U8, G.CONV_DW_F32, G.FUSE_CONV_INPUT_GRADS = saved
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert float(res[True][0].abs().max()) > 0
    names = res[True][2]
    if shape[0] == 84 and B == 64:
        assert sum("gemm_multi_dw_kernel" in n for n in names) == 1, names
        assert not any("pair_kernel<true" in n for n in names), names       # no convolution dW + dX pair is left
        assert abs(len(names) - len(res[False][2])) <= 1, (names, res[False][2])      # (about) the same number of launches
