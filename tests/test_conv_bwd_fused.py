"""The input-gradient chain of the Atari torso's third and second convolution as ONE launch (rlx_conv32_input_grad,
coach_amd/csrc/conv_bwd_fused.hip: both column matrices stay in LDS) against what it replaces per layer — the dcol product
inside the dW + dcol pair launch and rlx_col2im (tf.gradients through tf.layers.conv2d,
architectures/tensorflow_components/layers.py:108-121, architecture.py:312-385).

With 32-row tiles throughout (rlx_conv32_tail_tiles(0)) every dcol element is the same fp32 MFMA chain over K = 64 and the
gathers add the taps in col2im's order, so every gradient of a Clipped-PPO minibatch update — and every weight after the
Adam step — must be BIT-IDENTICAL.  The default (16-row tail tiles on v_mfma_f32_16x16x4_f32) sums the k of the tail rows in
another order: held to a few ulp of each gradient tensor's scale here, and to tests/tolerances.py against the oracle in
tests/test_ppo_full_size.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _update(dev, B, A, act, fused, steps=2, tail16=0):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    shape = (84, 84, 4)
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    saved_fuse, G.FUSE_CONV_INPUT_GRADS = G.FUSE_CONV_INPUT_GRADS, fused
    saved_dw, G.CONV_DW_F32 = G.CONV_DW_F32, False           # (the weight gradients through rlx_gemm on both sides: another
    _rlx.lib().conv32_tail_tiles(tail16)
    try:                                                     # order of their sums is tests/test_conv_dw_f32.py's subject)
        np.random.seed(1)
        net = ClippedPPONet(dev, shape, A, seed=2, activation=act)
        net.update_target(1.0)
        old = net.policy_probs(obs, B, use_target=True, tag="old")
        for _ in range(steps - 1):
            net.forward_backward(obs, B, acts, adv, vt, old)
            net.finish_update(1.0)
        with _rlx.KernelTimer(128) as timer:
            net.forward_backward(obs, B, acts, adv, vt, old)
        grads = net.params.grads.clone()
        net.finish_update(1.0)
        net.check_status()
        return grads, net.params.weights.clone(), net.scalars.clone(), [n for n, _ in timer.records]
    finally:
        G.FUSE_CONV_INPUT_GRADS = saved_fuse
        G.CONV_DW_F32 = saved_dw
        _rlx.lib().conv32_tail_tiles(1)                      # (the library's default)


@pytest.mark.parametrize("B", [64, 72, 63])
def test_fused_input_gradient_chain_is_bit_identical_in_a_ppo_update(rlx, dev, B):
    import torch
    ref = _update(dev, B, 6, "tanh", False)
    new = _update(dev, B, 6, "tanh", True)
    assert sum("conv32_input_grad_kernel" in n for n in new[3]) == 1, new[3]
    assert not any("conv32" in n for n in ref[3]) and sum("col2im" in n for n in ref[3]) == 2, ref[3]
    assert not any("col2im" in n for n in new[3]), new[3]
    g0, g1 = ref[0], new[0]
    assert torch.equal(g0, g1), "%d of %d gradient elements differ, max %g" % (
        int((g0 != g1).sum()), g0.numel(), float((g0 - g1).abs().max()))
    assert torch.equal(ref[1], new[1]) and torch.equal(ref[2], new[2])
    assert float(g0.abs().max()) > 0


def test_default_tail_tiles_in_a_ppo_update(rlx, dev):
    """the default form (16-row tail tiles) against the per-layer launches: every gradient tensor within 1e-5 of its scale,
    the weights after two Adam steps within tests/tolerances.py."""
    import torch
    from tolerances import WEIGHTS
    ref = _update(dev, 64, 6, "tanh", False)
    new = _update(dev, 64, 6, "tanh", True, tail16=1)
    assert any("conv32_input_grad_kernel<true" in n for n in new[3]), new[3]
    g0, g1 = ref[0], new[0]
    assert not torch.equal(g0, g1)
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())
    np.testing.assert_allclose(new[1].cpu().numpy(), ref[1].cpu().numpy(), **WEIGHTS)
    np.testing.assert_allclose(new[2][:5].cpu().numpy(), ref[2][:5].cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_small_batches_keep_the_per_layer_launches(rlx, dev):
    """fewer half images than workgroups the chip wants (FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS): the pair launches stay"""
    new = _update(dev, 12, 4, "tanh", True)           # two towers x 12 images = 48 half images < 64
    assert not any("conv32" in n for n in new[3]) and sum("col2im" in n for n in new[3]) == 2, new[3]
    new = _update(dev, 32, 4, "tanh", True)           # 128 half images (round 6: taken from 64 on, the DQN update's count)
    assert any("conv32" in n for n in new[3]) and not any("col2im" in n for n in new[3]), new[3]


def test_the_kernel_alone_against_gemm_and_col2im(rlx, dev):
    """rlx_conv32_input_grad on random operands (relu derivative: exact zeros and ones) against rlx_gemm + rlx_col2im."""
    import torch
    from coach_amd import _rlx
    B, T = 20, 2
    g = torch.Generator(device="cpu").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    dz3, w3, w2 = rnd(T, B * 49, 64), rnd(T, 576, 64) * 0.1, rnd(T, 512, 64) * 0.1
    y2, y1 = torch.relu(rnd(T, B * 81, 64)), torch.relu(rnd(T, B * 400, 32))
    lib, s = rlx, _rlx.current_stream()
    ws = torch.empty(1 << 22, device=dev)
    outs = {}
    lib.conv32_tail_tiles(0)                 # (32-row tiles throughout: the bit-identical form; the default is restored by
    for act, code in (("relu", _rlx.ACT["relu"]), ("tanh", _rlx.ACT["tanh"])):   # the test of the tail tiles below)
        yy2, yy1 = (y2, y1) if act == "relu" else (torch.tanh(y2 - 0.5), torch.tanh(y1 - 0.5))
        dcol3 = torch.empty(T, B * 49, 576, device=dev)
        _rlx.gemm(B * 49, 576, 64, dz3, w3, dcol3, b_strides=(1, 64), batch=T, a_batch_stride=B * 49 * 64,
                  b_batch_stride=576 * 64, c_batch_stride=B * 49 * 576, workspace=ws)
        dz2 = torch.empty(T, B * 81, 64, device=dev)
        lib.col2im(dcol3, dz2, yy2, code, T * B, 9, 9, 64, 3, 3, 1, s)
        dcol2 = torch.empty(T, B * 81, 512, device=dev)
        _rlx.gemm(B * 81, 512, 64, dz2, w2, dcol2, b_strides=(1, 64), batch=T, a_batch_stride=B * 81 * 64,
                  b_batch_stride=512 * 64, c_batch_stride=B * 81 * 512, workspace=ws)
        dz1 = torch.empty(T, B * 400, 32, device=dev)
        lib.col2im(dcol2, dz1, yy1, code, T * B, 20, 20, 32, 4, 4, 2, s)
        f2, f1 = torch.full_like(dz2, float("nan")), torch.full_like(dz1, float("nan"))
        lib.conv32_input_grad(dz3, B * 49 * 64, w3, 576 * 64, yy2, B * 81 * 64, f2, B * 81 * 64, w2, 512 * 64,
                              yy1, B * 400 * 32, f1, B * 400 * 32, B, T, code, s)
        torch.cuda.synchronize()
        for name, a, b in (("dz2", dz2, f2), ("dz1", dz1, f1)):
            assert torch.equal(a, b), "%s %s: %d of %d elements differ, max %g" % (
                act, name, int((a != b).sum()), a.numel(), float((a - b).abs().nan_to_num(1e9).max()))
        assert float(dz1.abs().max()) > 0
    lib.conv32_tail_tiles(1)


def test_sixteen_row_tail_tiles_against_the_bit_identical_form(rlx, dev):
    """rlx_conv32_tail_tiles(1): the second row tile of both products as a 16-row tile on v_mfma_f32_16x16x4_f32 (two
    accumulator chains per wave).  Rows >= 32 of a column matrix sum their 64 k in another order: every gradient within a
    few ulp of the result's scale of the default form, and the positions whose taps all come from rows < 32 bit-identical."""
    import torch
    from coach_amd import _rlx
    B, T = 20, 2
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    dz3, w3, w2 = rnd(T, B * 49, 64), rnd(T, 576, 64) * 0.1, rnd(T, 512, 64) * 0.1
    y2, y1 = torch.tanh(rnd(T, B * 81, 64)), torch.tanh(rnd(T, B * 400, 32))
    s = _rlx.current_stream()
    code = _rlx.ACT["tanh"]
    out = {}
    try:
        for tail in (0, 1):
            for ahead in (1, 2):             # (operands requested one or two jobs ahead: the same sums)
                rlx.conv32_tail_tiles(tail)
                rlx.conv32_prefetch(ahead)
                f2, f1 = torch.full_like(y2, float("nan")), torch.full_like(y1, float("nan"))
                rlx.conv32_input_grad(dz3, B * 49 * 64, w3, 576 * 64, y2, B * 81 * 64, f2, B * 81 * 64, w2, 512 * 64,
                                      y1, B * 400 * 32, f1, B * 400 * 32, B, T, code, s)
                torch.cuda.synchronize()
                if ahead == 1:
                    out[tail] = (f2, f1)
                else:
                    assert torch.equal(f2, out[tail][0]) and torch.equal(f1, out[tail][1]), (tail, ahead)
    finally:
        rlx.conv32_tail_tiles(1)             # (the library's defaults)
        rlx.conv32_prefetch(1)
    for name, a, b in (("dz2", out[0][0], out[1][0]), ("dz1", out[0][1], out[1][1])):
        assert not torch.isnan(b).any(), name
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 4e-6 * scale, (name, float((a - b).abs().max()), scale)
        assert not torch.equal(a, b), name                      # (another order of sums somewhere: the variant did run)
    # the first rows of the upper half image depend on full-tile rows only
    a, b = out[0][1].view(T, B, 20, 20, 32), out[1][1].view(T, B, 20, 20, 32)
    assert torch.equal(a[:, :, :2], b[:, :, :2])
