"""rlx_conv_dw_u8 (coach_amd/csrc/conv_dw_u8.hip): the first convolution's weight and bias gradient straight from the
uint8 frames — one workgroup per (image, pair of kernel rows), one deferred split per image — against the numpy oracle
(oracle.nn.im2col: dW = cols(frames / 255)^T dz, db = column sums of dz; tf.gradients through tf.layers.conv2d,
rl_coach/architectures/tensorflow_components/layers.py:108-121, architecture.py:187-220), and inside a Clipped-PPO
minibatch update against the implicit-im2col product of rlx_gemm it replaces."""
import ctypes

import numpy as np
import pytest

from oracle import nn as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,KH,S,T", [(64, 84, 84, 8, 4, 2),      # BASELINE C2 / C3: the Atari torso's conv1
                                          (5, 36, 36, 8, 4, 2),       # batch not a multiple of 8 (XCD-aware block order)
                                          (9, 24, 36, 4, 4, 1),       # one tower of 64 filters, 4 kernel rows, OH != OW
                                          (2, 84, 84, 8, 4, 2)])
def test_conv_dw_u8_matches_oracle(rlx, dev, B, H, W, KH, S, T):
    import torch
    from coach_amd import _rlx
    C, KW = 4, 8
    Co = 64 // T
    assert rlx.conv_dw_u8_supported(B, H, W, C, KH, KW, S, Co, T) == 1
    rng = np.random.RandomState(B + H)
    frames = rng.randint(0, 256, size=(B, H, W, C)).astype(np.uint8)
    OH, OW = (H - KH) // S + 1, (W - KW) // S + 1
    P, K = OH * OW, KH * KW * C
    dz = rng.randn(T, B * P, Co).astype(np.float32)
    # the oracle's im2col takes square kernels: gather the KH x KW patches here, in its (ky, kx, c) order
    x = frames.astype(np.float32) / np.float32(255.0)
    cols = np.empty((B, OH, OW, KH, KW, C), dtype=np.float32)
    for ky in range(KH):
        for kx in range(KW):
            cols[:, :, :, ky, kx, :] = x[:, ky:ky + S * OH:S, kx:kx + S * OW:S, :]
    cols = cols.reshape(B * P, K)
    if KH == KW:
        ref_cols, _, _ = N.im2col(x, KH, S)
        assert np.array_equal(ref_cols, cols)
    dw_ref = np.stack([cols.astype(np.float64).T @ dz[t].astype(np.float64) for t in range(T)])
    db_ref = dz.astype(np.float64).sum(axis=1)

    need = ctypes.c_longlong()
    rlx.conv_dw_u8_workspace_floats(B, H, W, C, KH, KW, S, Co, T, ctypes.byref(need))
    ws = torch.full((need.value,), float("nan"), dtype=torch.float32, device=dev)
    dw = torch.full((T, K, Co), float("nan"), dtype=torch.float32, device=dev)
    db = torch.full((T, Co), float("nan"), dtype=torch.float32, device=dev)
    job = _rlx.SplitkJob()
    s_ = _rlx.current_stream()
    rlx.conv_dw_u8(torch.from_numpy(frames).to(dev), 255.0, torch.from_numpy(dz).to(dev), B * P * Co, B, H, W, C, KH, KW, S, Co,
                   T, dw, K * Co, db, Co, ws, need.value, ctypes.byref(job), s_)
    assert job.splits == B and job.M == K and job.N == 64 and job.n_fold == Co
    _rlx.splitk_reduce_jobs([job], s_)
    torch.cuda.synchronize()
    got_w, got_b = dw.cpu().numpy(), db.cpu().numpy()
    scale = np.abs(dw_ref).max()
    assert np.isfinite(got_w).all() and np.isfinite(got_b).all()
    # fp32 sums of B * P products of O(1) numbers: a few ulp of the largest partial sum
    np.testing.assert_allclose(got_w, dw_ref, rtol=2e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(got_b, db_ref, rtol=2e-5, atol=2e-6 * np.abs(db_ref).max())


@pytest.mark.parametrize("B", [32, 5, 64])
def test_conv_dw_u8_one_tower_of_32_filters_matches_oracle(rlx, dev, B):
    """The DQN update's conv1 (one tower of 32 filters): conv_dw_u8_body_half — waves = kernel row x half of the output
    rows, TWO splits of [K][32] per image."""
    import torch
    from coach_amd import _rlx
    H = W = 84
    C, KH, KW, S, Co, T = 4, 8, 8, 4, 32, 1
    assert rlx.conv_dw_u8_supported(B, H, W, C, KH, KW, S, Co, T) == 1
    rng = np.random.RandomState(B)
    frames = rng.randint(0, 256, size=(B, H, W, C)).astype(np.uint8)
    OH = OW = 20
    P, K = OH * OW, KH * KW * C
    dz = rng.randn(B * P, Co).astype(np.float32)
    cols, _, _ = N.im2col(frames.astype(np.float32) / np.float32(255.0), KH, S)
    dw_ref = cols.astype(np.float64).T @ dz.astype(np.float64)
    db_ref = dz.astype(np.float64).sum(axis=0)
    need = ctypes.c_longlong()
    rlx.conv_dw_u8_workspace_floats(B, H, W, C, KH, KW, S, Co, T, ctypes.byref(need))
    ws = torch.full((need.value,), float("nan"), dtype=torch.float32, device=dev)
    dw = torch.full((K, Co), float("nan"), dtype=torch.float32, device=dev)
    db = torch.full((Co,), float("nan"), dtype=torch.float32, device=dev)
    job = _rlx.SplitkJob()
    s_ = _rlx.current_stream()
    rlx.conv_dw_u8(torch.from_numpy(frames).to(dev), 255.0, torch.from_numpy(dz).to(dev), B * P * Co, B, H, W, C, KH, KW, S, Co,
                   T, dw, K * Co, db, Co, ws, need.value, ctypes.byref(job), s_)
    assert job.splits == 2 * B and job.M == K and job.N == 32
    _rlx.splitk_reduce_jobs([job], s_)
    torch.cuda.synchronize()
    got_w, got_b = dw.cpu().numpy(), db.cpu().numpy()
    assert np.isfinite(got_w).all() and np.isfinite(got_b).all()
    np.testing.assert_allclose(got_w, dw_ref, rtol=2e-5, atol=2e-6 * np.abs(dw_ref).max())
    np.testing.assert_allclose(got_b, db_ref, rtol=2e-5, atol=2e-6 * np.abs(db_ref).max())


def test_unsupported_shapes_are_refused(rlx):
    assert rlx.conv_dw_u8_supported(64, 84, 84, 3, 8, 8, 4, 32, 2) == 0        # a patch row is not 32 bytes
    assert rlx.conv_dw_u8_supported(1, 84, 84, 4, 8, 8, 4, 32, 2) == 0         # one image: nothing to defer
    assert rlx.conv_dw_u8_supported(200, 84, 84, 4, 8, 8, 4, 32, 2) == 0       # more images than deferred splits
    assert rlx.conv_dw_u8_supported(65, 84, 84, 4, 8, 8, 4, 32, 1) == 0        # one tower of 32 filters: 2 B splits <= 128
    assert rlx.conv_dw_u8_supported(32, 36, 36, 4, 8, 8, 4, 32, 1) == 0        # ... and the Atari geometry only
    assert rlx.conv_dw_u8_supported(32, 84, 84, 4, 8, 8, 4, 16, 1) == 0        # 16 folded channels


def _update(dev, B, flag):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    shape, A = (84, 84, 4), 6
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    saved, G.CONV_DW_U8 = G.CONV_DW_U8, flag
    saved_one, G.CONV_DW_ONE_LAUNCH = G.CONV_DW_ONE_LAUNCH, False      # (the kernel as a launch of its own)
    try:
        np.random.seed(1)
        net = ClippedPPONet(dev, shape, A, seed=2, activation="tanh")
        net.update_target(1.0)
        old = net.policy_probs(obs, B, use_target=True, tag="old")
        with _rlx.KernelTimer(128) as timer:
            net.forward_backward(obs, B, acts, adv, vt, old)
        grads = net.params.grads.clone()
        net.finish_update(1.0)
        net.check_status()
        return grads, net.params.weights.clone(), [n for n, _ in timer.records], net
    finally:
        G.CONV_DW_U8 = saved
        G.CONV_DW_ONE_LAUNCH = saved_one


@pytest.mark.parametrize("B", [64, 63])
def test_ppo_update_takes_the_kernel_and_agrees_with_the_tiled_product(rlx, dev, B):
    ref = _update(dev, B, False)
    new = _update(dev, B, True)
    assert sum("conv_dw_u8_kernel" in n for n in new[2]) == 1, new[2]
    assert not any("conv_dw_u8" in n for n in ref[2]), ref[2]
    g0, g1 = ref[0].cpu().numpy(), new[0].cpu().numpy()
    differ = g0 != g1
    assert differ.any()                                   # another summation order over the batch ...
    net = new[3]
    conv1 = np.zeros(g0.shape, dtype=bool)
    first = [k for k in net.params.entries if k.endswith("/kernel")][0]          # the torso's first layer
    for key in (first, first[:-len("kernel")] + "bias"):
        off, shape, towers, stride = net.params.entries[key]
        assert shape in ((256, 32), (32,)), (key, shape)
        for t in range(towers):
            conv1[off + t * stride: off + t * stride + int(np.prod(shape))] = True
    assert not differ[~conv1].any()                       # ... of the first convolution's gradient and nothing else
    np.testing.assert_allclose(g1[conv1], g0[conv1], rtol=1e-4, atol=2e-6 * np.abs(g0[conv1]).max())
    assert np.abs(g0[conv1]).max() > 0


def test_dqn_update_takes_the_lds_resident_backward_and_agrees_with_the_tiled_one(rlx, dev):
    """DQNNet.learn_from_batch at the C3 shape (B = 32, one tower): with the fused input-gradient chain allowed at 64
    half-image workgroups the backward pass of the torso is rlx_conv32_input_grad + ONE rlx_conv_dw_multi launch (conv1
    through conv_dw_u8_body_half) instead of two dW + dX pairs, two col2im launches and the register-staged uint8 product;
    the gradients agree within another fp32 summation order (tests/tolerances.py)."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import DQNNet
    B, A, shape = 32, 4, (84, 84, 4)
    rng = np.random.RandomState(0)
    both = torch.from_numpy(rng.randint(0, 256, size=(2, B) + shape).astype(np.uint8)).to(dev)
    s, s2 = both[0], both[1]
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    rew = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    done = torch.from_numpy((rng.rand(B) < 0.2).astype(np.uint8)).to(dev)
    res = {}
    saved = G.FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS
    for wg in (1 << 30, 64):
        G.FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS = wg
        try:
            net = DQNNet(dev, shape, A, seed=2)
            with _rlx.KernelTimer(256) as timer:
                net.learn_from_batch(s, s2, B, acts, rew, done, 0.99, states_pair=both)
            net.check_status()
            res[wg] = (net.params.grads.clone(), net.params.weights.clone(), [n for n, _ in timer.records])
        finally:
            G.FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS = saved
    old, new = res[1 << 30], res[64]
    assert any("col2im" in n for n in old[2]) and not any("conv_dw_multi" in n for n in old[2]), old[2]
    assert sum("conv_dw_multi_kernel" in n for n in new[2]) == 1 and sum("conv32_input_grad" in n for n in new[2]) == 1, new[2]
    assert not any(k in n for n in new[2] for k in ("col2im", "gemm_fast_kernel", "gemm_dma_pair_kernel<true")), new[2]
    g0, g1 = old[0].cpu().numpy(), new[0].cpu().numpy()
    assert np.abs(g0).max() > 0
    np.testing.assert_allclose(g1, g0, rtol=1e-4, atol=2e-6 * np.abs(g0).max())
    np.testing.assert_allclose(new[1].cpu().numpy(), old[1].cpu().numpy(), rtol=0, atol=1e-6)
