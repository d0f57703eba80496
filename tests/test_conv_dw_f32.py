"""rlx_conv_dw_f32 (coach_amd/csrc/conv_dw_f32.hip): an inner convolution's weight and bias gradient with the image pair's
input rows and dz in LDS — one workgroup per (tower, pair of images, kernel row), one deferred split per pair — against the
numpy oracle (oracle.nn.im2col: dW = cols(x)^T dz, db = column sums of dz; tf.gradients through tf.layers.conv2d,
rl_coach/architectures/tensorflow_components/layers.py:108-121, architecture.py:187-220), and inside a Clipped-PPO
minibatch update (input gradients through rlx_conv32_input_grad) against the tiled products of rlx_gemm."""
import ctypes

import numpy as np
import pytest

from oracle import nn as N

pytestmark = pytest.mark.gpu

CONV3 = (9, 9, 64, 3, 1)        # H, W, C, kernel, stride of the Atari torso's third convolution
CONV2 = (20, 20, 32, 4, 2)      # ... and of its second


@pytest.mark.parametrize("geom", [CONV3, CONV2])
@pytest.mark.parametrize("B,T", [(64, 2), (63, 2), (5, 1), (3, 2)])
def test_conv_dw_f32_matches_oracle(rlx, dev, geom, B, T):
    import torch
    from coach_amd import _rlx
    H, W, C, k, S = geom
    Co = 64
    assert rlx.conv_dw_f32_supported(B, H, W, C, k, k, S, Co, T) == 1
    rng = np.random.RandomState(B + H + T)
    x = np.tanh(rng.randn(T, B, H, W, C)).astype(np.float32)
    cols = [N.im2col(x[t], k, S) for t in range(T)]
    OH, OW = cols[0][1], cols[0][2]
    P, K = OH * OW, k * k * C
    dz = rng.randn(T, B * P, Co).astype(np.float32)
    dw_ref = np.stack([cols[t][0].astype(np.float64).T @ dz[t].astype(np.float64) for t in range(T)])
    db_ref = dz.astype(np.float64).sum(axis=1)
    need = ctypes.c_longlong()
    rlx.conv_dw_f32_workspace_floats(B, H, W, C, k, k, S, Co, T, ctypes.byref(need))
    ws = torch.full((need.value,), float("nan"), dtype=torch.float32, device=dev)
    dw = torch.full((T, K, Co), float("nan"), dtype=torch.float32, device=dev)
    db = torch.full((T, Co), float("nan"), dtype=torch.float32, device=dev)
    job = _rlx.SplitkJob()
    s_ = _rlx.current_stream()
    rlx.conv_dw_f32(torch.from_numpy(x).to(dev), B * H * W * C, torch.from_numpy(dz).to(dev), B * P * Co, B, H, W, C, k, k, S, Co, T,
                    dw, K * Co, db, Co, ws, need.value, ctypes.byref(job), s_)
    assert job.splits == (B + 1) // 2 and job.M == K and job.N == 64 and job.batch == T and job.n_fold == 0
    _rlx.splitk_reduce_jobs([job], s_)
    torch.cuda.synchronize()
    got_w, got_b = dw.cpu().numpy(), db.cpu().numpy()
    assert np.isfinite(got_w).all() and np.isfinite(got_b).all()
    np.testing.assert_allclose(got_w, dw_ref, rtol=2e-5, atol=2e-6 * np.abs(dw_ref).max())
    np.testing.assert_allclose(got_b, db_ref, rtol=2e-5, atol=2e-6 * np.abs(db_ref).max())


def test_unsupported_shapes_are_refused(rlx):
    assert rlx.conv_dw_f32_supported(64, 9, 9, 64, 3, 3, 1, 32, 2) == 0       # 32 filters
    assert rlx.conv_dw_f32_supported(64, 10, 10, 64, 3, 3, 1, 64, 2) == 0     # another geometry
    assert rlx.conv_dw_f32_supported(2, 9, 9, 64, 3, 3, 1, 64, 2) == 0        # one image pair: nothing to defer
    assert rlx.conv_dw_f32_supported(300, 9, 9, 64, 3, 3, 1, 64, 2) == 0      # more pairs than deferred splits


def _update(dev, B, flag, one_launch=False):
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
    saved = G.CONV_DW_F32, G.FUSE_CONV_INPUT_GRADS, G.CONV_DW_ONE_LAUNCH
    G.CONV_DW_F32, G.FUSE_CONV_INPUT_GRADS, G.CONV_DW_ONE_LAUNCH = flag, True, one_launch
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
        return grads, [n for n, _ in timer.records], net
    finally:
        G.CONV_DW_F32, G.FUSE_CONV_INPUT_GRADS, G.CONV_DW_ONE_LAUNCH = saved


@pytest.mark.parametrize("B", [64, 63])
def test_ppo_update_takes_the_kernel_and_agrees_with_the_tiled_products(rlx, dev, B):
    ref = _update(dev, B, False)
    new = _update(dev, B, True)
    assert sum("conv_dw_f32_kernel" in n for n in new[1]) == 2, new[1]
    assert not any("conv_dw_f32" in n for n in ref[1]), ref[1]
    g0, g1 = ref[0].cpu().numpy(), new[0].cpu().numpy()
    differ = g0 != g1
    assert differ.any()
    net = new[2]
    inner = np.zeros(g0.shape, dtype=bool)
    kernels = [k for k in net.params.entries if k.endswith("/kernel")]
    for key in kernels[1:3]:                                  # the torso's second and third layer
        for kk in (key, key[:-len("kernel")] + "bias"):
            off, shape, towers, stride = net.params.entries[kk]
            assert shape in ((512, 64), (576, 64), (64,)), (kk, shape)
            for t in range(towers):
                inner[off + t * stride: off + t * stride + int(np.prod(shape))] = True
    assert not differ[~inner].any()                           # another summation order of these two gradients, nothing else
    np.testing.assert_allclose(g1[inner], g0[inner], rtol=1e-4, atol=2e-6 * np.abs(g0[inner]).max())
    assert np.abs(g0[inner]).max() > 0


@pytest.mark.parametrize("B", [64, 63])
def test_one_launch_of_the_three_layers_is_bit_identical_to_three_launches(rlx, dev, B):
    """rlx_conv_dw_multi (csrc/conv_dw_multi.hip): the same device code, the same partial sums, one grid."""
    import torch
    three = _update(dev, B, True, one_launch=False)
    try:
        rlx.conv_dw_passes(1)                # (the three launches stage in one pass; two passes: the test below)
        one = _update(dev, B, True, one_launch=True)
    finally:
        rlx.conv_dw_passes(2)                # the library's default
    assert sum("conv_dw_multi_kernel" in n for n in one[1]) == 1 and not any("conv_dw_f32_kernel" in n or "conv_dw_u8_kernel" in n
                                                                             for n in one[1]), one[1]
    assert sum("conv_dw_f32_kernel" in n for n in three[1]) == 2 and sum("conv_dw_u8_kernel" in n for n in three[1]) == 1, three[1]
    assert len(one[1]) == len(three[1]) - 2
    assert torch.equal(one[0], three[0])


@pytest.mark.parametrize("B,passes", [(64, 2), (63, 2), (64, 4), (63, 4)])
def test_two_pass_staging_of_the_one_launch_form(rlx, dev, B, passes):
    """rlx_conv_dw_passes(2): conv1's and conv2's operands in two passes over the output rows (half the LDS, two workgroups per
    CU).  Every accumulator runs through its positions in the same order — weight gradients and conv1's bias gradient bit for
    bit; conv2's bias gradient groups its column sums by pass (last bits)."""
    import torch
    try:
        rlx.conv_dw_pairs_per_workgroup(1)   # (a split per image pair in both forms: the grouping of the batch sum is not under test)
        rlx.conv_dw_passes(1)
        one = _update(dev, B, True, one_launch=True)
        rlx.conv_dw_passes(passes)
        two = _update(dev, B, True, one_launch=True)
    finally:
        rlx.conv_dw_passes(2)                # the library's defaults
        rlx.conv_dw_pairs_per_workgroup(0)
    assert sum("conv_dw_multi_kernel<%d>" % passes in n for n in two[1]) == 1, two[1]
    g1, g2 = one[0], two[0]
    net = one[2]
    kernels = [k for k in net.params.entries if k.endswith("/kernel")]
    same = torch.ones(g1.numel(), dtype=torch.bool, device=g1.device)
    for key in kernels[1:3]:                 # (conv2 and conv3 in two passes: their column sums group by pass)
        off, shape, towers, stride = net.params.entries[key[:-len("kernel")] + "bias"]
        assert shape == (64,)
        for t in range(towers):
            same[off + t * stride: off + t * stride + 64] = False
    assert torch.equal(g1[same], g2[same]), "%d elements differ outside the inner layers' bias gradients" % int((g1[same] != g2[same]).sum())
    a, b = g1[~same], g2[~same]
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) and float(a.abs().max()) > 0


@pytest.mark.parametrize("B", [64, 63, 34])
def test_two_image_pairs_per_workgroup_halve_the_splits(rlx, dev, B):
    """rlx_conv_dw_pairs_per_workgroup(2): an fp32 item's workgroup takes two image pairs into the same accumulators — half
    the splits in the deferred arena; another grouping of the sum over the batch (fp32 tolerance), conv1 untouched."""
    try:
        rlx.conv_dw_pairs_per_workgroup(1)
        ref = _update(dev, B, True, one_launch=True)
        rlx.conv_dw_pairs_per_workgroup(2)
        new = _update(dev, B, True, one_launch=True)
    finally:
        rlx.conv_dw_pairs_per_workgroup(0)              # the library's default: by the number of workgroups
    assert sum("conv_dw_multi_kernel" in n for n in new[1]) == 1, new[1]
    g0, g1 = ref[0].cpu().numpy(), new[0].cpu().numpy()
    assert np.abs(g0).max() > 0
    np.testing.assert_allclose(g1, g0, rtol=1e-4, atol=2e-6 * np.abs(g0).max())
    assert (g0 != g1).any()
