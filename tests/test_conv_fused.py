"""conv2 -> conv3 of the Atari torso as one launch (rlx_conv23_forward, coach_amd/csrc/conv_fused.hip) against the two
tiled implicit-im2col launches it replaces (tf.layers.conv2d twice, tensorflow_components/layers.py:108-121).

The fused kernel is taken only where rlx_gemm would run both products on 32 x 64 tiles with two wave groups per K slab,
and there its fp32 sums are the same MFMA chains in the same order: every activation must be BIT-IDENTICAL, and so must a
whole Clipped-PPO update built on it.  (Against the numpy oracle the convolution layers are covered by tests/test_nn.py and
tests/test_ppo_full_size.py, which now run through this kernel at the C2 size.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEFAULT_RING = (4, 2)        # rlx_conv23_depth's defaults (csrc/conv_fused.hip g_depth, g_step)


@pytest.fixture(autouse=True)
def _tile_rule_wave_groups():
    """This file tests the BIT-IDENTICAL form: the fused launch with the wave groups the tiled launches would have used
    (nn.graph.CONV_FORWARD_WAVE_GROUPS = None).  The default since round 6 is four groups everywhere (conv3 on all eight
    waves at the two-tower minibatch too): test_four_wave_groups_everywhere_stay_within_the_tolerances below."""
    from coach_amd.nn import graph as G
    saved, G.CONV_FORWARD_WAVE_GROUPS = G.CONV_FORWARD_WAVE_GROUPS, None
    yield
    G.CONV_FORWARD_WAVE_GROUPS = saved


def _torso(dev, act, seed):
    import torch
    from coach_amd.nn import graph as G, networks as NW
    params = G.FlatParams()
    torso, feat = NW.build_torso(params, "main", (84, 84, 4), act, 2, "Medium", "Medium")
    params.finalize(dev)
    rng = np.random.RandomState(seed)
    torso.initialize(rng)
    for l in torso.layers:                                   # non-zero biases: the epilogue's bias add is exercised
        for t in range(2):
            params.w(l.bname, t).copy_(torch.from_numpy(rng.randn(*params.w(l.bname, t).shape).astype(np.float32) * 0.1))
    return params, torso


@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("B,depth", [(64, (4, 1)), (72, (4, 1)), (64, (2, 1)), (64, (3, 1)), (64, (6, 1)), (64, (8, 1)),
                                     (64, (4, 2)), (72, (6, 2)), (64, (8, 2))])
def test_fused_pair_equals_the_two_tiled_launches_bit_for_bit(rlx, dev, act, B, depth, request):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    rlx.conv23_depth(*depth)
    request.addfinalizer(lambda: rlx.conv23_depth(*DEFAULT_RING))
    G.FUSE_CONV_FIRST = False                    # (this test: the PAIR kernel in all its ring shapes)
    request.addfinalizer(lambda: setattr(G, "FUSE_CONV_FIRST", True))
    assert G._kw2_tiling(B * 81, 64, 2) and G._kw2_tiling(B * 49, 64, 2)
    params, torso = _torso(dev, act, 1)
    obs = torch.from_numpy(np.random.RandomState(2).randint(0, 256, size=(B, 84, 84, 4)).astype(np.uint8)).to(dev)
    x = G.input_tensor(obs, B, 84 * 84 * 4, u8=True, div=255.0)
    outs, names = {}, {}
    for fused in (False, True):
        G.FUSE_CONV_PAIR = fused
        try:
            ctx = G.Context(dev)
            with _rlx.KernelTimer(64) as timer:
                acts = torso.forward(ctx, x, tag="t")
            outs[fused] = [a.data.clone() for a in acts[1:]]
            names[fused] = [n for n, _ in timer.records]
        finally:
            G.FUSE_CONV_PAIR = True
    assert not any("conv23" in n for n in names[False])
    assert sum("conv23_forward_kernel" in n for n in names[True]) == 1
    gemms = lambda ns: [n for n in ns if not n.startswith("conv_tables")]     # (offset tables: built on a context's first pass)
    assert len(gemms(names[True])) == len(gemms(names[False])) - 1          # one launch instead of two
    assert len(outs[True]) == len(outs[False]) == 4
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert a.shape == b.shape
        assert torch.equal(a, b), "layer %d: %d of %d elements differ, max %g" % (
            i, int((a != b).sum()), a.numel(), float((a - b).abs().max()))
    assert float(outs[True][2].abs().max()) > 0.05              # (not a comparison of zeros)


@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("B,towers", [(64, 1), (32, 2), (36, 2), (70, 1)])
def test_fused_pair_with_four_wave_groups_equals_the_32x32_tiling_bit_for_bit(rlx, dev, act, B, towers, request):
    """One tower of 64 images (acting) or two of 32 (the DQN update's online + target pass): rlx_gemm runs conv2 / conv3 on
    32 x 32 tiles with FOUR wave groups per K slab, one accumulator per k-quad, summed ((0 + 1) + 2) + 3 — the fused
    kernel's wave_groups = 4 form reproduces exactly that."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    assert G._tiled_wave_groups(B * 81, 64, towers) == 4 and G._tiled_wave_groups(B * 49, 64, towers) == 4
    G.FUSE_CONV_FIRST = False
    request.addfinalizer(lambda: setattr(G, "FUSE_CONV_FIRST", True))
    params, torso = _torso(dev, act, 5)
    obs = torch.from_numpy(np.random.RandomState(6).randint(0, 256, size=(B, 84, 84, 4)).astype(np.uint8)).to(dev)
    x = G.input_tensor(obs, B, 84 * 84 * 4, u8=True, div=255.0)
    kw = dict(t0=1, nt=1) if towers == 1 else {}
    outs, names = {}, {}
    for fused in (False, True):
        G.FUSE_CONV_PAIR = fused
        try:
            ctx = G.Context(dev)
            with _rlx.KernelTimer(64) as timer:
                acts = torso.forward(ctx, x, tag="t", **kw)
            outs[fused] = [a.data.clone() for a in acts[1:]]
            names[fused] = [n for n, _ in timer.records]
        finally:
            G.FUSE_CONV_PAIR = True
    assert sum("conv23_forward_kernel<4, 2, 4>" in n for n in names[True]) == 1, names[True]
    assert any("gemm_dma_kernel" in n for n in names[False]) and not any("conv23" in n for n in names[False])
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert torch.equal(a, b), "layer %d: %d of %d elements differ, max %g" % (
            i, int((a != b).sum()), a.numel(), float((a - b).abs().max()))
    assert float(outs[True][2].abs().max()) > 0.05


@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("B,towers,groups,chunks", [(64, 2, 2, 1), (72, 2, 2, 1), (63, 2, 2, 1), (64, 1, 4, 3), (70, 1, 4, 3),
                                                    (32, 2, 4, 3), (36, 2, 4, 3)])
def test_fused_triple_equals_the_tiled_launches_bit_for_bit(rlx, dev, act, B, towers, groups, chunks):
    """rlx_conv123_forward: conv1 from the uint8 frames in front of the fused pair.  Two towers of 63 .. 75 frames: conv1's own
    launch sums K = 256 in ONE chain (the towers folded into N, 64 x 64 tiles, no K split); one tower of 64 or two of 32: in
    THREE chains over 96-long chunks that a reduce launch combines — the kernel reproduces either, every activation of the
    three layers is bit-identical to the tiled launches' and to the pair kernel's behind a tiled conv1."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    params, torso = _torso(dev, act, 7)
    obs = torch.from_numpy(np.random.RandomState(8).randint(0, 256, size=(B, 84, 84, 4)).astype(np.uint8)).to(dev)
    x = G.input_tensor(obs, B, 84 * 84 * 4, u8=True, div=255.0)
    kw = dict(t0=1, nt=1) if towers == 1 else {}
    outs, names = {}, {}
    for mode, (pair, first) in {"tiled": (False, False), "pair": (True, False), "triple": (True, True)}.items():
        G.FUSE_CONV_PAIR, G.FUSE_CONV_FIRST = pair, first
        try:
            ctx = G.Context(dev)
            with _rlx.KernelTimer(64) as timer:
                acts = torso.forward(ctx, x, tag="t", **kw)
            outs[mode] = [a.data.clone() for a in acts[1:]]
            names[mode] = [n for n, _ in timer.records if not n.startswith("conv_tables")]
        finally:
            G.FUSE_CONV_PAIR = G.FUSE_CONV_FIRST = True
    want = "conv23_forward_kernel<4, 2, %d, %d>" % (groups, chunks)
    assert sum(want in n for n in names["triple"]) == 1, names["triple"]
    assert want in names["triple"][0], names["triple"]               # (the FIRST launch of the pass)
    assert any("splitk_reduce4_kernel" in n for n in names["tiled"][:2]) == (chunks == 3), names["tiled"]
    assert len(names["triple"]) == len(names["tiled"]) - (3 if chunks == 3 else 2)
    for mode in ("tiled", "pair"):
        for i, (a, b) in enumerate(zip(outs["triple"], outs[mode])):
            assert a.shape == b.shape
            assert torch.equal(a, b), "%s, layer %d: %d of %d elements differ, max %g" % (
                mode, i, int((a != b).sum()), a.numel(), float((a - b).abs().max()))
    assert len(outs["triple"]) == 4 and float(outs["triple"][0].abs().max()) > 0.05


def test_the_triple_is_not_taken_where_conv1_sums_in_another_order(rlx, dev):
    """uint8 operands through the LDS-DMA ring (rlx_gemm_pipeline(2)) pair the k of an MFMA step differently: rlx_gemm_describe
    reports it and the forward pass keeps conv1's own launch (the pair kernel still follows it)."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    params, torso = _torso(dev, "tanh", 9)
    obs = torch.from_numpy(np.random.RandomState(10).randint(0, 256, size=(64, 84, 84, 4)).astype(np.uint8)).to(dev)
    x = G.input_tensor(obs, 64, 84 * 84 * 4, u8=True, div=255.0)
    rlx.gemm_pipeline(2)
    try:
        ctx = G.Context(dev)
        with _rlx.KernelTimer(64) as timer:
            torso.forward(ctx, x, tag="t")
        names = [n for n, _ in timer.records]
    finally:
        rlx.gemm_pipeline(1)
    assert any("gemm_dma_kernel" in n for n in names) and any("conv23_forward_kernel<4, 2>" in n for n in names), names


def test_dqn_update_through_the_fused_pair_is_bit_identical(rlx, dev):
    """DQNNet.learn_from_batch at the C3 shape (B = 32, online(s) + target(s') as the two towers of one pass over the replay
    buffer's collated [2, B, ...] states — parallel_prediction, dqn_agent.py:86-89 —, relu)."""
    import torch
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import DQNNet
    B, A, shape = 32, 4, (84, 84, 4)
    rng = np.random.RandomState(0)
    s = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    s2 = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    both = torch.stack([s, s2]).contiguous()
    s, s2 = both[0], both[1]
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    rew = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    done = torch.from_numpy((rng.rand(B) < 0.2).astype(np.uint8)).to(dev)
    res = {}
    for fused in (False, True):
        G.FUSE_CONV_PAIR = fused
        try:
            net = DQNNet(dev, shape, A, seed=2)
            for _ in range(2):
                net.learn_from_batch(s, s2, B, acts, rew, done, 0.99, states_pair=both)
            net.check_status()
            res[fused] = (net.params.weights.clone(), net.loss.clone())
            if fused:                 # (conv1 included: 32 frames x (online, target) = 200 tiles of 128 x 32, three K chunks)
                from coach_amd import _rlx
                with _rlx.KernelTimer(256) as timer:
                    net.learn_from_batch(s, s2, B, acts, rew, done, 0.99, states_pair=both)
                assert any("conv23_forward_kernel<4, 2, 4, 3>" in n for n, _ in timer.records), [n for n, _ in timer.records]
        finally:
            G.FUSE_CONV_PAIR = True
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_other_batches_keep_the_tiled_launches(rlx, dev):
    """where rlx_gemm would pick another tiling (8 images: thin / split K; 256: 64 x 64 tiles) the sums would differ in their
    last bits — the fused kernel is not taken, and the pinned small-batch trajectories of the other tests do not move."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    params, torso = _torso(dev, "tanh", 3)
    for B in (8, 16, 256):
        obs = torch.from_numpy(np.random.RandomState(4).randint(0, 256, size=(B, 84, 84, 4)).astype(np.uint8)).to(dev)
        ctx = G.Context(dev)
        with _rlx.KernelTimer(64) as timer:
            torso.forward(ctx, G.input_tensor(obs, B, 84 * 84 * 4, u8=True, div=255.0), tag="t")
        assert not any("conv23" in n for n, _ in timer.records), B


def test_clipped_ppo_update_through_the_fused_pair_is_bit_identical(rlx, dev):
    import torch
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    B, A, shape = 64, 6, (84, 84, 4)
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    res = {}
    for fused in (False, True):
        G.FUSE_CONV_PAIR = fused
        try:
            np.random.seed(1)
            net = ClippedPPONet(dev, shape, A, seed=2)
            net.update_target(1.0)
            old = net.policy_probs(obs, B, use_target=True, tag="old")
            for _ in range(2):
                net.forward_backward(obs, B, acts, adv, vt, old)
                net.finish_update(1.0)
            net.check_status()
            res[fused] = (net.params.weights.clone(), net.scalars.clone())
        finally:
            G.FUSE_CONV_PAIR = True
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_four_wave_groups_everywhere_stay_within_the_tolerances(rlx, dev):
    """CONV_FORWARD_WAVE_GROUPS = 4 (the default): at the two-tower minibatch the k-quads of a slab are grouped as the 32 x 32
    tiling groups them, not as the 32 x 64 tiling the tiled launches would use there — activations within tolerances.OUT of
    the bit-identical form, weights after two updates within tolerances.WEIGHTS."""
    import torch
    from tolerances import OUT, WEIGHTS
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    B, A, shape = 64, 6, (84, 84, 4)
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    res = {}
    for groups in (None, 4):
        G.CONV_FORWARD_WAVE_GROUPS = groups
        np.random.seed(1)
        net = ClippedPPONet(dev, shape, A, seed=2)
        net.update_target(1.0)
        old = net.policy_probs(obs, B, use_target=True, tag="old")
        with _rlx.KernelTimer(64) as timer:
            net.forward_backward(obs, B, acts, adv, vt, old)
        names = [n for n, _ in timer.records]
        net.finish_update(1.0)
        net.forward_backward(obs, B, acts, adv, vt, old)
        net.finish_update(1.0)
        net.check_status()
        h = net.ctx.buffer(net.torso.layers[2].name, (2, B * 49, 64), tag="train").clone()
        res[groups] = (net.params.weights.clone(), h, names)
    assert any("conv23_forward_kernel<4, 2, 2, 1>" in n for n in res[None][2]), res[None][2]
    assert any("conv23_forward_kernel<4, 2, 4, 1>" in n for n in res[4][2]), res[4][2]
    np.testing.assert_allclose(res[4][1].cpu().numpy(), res[None][1].cpu().numpy(), **OUT)
    np.testing.assert_allclose(res[4][0].cpu().numpy(), res[None][0].cpu().numpy(), **WEIGHTS)
    assert not torch.equal(res[4][1], res[None][1])


def test_the_tile_rule_follows_rlx_gemm_tuning(rlx, dev):
    """_kw2_tiling reads the library's LIVE thresholds: after rlx_gemm_tuning moved them the fused kernel is not taken where
    the tiled launches no longer run the configuration it reproduces."""
    from coach_amd.nn import graph as G
    assert G._kw2_tiling(64 * 81, 64, 2) and G._kw2_tiling(64 * 49, 64, 2)
    rlx.gemm_tuning(0, 192, -1)                 # (no in-workgroup K split at all)
    try:
        assert not G._kw2_tiling(64 * 81, 64, 2)
    finally:
        rlx.gemm_tuning(192, 192, -1)
    assert G._kw2_tiling(64 * 81, 64, 2)
