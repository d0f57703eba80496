"""rlx_ppo_fc_rows + rlx_splitk_reduce_jobs_ppo_tail / rlx_ppo_heads_tail (csrc/gemm.hip): the row-local part of the discrete
Clipped-PPO heads (forward, loss terms, dV / dlogits, dz of the last dense layer) inside that layer's K-split reduction and
the all-rows part (heads' dW / db, loss scalars) as extra workgroups of the backward pass's deferred-reduction launch —
against the three-launch path it replaces (rlx_gemm with row_heads + rlx_ppo_heads_loss_backward) on identical inputs.
Same per-row arithmetic, same chains, same reduction trees: BIT equality of every scalar, gradient and side output.  The
oracle comparisons at the C2 size run through this path by default (tests/test_ppo_full_size.py,
tests/test_ppo_long_episodes.py, tests/test_ppo_agent.py)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _inputs(net, dev, B, A, shape, bad_action=False):
    import torch
    rng = np.random.RandomState(B)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    a = rng.randint(0, A, size=B).astype(np.int32)
    if bad_action:
        a[B // 2] = A + 3
    actions = torch.from_numpy(a).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    net.update_target(1.0)
    # an old policy that differs from the current one (ratios off 1, some clipped)
    net.params.weights.add_(torch.from_numpy((rng.randn(net.params.size) * 2e-3).astype(np.float32)).to(dev))
    old = net.policy_probs(obs, B, use_target=True, tag="old").clone()
    return obs, actions, adv, vt, old


def _run(net, row_local, obs, B, actions, adv, vt, old, split=False):
    import torch
    net.HEADS_ROW_LOCAL = row_local          # (instance attribute)
    net.params.grads.zero_()
    net.scalars.zero_()
    net.status.zero_()
    ratio = torch.zeros(B, dtype=torch.float32, device=obs.device)
    clipped = torch.zeros(B, dtype=torch.float32, device=obs.device)
    net.forward_backward(obs, B, actions, adv, vt, old, 1.0, ratio, clipped, stop_after_dense=split)
    if split:
        net.backward_rest()
    torch.cuda.synchronize()
    ctx = net.ctx
    fc = net.torso.layers[-1]
    A = net.A
    out = {"scalars": net.scalars[:5].clone(), "grads": net.params.grads.clone(), "ratio": ratio, "clipped": clipped,
           "h": ctx.buffer(fc.name, (2, B, fc.N), tag="train").clone(),
           "dz": ctx.buffer(fc.name + ":grad", (2, B, fc.N), tag="train").clone(),
           "v": ctx.buffer("main/v_head/dense", (1, B, 1), tag="train").clone(),
           "logits": ctx.buffer("main/ppo_head/policy_fc", (1, B, A), tag="train").clone(),
           "dv": ctx.buffer("main/v_head/dense:grad", (1, B, 1), tag="train").clone(),
           "dlogits": ctx.buffer("main/ppo_head/policy_fc:grad", (1, B, A), tag="train").clone(),
           "status": int(net.status.item())}
    net.status.zero_()
    return out


def _same(a, b):
    import torch
    for k in a:
        if k == "status":
            assert a[k] == b[k], k
        else:
            assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(64, 6), (37, 4), (8, 16), (200, 3)])
def test_row_local_heads_equal_the_three_launch_path_bit_for_bit(dev, B, A):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn.networks import ClippedPPONet
    shape = (84, 84, 4)
    net = ClippedPPONet(dev, shape, A, seed=5)
    obs, actions, adv, vt, old = _inputs(net, dev, B, A, shape)
    before = _rlx.CALL_COUNT
    ref = _run(net, False, obs, B, actions, adv, vt, old)
    calls_ref = _rlx.CALL_COUNT - before
    before = _rlx.CALL_COUNT
    new = _run(net, True, obs, B, actions, adv, vt, old)
    calls_new = _rlx.CALL_COUNT - before
    assert float(ref["scalars"].abs().max()) > 0 and float(ref["grads"].abs().max()) > 0
    assert float(ref["dz"].abs().max()) > 0 and ref["status"] == 0
    _same(new, ref)
    if B <= 75:                              # (the heads' launch is gone; beyond the fused convolutions' batch range the
        assert calls_new < calls_ref, (calls_new, calls_ref)    #  support query evens the count of LIBRARY calls out)
    # twice in a row, and with the heads' all-rows part as a launch of its own in front of the rest of the backward pass
    _same(_run(net, True, obs, B, actions, adv, vt, old), ref)
    _same(_run(net, True, obs, B, actions, adv, vt, old, split=True), ref)
    _same(_run(net, False, obs, B, actions, adv, vt, old, split=True), ref)
    assert net.ctx.ppo_tail is None


@pytest.mark.gpu
def test_row_local_heads_report_an_invalid_action_like_the_three_launch_path(dev):
    from coach_amd.nn.networks import ClippedPPONet
    B, A, shape = 64, 6, (84, 84, 4)
    net = ClippedPPONet(dev, shape, A, seed=3)
    obs, actions, adv, vt, old = _inputs(net, dev, B, A, shape, bad_action=True)
    ref = _run(net, False, obs, B, actions, adv, vt, old)
    new = _run(net, True, obs, B, actions, adv, vt, old)
    assert ref["status"] == 1 and new["status"] == 1
    for k in ("scalars", "grads", "dz", "dlogits", "dv"):
        import torch
        assert torch.equal(new[k], ref[k]), k


@pytest.mark.gpu
def test_row_local_heads_in_a_captured_graph(dev):
    """the whole minibatch update (forward / backward / Adam) captured and replayed, with the row-local heads and with the
    three-launch path: the same weights and scalars after the same number of replays, bit for bit (the tail is a node —
    part of the deferred-reduction launch — of the captured graph)."""
    import torch
    from coach_amd.agents.vector_agent import capture
    from coach_amd.nn.networks import ClippedPPONet
    B, A, shape = 64, 6, (84, 84, 4)
    res = {}
    for row_local in (True, False):
        np.random.seed(11)                   # (the value head's normalized-columns init draws from the global stream)
        net = ClippedPPONet(dev, shape, A, seed=7)
        net.HEADS_ROW_LOCAL = row_local
        obs, actions, adv, vt, old = _inputs(net, dev, B, A, shape)

        def step():
            net.forward_backward(obs, B, actions, adv, vt, old)
            net.finish_update(1.0)
        step()                               # warm: every buffer exists
        torch.cuda.synchronize()
        g = capture(step)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert net.ctx.ppo_tail is None
        res[row_local] = (net.params.weights.clone(), net.scalars.clone())
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert float(res[True][1][:5].abs().max()) > 0


@pytest.mark.gpu
def test_vector_network_falls_back(dev):
    """a network whose last dense layer is not a K-split product (small MLP): rlx_ppo_fc_rows_supported says no, the
    layer-by-layer path runs, nothing is left pending."""
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    B, A = 32, 3
    net = ClippedPPONet(dev, (8,), A, seed=1)
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randn(B, 8).astype(np.float32)).to(dev)
    actions = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    net.update_target(1.0)
    old = net.policy_probs(obs, B, use_target=True, tag="old").clone()
    out = []
    for flag in (True, False):
        net.HEADS_ROW_LOCAL = flag
        net.params.grads.zero_()
        net.forward_backward(obs, B, actions, adv, vt, old)
        torch.cuda.synchronize()
        out.append((net.params.grads.clone(), net.scalars.clone()))
        assert net.ctx.ppo_tail is None
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
