"""rlx_dqn_head_loss_backward (coach_amd/csrc/targets.hip): DQNAgent.learn_from_batch's TD targets, |TD errors|, QHead loss
and dQ (agents/dqn_agent.py:92-113, heads/q_head.py, head.py:172-181) AND the Q head's backward pass as one launch, against
the two launches it replaces (rlx_dqn_head_loss + rlx_dense_small_backward): bit-identical gradients, loss, TD errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("image,B,A,huber,weighted", [(True, 32, 4, True, True), (True, 32, 6, False, False),
                                                      (False, 20, 3, True, True)])
def test_one_launch_equals_two(rlx, dev, image, B, A, huber, weighted):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn.networks import DQNNet
    shape = (84, 84, 4) if image else (17,)
    rng = np.random.RandomState(B + A)
    if image:
        both = torch.from_numpy(rng.randint(0, 256, size=(2, B) + shape).astype(np.uint8)).to(dev)
    else:
        both = torch.from_numpy(rng.randn(2, B, *shape).astype(np.float32)).to(dev)
    s, s2 = both[0], both[1]
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    rew = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    done = torch.from_numpy((rng.rand(B) < 0.3).astype(np.uint8)).to(dev)
    w = torch.from_numpy(rng.rand(B) + 0.1).to(dev) if weighted else None
    res = {}
    saved, saved_mlp = DQNNet.HEAD_LOSS_BACKWARD_ONE_LAUNCH, DQNNet.FUSED_MLP
    DQNNet.FUSED_MLP = False                    # (the vector network: the general path, not the one-launch MLP update)
    try:
        for one in (False, True):
            DQNNet.HEAD_LOSS_BACKWARD_ONE_LAUNCH = one
            net = DQNNet(dev, shape, A, seed=3, replace_mse_with_huber_loss=huber)
            td = torch.zeros(B, dtype=torch.float64, device=dev)
            with _rlx.KernelTimer(256) as timer:
                net.learn_from_batch(s, s2, B, acts, rew, done, 0.99, importance_weights=w, td_errors=td,
                                     states_pair=both if image else None)
            net.check_status()
            res[one] = (net.params.grads.clone(), net.params.weights.clone(), net.loss.clone(), td.clone(),
                        [n for n, _ in timer.records])
    finally:
        DQNNet.HEAD_LOSS_BACKWARD_ONE_LAUNCH, DQNNet.FUSED_MLP = saved, saved_mlp
    two, one = res[False], res[True]
    assert any("dqn_head_loss_kernel" in n for n in two[4]) and any("dense_small_bwd" in n for n in two[4]), two[4]
    assert sum("dqn_head_loss_bwd_kernel" in n for n in one[4]) == 1, one[4]
    assert not any("dqn_head_loss_kernel" in n or "dense_small_bwd" in n for n in one[4]), one[4]
    assert len(one[4]) == len(two[4]) - 1
    for a, b in zip(two[:4], one[:4]):
        assert torch.equal(a, b)
    assert float(two[0].abs().max()) > 0 and float(two[3].abs().max()) > 0
