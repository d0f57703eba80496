"""K7/K10 bootstrapped targets against the reference agents' own learn_from_batch arithmetic."""
import numpy as np
import pytest

from oracle import targets as T
from tests.util import dev_tensor, status_tensor


@pytest.mark.parametrize("name", ["dqn", "ddqn"])
def test_oracle_dqn_targets_match_reference(golden, name):
    g = golden("targets")
    tt, err = T.dqn_targets(g[name + "_q_next_t"], g[name + "_q_onl"], g[name + "_actions"],
                            g[name + "_rewards"], g[name + "_go"], 0.99,
                            q_next_online=g[name + "_q_next_o"] if name == "ddqn" else None)
    assert np.array_equal(tt, g[name + "_targets"])
    assert np.array_equal(err, g[name + "_errors"])


def test_oracle_ac_targets_match_reference(golden):
    g = golden("targets")
    t = T.ac_td_targets(g["ddpg_rewards"], g["ddpg_go"], g["ddpg_q_next"], 0.99, False, (-3.0, 3.0))
    assert np.array_equal(t, g["ddpg_targets"])
    t = T.ac_td_targets(g["td3_rewards"], g["td3_go"], g["td3_q_next"], 0.99)
    assert np.array_equal(t, g["td3_targets"])
    sm = T.td3_smooth_actions(g["td3_next_actions"], g["td3_noise"], 0.5, -0.8, 0.8)
    assert np.array_equal(sm, g["td3_smoothed"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dqn", "ddqn"])
def test_hip_dqn_targets_match_reference(golden, rlx, dev, name):
    import torch
    g = golden("targets")
    B, A = g[name + "_q_onl"].shape
    tt = dev_tensor(g[name + "_q_onl"], dev)
    err = torch.empty(B, dtype=torch.float64, device=dev)
    st = status_tensor(dev)
    sel = dev_tensor(g[name + "_q_next_o"], dev) if name == "ddqn" else None
    rlx.dqn_targets(dev_tensor(g[name + "_q_next_t"], dev), sel, tt,
                    dev_tensor(g[name + "_actions"], dev, np.int32), dev_tensor(g[name + "_rewards"], dev),
                    dev_tensor(g[name + "_go"], dev, np.uint8), 0.99, B, A, err, st, 0)
    assert int(st.item()) == 0
    assert np.array_equal(tt.cpu().numpy(), g[name + "_targets"])      # fp32, bit-exact
    assert np.array_equal(err.cpu().numpy(), g[name + "_errors"])      # fp64, bit-exact


@pytest.mark.gpu
def test_hip_dqn_targets_flags_bad_action(rlx, dev):
    import torch
    q = torch.zeros(2, 3, dtype=torch.float32, device=dev)
    st = status_tensor(dev)
    rlx.dqn_targets(q, None, q.clone(), dev_tensor([0, 3], dev, np.int32), dev_tensor([0., 0.], dev, np.float32),
                    dev_tensor([0, 0], dev, np.uint8), 0.99, 2, 3, None, st, 0)
    assert int(st.item()) == 1


@pytest.mark.gpu
def test_hip_ac_targets_match_reference(golden, rlx, dev):
    import torch
    g = golden("targets")
    for name, clip in (("ddpg", (1, -3.0, 3.0)), ("td3", (0, 0.0, 0.0))):
        B = len(g[name + "_rewards"])
        out = torch.empty(B, dtype=torch.float32, device=dev)
        rlx.ac_td_targets(dev_tensor(g[name + "_rewards"], dev), dev_tensor(g[name + "_go"], dev, np.uint8),
                          dev_tensor(g[name + "_q_next"], dev), 1, 0.99, 0, clip[0], clip[1], clip[2], B, out, 0)
        assert np.array_equal(out.cpu().numpy(), g[name + "_targets"][:, 0].astype(np.float32))
    na = g["td3_next_actions"]
    B, A = na.shape
    out = torch.empty(B, A, dtype=torch.float32, device=dev)
    lo = dev_tensor(np.full(A, -0.8), dev, np.float32)
    hi = dev_tensor(np.full(A, 0.8), dev, np.float32)
    rlx.td3_smooth_actions(dev_tensor(na, dev), dev_tensor(g["td3_noise"], dev), 0.5, lo, hi, B, A, out, 0)
    ref = np.clip(na + g["td3_noise"].clip(-0.5, 0.5), np.float32(-0.8), np.float32(0.8))
    assert np.array_equal(out.cpu().numpy(), ref.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dqn", "ddqn"])
@pytest.mark.parametrize("huber", [0, 1])
def test_hip_dqn_head_loss_matches_reference_targets_and_oracle_loss(golden, rlx, dev, name, huber):
    """The fused launch (targets + |TD error| + QHead loss + gradient): TD targets / errors bit-exact
    against the fixtures produced by the reference's DQNAgent / DDQNAgent.learn_from_batch, loss and
    gradient against the oracle's head loss on those targets (importance weights in fp64)."""
    import torch
    from oracle.losses import regression_head_loss
    g = golden("targets")
    q = g[name + "_q_onl"]
    B, A = q.shape
    rng = np.random.RandomState(B)
    w = rng.rand(B) + 0.1
    dq = torch.empty(B, A, device=dev)
    tt = torch.empty(B, A, device=dev)
    err = torch.empty(B, dtype=torch.float64, device=dev)
    loss = torch.zeros(1, device=dev)
    st = status_tensor(dev)
    sel = dev_tensor(g[name + "_q_next_o"], dev) if name == "ddqn" else None
    rlx.dqn_head_loss(dev_tensor(q, dev), A, dev_tensor(g[name + "_q_next_t"], dev), sel, A,
                      dev_tensor(g[name + "_actions"], dev, np.int32), dev_tensor(g[name + "_rewards"], dev),
                      dev_tensor(g[name + "_go"], dev, np.uint8), dev_tensor(w, dev), 0.99, B, A, huber, 1.0, dq, A,
                      err, tt, A, loss, st, 0)
    assert int(st.item()) == 0
    assert np.array_equal(tt.cpu().numpy(), g[name + "_targets"])      # fp32, bit-exact
    assert np.array_equal(err.cpu().numpy(), g[name + "_errors"])      # fp64, bit-exact
    ref_loss, ref_dq = regression_head_loss(q, g[name + "_targets"], w.astype(np.float32), "huber" if huber else "mse")
    np.testing.assert_allclose(float(loss.item()), ref_loss, rtol=1e-6)
    np.testing.assert_allclose(dq.cpu().numpy(), ref_dq, rtol=1e-6, atol=1e-9)
