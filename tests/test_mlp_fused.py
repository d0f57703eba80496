"""rlx_mlp_dqn_update (csrc/mlp_fused.hip): the ONE-launch DQN update of an MLP Q network against (a) the
layer-by-layer device path it replaces and (b) the numpy oracle of DQNAgent.learn_from_batch
(oracle.agents.DQNOracle, pinned to the reference's own learn_from_batch by tests/golden/updates.npz /
loop.npz).  Same weights, same batches: losses, |TD errors|, gradient norms and the weights after several
updates must agree to fp32 accumulation noise (the two device paths sum in different orders)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _net(dev, dims, A, huber, fused):
    from coach_amd.nn.networks import DQNNet
    d0, h1, h2 = dims
    net = DQNNet(dev, (d0,), A, embedder=[h1], middleware=[h2], learning_rate=1e-3,
                 replace_mse_with_huber_loss=huber, seed=4)
    if not fused:
        net._fused = None
    else:
        assert net._fused is not None
    return net


@pytest.mark.parametrize("dims,A,B,huber,ddqn,per", [
    ((4, 256, 512, ), 2, 32, False, False, False),       # BASELINE C1 (CartPole_DQN preset network)
    ((4, 256, 512, ), 2, 32, True, True, True),
    ((8, 96, 96, ), 5, 20, True, False, True),            # ragged batch, 3 workgroups
    ((15, 128, 64, ), 3, 7, False, True, False),          # odd obs dim, 2 workgroups
])
def test_fused_update_matches_layerwise_and_oracle(dev, dims, A, B, huber, ddqn, per):
    import torch
    from oracle.agents import DQNOracle
    fused, plain = _net(dev, dims, A, huber, True), _net(dev, dims, A, huber, False)
    assert torch.equal(fused.params.weights, plain.params.weights)
    orc = DQNOracle(plain.params.named_arrays(), (dims[0],), A, "relu", 1e-3, 0.9, 0.99, 1e-4, huber)
    rng = np.random.RandomState(B + A)
    td = {k: torch.zeros(B, dtype=torch.float64, device=dev) for k in "fp"}
    for it in range(4):
        s = rng.randn(B, dims[0]).astype(np.float32)
        s2 = rng.randn(B, dims[0]).astype(np.float32)
        a = rng.randint(0, A, B).astype(np.int32)
        r = rng.randn(B).astype(np.float32)
        done = (rng.rand(B) < 0.2)
        w = (rng.rand(B) + 0.1) if per else None
        dv = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dev).to(dt)
        args = (dv(s, torch.float32), dv(s2, torch.float32), B, dv(a, torch.int32), dv(r, torch.float32),
                dv(done.astype(np.uint8), torch.uint8), 0.97)
        res = {}
        for k, net in (("f", fused), ("p", plain)):
            loss = net.learn_from_batch(*args, importance_weights=None if w is None else dv(w, torch.float64),
                                        td_errors=td[k], double_dqn=ddqn)
            net.check_status()
            res[k] = (float(loss.item()), float(net.norm.item()))
        ref = orc.learn_from_batch(s, s2, a, r, done, 0.97, None if w is None else w.astype(np.float32), ddqn)
        np.testing.assert_allclose(res["f"][0], res["p"][0], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(res["f"][1], res["p"][1], rtol=2e-4)
        np.testing.assert_allclose(td["f"].cpu().numpy(), td["p"].cpu().numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(res["f"][0], ref["loss"], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(td["f"].cpu().numpy(), ref["td_errors"], rtol=2e-3, atol=5e-6)
        if it == 1:
            fused.update_target(1.0); plain.update_target(1.0); orc.update_target(1.0)
    wf, wp, wo = fused.params.named_arrays(), plain.params.named_arrays(), orc.weights()
    for name in wf:
        np.testing.assert_allclose(wf[name][0], wp[name][0], rtol=0, atol=2e-5, err_msg=name)
        np.testing.assert_allclose(wf[name][0], wo[name][0], rtol=0, atol=5e-5, err_msg=name)
    assert torch.allclose(fused.adam.state, plain.adam.state)
    assert int(fused._fused["sync"].abs().sum().item()) == 0          # barrier words re-armed


def test_fused_update_is_reproducible(dev):
    """fixed summation orders: two runs from the same state give bit-identical weights."""
    import torch
    outs = []
    for _ in range(2):
        net = _net(dev, (4, 256, 512), 2, False, True)
        rng = np.random.RandomState(0)
        td = torch.zeros(32, dtype=torch.float64, device=dev)
        for it in range(3):
            t = lambda x, dt: torch.from_numpy(x).to(dev).to(dt)
            net.learn_from_batch(t(rng.randn(32, 4).astype(np.float32), torch.float32),
                                 t(rng.randn(32, 4).astype(np.float32), torch.float32), 32,
                                 t(rng.randint(0, 2, 32).astype(np.int32), torch.int32),
                                 t(rng.randn(32).astype(np.float32), torch.float32),
                                 t((rng.rand(32) < 0.1).astype(np.uint8), torch.uint8), 0.99, td_errors=td)
        outs.append((net.params.weights.clone(), td.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
