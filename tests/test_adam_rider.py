"""Adam rider (rlx_adam_rider_arm / _flush, rlx_adam_tf1_step_parts; coach_amd/csrc/adam_rider.hpp): the Adam step of a
parameter range carried by workgroups of a dW + dX pair launch instead of the update's closing Adam launch.

What must hold (the reference applies every gradient in one session.run after the backward pass,
architectures/tensorflow_components/architecture.py:469-521 — per parameter the arithmetic is elementwise):
  * weights, m, v after an update are BIT-IDENTICAL to the single Adam launch, whichever kernel stepped a parameter
    (the rider's arithmetic is optim.hip's under `fp contract(off)`, in a translation unit compiled with contraction on);
  * the beta powers advance once;
  * tf.global_norm covers the whole buffer (another summation order: relative 1e-6);
  * a rider nobody carried is launched by the flush — never lost, never applied twice."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _buffers(dev, n, seed):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda scale=1.0: (torch.randn(n, generator=g, dtype=torch.float32) * scale).to(dev)
    return dict(w=mk(), g=mk(0.01), m=mk(0.001), v=(mk(0.001) ** 2))


@pytest.mark.parametrize("n,offset,launches", [(40000, 8000, 1), (3382791 // 4 * 4, 174208, 1), (100000, 4, 2),
                                               (4096, 1024, 1)])
def test_flushed_rider_plus_partial_step_equals_one_adam_launch(rlx, dev, n, offset, launches):
    import torch
    from coach_amd import _rlx
    s = _rlx.current_stream()
    ref, tst = _buffers(dev, n, 1), _buffers(dev, n, 1)
    lr, b1, b2, eps, scale = 2.5e-4, 0.9, 0.99, 1e-4, 0.5
    out = {}
    for name, b in (("ref", ref), ("tst", tst)):
        state = torch.empty(2, dtype=torch.float32, device=dev)
        rlx.adam_init(torch.empty(4, device=dev), torch.empty(4, device=dev), 4, state, b1, b2, s)
        ticket = torch.zeros(_rlx.ADAM_TICKET_WORDS, dtype=torch.int32, device=dev)
        ws = torch.zeros(8192, dtype=torch.float32, device=dev)
        norm = torch.zeros(1, dtype=torch.float32, device=dev)
        for _ in range(3):                                     # three steps: the beta powers matter from the second on
            if name == "ref":
                rlx.adam_tf1_step(b["w"], b["g"], b["m"], b["v"], n, lr, b1, b2, eps, state, scale, norm, ws, ws.numel(),
                                  None, None, 0, None, 0.0, ticket, s)
            else:
                blocks = 64
                own = ctypes.c_int()
                rlx.adam_step_blocks(offset, ws.numel(), blocks, ctypes.byref(own))
                rlx.adam_rider_arm(b["w"][offset:], b["g"][offset:], b["m"][offset:], b["v"][offset:], n - offset, lr, b1,
                                   b2, eps, state, scale, ws[own.value:], blocks, launches)
                with pytest.raises(Exception, match="already pending"):
                    rlx.adam_rider_arm(b["w"][offset:], b["g"][offset:], b["m"][offset:], b["v"][offset:], n - offset, lr,
                                       b1, b2, eps, state, scale, ws[own.value:], blocks, launches)
                pending = ctypes.c_int(-1)
                rlx.adam_rider_flush(s, ctypes.byref(pending))
                assert pending.value == 1
                rlx.adam_rider_flush(s, ctypes.byref(pending))              # nothing left: a no-op
                assert pending.value == 0
                rlx.adam_tf1_step_parts(b["w"], b["g"], b["m"], b["v"], offset, lr, b1, b2, eps, state, scale, norm, ws,
                                        ws.numel(), None, None, 0, None, 0.0, ticket, blocks, s)
        out[name] = (state.cpu().numpy(), float(norm.item()))
    for k in ("w", "m", "v"):
        assert torch.equal(ref[k], tst[k]), k
    np.testing.assert_array_equal(out["ref"][0], out["tst"][0])
    np.testing.assert_allclose(out["tst"][1], out["ref"][1], rtol=2e-6)
    np.testing.assert_allclose(out["ref"][1], float(torch.linalg.vector_norm(ref["g"].double()).item()), rtol=1e-5)


@pytest.mark.parametrize("shape,B,A", [((84, 84, 4), 64, 6), ((44, 44, 4), 16, 4)])
def test_update_with_the_rider_on_the_convolution_backward_equals_the_plain_update(rlx, dev, shape, B, A):
    """ClippedPPONet.forward_backward(adam_rider=1.0) + finish_update against the same network with the rider off: three
    updates, every weight / Adam slot bit for bit; at the C2 shape the rider must actually have TRAVELLED on a pair
    launch (no stand-alone rider kernel in the trace, an Adam launch over the convolution parameters only)."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn.networks import ClippedPPONet
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    res = {}
    for rider in (False, True):
        np.random.seed(1)
        net = ClippedPPONet(dev, shape, A, seed=2)
        net.ADAM_RIDER = rider
        net.update_target(1.0)
        old = net.policy_probs(obs, B, use_target=True, tag="old")
        acc = torch.zeros(8, dtype=torch.float32, device=dev)
        names = []
        for it in range(3):
            with _rlx.KernelTimer(256) as timer:
                net.forward_backward(obs, B, acts, adv, vt, old, adam_rider=1.0)
                net.finish_update(1.0, signal_acc=acc)
            names = [n for n, _ in timer.records]
        net.check_status()
        res[rider] = dict(w=net.params.weights.clone(), m=net.adam.m.clone(), v=net.adam.v.clone(),
                          state=net.adam.state.clone(), norm=float(net.norm.item()), acc=acc.cpu().numpy(), names=names)
    for k in ("w", "m", "v", "state"):
        assert torch.equal(res[False][k], res[True][k]), k
    np.testing.assert_allclose(res[True]["norm"], res[False]["norm"], rtol=2e-6)
    np.testing.assert_allclose(res[True]["acc"], res[False]["acc"], rtol=2e-6)
    assert not any("adam_rider_kernel" in n for n in res[False]["names"])
    pair = [n for n in res[True]["names"] if "pair_kernel" in n]
    if shape[0] == 84:
        assert pair, res[True]["names"]
        assert not any("adam_rider_kernel" in n for n in res[True]["names"]), res[True]["names"]
    # the same number of launches either way, unless the rider had to go out on its own
    extra = sum("adam_rider_kernel" in n for n in res[True]["names"])
    assert len(res[True]["names"]) == len(res[False]["names"]) + extra
