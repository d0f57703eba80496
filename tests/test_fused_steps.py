"""One-launch forms of per-step sequences against the launches they replace (same inputs, same device):

* rlx_mlp_q_act      = layer-by-layer Q(s) + rlx_egreedy                       (values to fp32 noise, same choices)
* rlx_observe_step   = rlx_reward_filter -> rlx_episode_stats_step -> rlx_copy_columns -> rlx_select_rows   (bit-exact)
* rlx_ac_critic_losses = rlx_min_pair -> rlx_ac_td_targets -> n x rlx_regression_loss                       (bit-exact)
* rlx_ac_merge_inputs  = rlx_td3_smooth_actions + the strided copies of CriticNet.forward_pair              (bit-exact)
* rlx_sac_min_targets  = rlx_min_pair (+ gradients) + rlx_sac_value_targets                                 (bit-exact)
The replaced entry points are themselves pinned to the numpy oracle elsewhere (test_targets / test_replay / ...)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(x, dev, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t if dt is None else t.to(dt)


@pytest.mark.parametrize("dims,A,n_env", [((4, 256, 512), 2, 1), ((8, 96, 160), 5, 3), ((15, 64, 64), 16, 8)])
def test_q_act_matches_layerwise_forward_and_egreedy(dev, dims, A, n_env):
    import torch
    from coach_amd import _rlx
    from coach_amd.nn.networks import DQNNet
    d0, h1, h2 = dims
    net = DQNNet(dev, (d0,), A, embedder=[h1], middleware=[h2], learning_rate=1e-3, seed=3)
    assert net.can_act_fused(n_env)
    lib = _rlx.lib()
    rng = np.random.RandomState(n_env + A)
    for it in range(6):
        s = _t(rng.randn(n_env, d0).astype(np.float32), dev)
        u = _t(rng.rand(n_env), dev)
        ra = _t(rng.randint(0, A, n_env).astype(np.int32), dev)
        tie = _t(rng.rand(n_env, A), dev)
        eps = 0.3
        q_ref = net.q_values(s, n_env, tag="ref").data.view(n_env, A).clone()
        a_ref = torch.zeros(n_env, dtype=torch.int32, device=dev)
        q_f = torch.zeros(n_env, A, dtype=torch.float32, device=dev)
        a_f = torch.full((n_env,), -1, dtype=torch.int32, device=dev)
        net.q_act(s, n_env, u, ra, tie, eps, q_f, a_f)
        np.testing.assert_allclose(q_f.cpu().numpy(), q_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
        # the choice restates rlx_egreedy: same values in -> same actions out
        lib.egreedy(q_f, A, u, ra, tie, eps, n_env, A, a_ref, _rlx.current_stream())
        assert torch.equal(a_f, a_ref)
    # values only (target network, no choice)
    net.update_target(1.0)
    q_t = torch.zeros(n_env, A, dtype=torch.float32, device=dev)
    net.q_act(s, n_env, None, None, None, 0.0, q_t, None, use_target=True)
    np.testing.assert_allclose(q_t.cpu().numpy(), q_f.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("n_env,D,act_shape", [(1, 4, ()), (5, 17, (6,)), (64, 3, ())])
def test_observe_step_equals_the_four_launches(dev, n_env, D, act_shape):
    import torch
    from coach_amd import _rlx
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    rng = np.random.RandomState(n_env)
    rows = 4 * n_env + 3
    adt = torch.int32 if act_shape == () else torch.float32

    def fresh():
        st = dict(
            filtered=torch.zeros(n_env, dtype=torch.float32, device=dev),
            ep_return=torch.zeros(n_env, dtype=torch.float64, device=dev),
            ep_len=torch.zeros(n_env, dtype=torch.int32, device=dev),
            acc=torch.zeros(8, dtype=torch.float64, device=dev),
            last_return=torch.zeros(n_env, dtype=torch.float64, device=dev),
            cur=_t(np.arange(n_env * D, dtype=np.float32).reshape(n_env, D), dev),
            m_action=torch.zeros((rows,) + act_shape, dtype=adt, device=dev),
            m_reward=torch.zeros(rows, dtype=torch.float32, device=dev),
            m_go=torch.zeros(rows, dtype=torch.uint8, device=dev),
            m_obs=torch.zeros(rows, D, dtype=torch.float32, device=dev),
            m_next=torch.zeros(rows, D, dtype=torch.float32, device=dev),
            status=torch.zeros(1, dtype=torch.int32, device=dev))
        lib.episode_stats_init(st["ep_return"], st["ep_len"], n_env, st["acc"], s_)
        return st
    A, Bst = fresh(), fresh()
    for step in range(7):
        reward = _t((rng.randn(n_env) * 3).astype(np.float32), dev)
        go = _t((rng.rand(n_env) < 0.3).astype(np.uint8), dev)
        stored_go = _t(np.zeros(n_env, dtype=np.uint8), dev) if step % 2 else go
        nxt = _t(rng.randn(n_env, D).astype(np.float32), dev)
        rst = _t(rng.randn(n_env, D).astype(np.float32), dev)
        acts = _t(rng.randint(0, 5, (n_env,) + act_shape), dev, adt)
        dst = _t(((step * n_env + np.arange(n_env)) % rows).astype(np.int32), dev)
        # ---- the four launches
        lib.reward_filter(reward, A["filtered"], n_env, 0.5, 1, -1.0, 1.0, s_)
        lib.episode_stats_step(A["filtered"], go, A["ep_return"], A["ep_len"], n_env, A["acc"], A["last_return"], None, s_)
        pairs = [(acts, A["m_action"]), (A["filtered"], A["m_reward"]), (stored_go, A["m_go"]),
                 (A["cur"], A["m_obs"]), (nxt, A["m_next"])]
        lib.copy_columns(_rlx.make_columns(pairs), len(pairs), None, dst, 0, 0, n_env, rows, n_env, A["status"], s_)
        lib.select_rows(go, rst, nxt, A["cur"], n_env, D * 4, s_)
        # ---- one launch
        d = _rlx.ObserveDesc()
        p = lambda t: t.data_ptr()
        d.reward, d.filtered_reward, d.reward_rescale, d.has_clip, d.clip_low, d.clip_high = \
            p(reward), p(Bst["filtered"]), 0.5, 1, -1.0, 1.0
        d.game_over, d.stored_game_over = p(go), p(stored_go)
        d.ep_return, d.ep_len, d.acc, d.last_return, d.last_len = p(Bst["ep_return"]), p(Bst["ep_len"]), p(Bst["acc"]), \
            p(Bst["last_return"]), None
        d.actions, d.action_row_bytes = p(acts), acts.element_size() * acts[0].numel()
        d.cur_state, d.next_obs, d.reset_obs, d.obs_row_bytes = p(Bst["cur"]), p(nxt), p(rst), D * 4
        d.mem_action, d.mem_reward, d.mem_game_over, d.mem_obs, d.mem_next_obs = \
            p(Bst["m_action"]), p(Bst["m_reward"]), p(Bst["m_go"]), p(Bst["m_obs"]), p(Bst["m_next"])
        d.dst_rows, d.mem_rows, d.status, d.n_env = p(dst), rows, p(Bst["status"]), n_env
        lib.observe_step(ctypes.byref(d), s_)
        for k in A:
            assert torch.equal(A[k], Bst[k]), (step, k)
    assert int(A["status"].item()) == 0
    # an out-of-range destination row raises the status bit, as rlx_copy_columns does
    bad = _t(np.full(n_env, rows, dtype=np.int32), dev)
    d.dst_rows = bad.data_ptr()
    lib.observe_step(ctypes.byref(d), s_)
    assert int(Bst["status"].item()) & 1


@pytest.mark.parametrize("T,B,twin_next,weight", [(2, 100, True, 1.0), (1, 32, False, 1.0), (2, 256, False, 0.5)])
def test_ac_critic_losses_equals_the_separate_launches(dev, T, B, twin_next, weight):
    import torch
    from coach_amd import _rlx
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    rng = np.random.RandomState(B)
    q = _t(rng.randn(T, B).astype(np.float32), dev)
    qn1 = _t(rng.randn(B).astype(np.float32), dev)
    qn2 = _t(rng.randn(B).astype(np.float32), dev) if twin_next else None
    r = _t(rng.randn(B).astype(np.float32), dev)
    go = _t((rng.rand(B) < 0.2).astype(np.uint8), dev)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    qmin_a, y_a, dq_a, loss_a = z(B), z(B), z(T, B), z(T + 1)
    if twin_next:
        lib.min_pair(qn1, qn2, qmin_a, None, None, 0.0, B, s_)
    else:
        qmin_a.copy_(qn1)
    lib.ac_td_targets(r, go, qmin_a, 1, 0.99, 0, 1, -3.0, 3.0, B, y_a, s_)
    for t in range(T):
        lib.regression_loss(q[t], 1, y_a, 1, None, B, 1, 0, weight, 1.0, dq_a[t], 1, loss_a[t:t + 1], s_)
    qmin_b, y_b, dq_b, loss_b = z(B), z(B), z(T, B), z(T + 1)
    lib.ac_critic_losses(qn1, qn2, r, go, 0.99, 0, 1, -3.0, 3.0, q, T, B, weight, qmin_b, y_b, dq_b, loss_b, s_)
    assert torch.equal(y_a, y_b) and torch.equal(dq_a, dq_b) and torch.equal(qmin_a, qmin_b)
    assert torch.equal(loss_a[:T], loss_b[:T])
    total = loss_b[0] if T == 1 else loss_b[0] + loss_b[1]
    assert float(loss_b[T]) == float(total)


@pytest.mark.parametrize("smooth", [True, False])
def test_ac_merge_inputs_equals_smoothing_plus_copies(dev, smooth):
    import torch
    from coach_amd import _rlx
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    B, A, D = 100, 6, 17
    rng = np.random.RandomState(7)
    acts = _t(rng.uniform(-1, 1, (B, A)).astype(np.float32), dev)
    nacts = _t(rng.uniform(-1, 1, (B, A)).astype(np.float32), dev)
    obs = _t(rng.randn(B, D).astype(np.float32), dev)
    nobs = _t(rng.randn(B, D).astype(np.float32), dev)
    noise = _t(rng.normal(0, 0.2, (B, A)), dev)
    low, high = _t(np.full(A, -1, np.float32), dev), _t(np.full(A, 1, np.float32), dev)
    sm = torch.zeros(B, A, dtype=torch.float32, device=dev)
    if smooth:
        lib.td3_smooth_actions(nacts, noise, 0.5, low, high, B, A, sm, s_)
    else:
        sm.copy_(nacts)
    ref = torch.stack([torch.cat([acts, obs], 1), torch.cat([sm, nobs], 1)])
    out = torch.full((2, B, A + D), 9.0, dtype=torch.float32, device=dev)
    only = torch.full((B, A + D), 9.0, dtype=torch.float32, device=dev)
    lib.ac_merge_inputs(acts, obs, nacts, noise if smooth else None, 0.5, low, high, nobs, B, A, D, out, only, s_)
    assert torch.equal(out, ref)
    assert torch.equal(only[:, A:], obs) and bool((only[:, :A] == 9.0).all())


def test_sac_min_targets_equals_min_pair_and_value_targets(dev):
    import torch
    from coach_amd import _rlx
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    B = 256
    rng = np.random.RandomState(1)
    q1, q2 = _t(rng.randn(B).astype(np.float32), dev), _t(rng.randn(B).astype(np.float32), dev)
    q2[:7] = q1[:7]                                                 # ties go to q1
    lp = _t(rng.randn(B).astype(np.float32), dev)
    z = lambda: torch.zeros(B, dtype=torch.float32, device=dev)
    m_a, g1_a, g2_a, vt_a = z(), z(), z(), z()
    lib.min_pair(q1, q2, m_a, g1_a, g2_a, 1.0 / B, B, s_)
    lib.sac_value_targets(m_a, lp, B, vt_a, s_)
    m_b, g1_b, g2_b, vt_b = z(), z(), z(), z()
    lib.sac_min_targets(q1, q2, lp, 1.0 / B, B, m_b, vt_b, g1_b, g2_b, s_)
    for a, b in ((m_a, m_b), (g1_a, g1_b), (g2_a, g2_b), (vt_a, vt_b)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,with_norm,with_mix,norm_in_kernel", [
    (3_400_000, True, False, 2), (130_001, True, True, 2), (70_000, False, True, 2), (257, False, False, 2),
    (5_000_003, True, True, 2),        # beyond a workgroup's registers: the separate finish launch
    (3_400_000, True, False, 1), (5_000_003, True, True, 1), (130_001, True, True, 1),   # grid-stride one-launch form
    (130_001, True, True, 0), (3_400_000, True, False, 0)])                              # always two launches
def test_adam_step_one_launch_equals_adam_finish_and_mix(dev, n, with_norm, with_mix, norm_in_kernel):
    """rlx_adam_tf1_step (Adam + norm + beta-power advance by the last workgroup to finish, + the soft target update in
    the same pass) against rlx_adam_tf1(_norm) followed by rlx_mix_weights: weights, slots, beta powers, norm, signal
    sums and target bit for bit, over several steps (the ticket re-arms itself)."""
    import torch
    from coach_amd import _rlx
    lib, s_ = _rlx.lib(), _rlx.current_stream()
    lib.adam_norm_in_kernel(norm_in_kernel)
    rng = np.random.RandomState(n % 1000)
    w0 = rng.randn(n).astype(np.float32)
    t0 = rng.randn(n).astype(np.float32)

    def fresh():
        st = dict(w=_t(w0, dev), t=_t(t0, dev), m=torch.empty(n, dtype=torch.float32, device=dev),
                  v=torch.empty(n, dtype=torch.float32, device=dev), state=torch.empty(2, dtype=torch.float32, device=dev),
                  norm=torch.zeros(1, dtype=torch.float32, device=dev), acc=torch.zeros(3, dtype=torch.float32, device=dev))
        lib.adam_init(st["m"], st["v"], n, st["state"], 0.9, 0.99, s_)
        return st
    a, b = fresh(), fresh()
    ws_a = torch.empty(1 << 12, dtype=torch.float32, device=dev)
    ws_b = torch.empty(1 << 12, dtype=torch.float32, device=dev)
    ticket = torch.zeros(_rlx.ADAM_TICKET_WORDS, dtype=torch.int32, device=dev)
    for step in range(4):
        g = _t((rng.randn(n) * 0.1).astype(np.float32), dev)
        src = _t(rng.randn(3).astype(np.float32), dev)
        if with_norm:
            lib.adam_tf1_norm(a["w"], g, a["m"], a["v"], n, 1e-3, 0.9, 0.99, 1e-4, a["state"], 0.5, a["norm"], ws_a,
                              ws_a.numel(), src, a["acc"], 3, s_)
        else:
            lib.adam_tf1(a["w"], g, a["m"], a["v"], n, 1e-3, 0.9, 0.99, 1e-4, a["state"], 0.5, s_)
        if with_mix:
            lib.mix_weights(a["t"], a["w"], n, 0.005, s_)
        lib.adam_tf1_step(b["w"], g, b["m"], b["v"], n, 1e-3, 0.9, 0.99, 1e-4, b["state"], 0.5,
                          b["norm"] if with_norm else None, ws_b if with_norm else None, ws_b.numel() if with_norm else 0,
                          src if with_norm else None, b["acc"] if with_norm else None, 3 if with_norm else 0,
                          b["t"] if with_mix else None, 0.005, ticket, s_)
        for k in a:
            assert torch.equal(a[k], b[k]), (step, k)
        assert int(ticket.abs().sum().item()) == 0
    lib.adam_norm_in_kernel(2)
