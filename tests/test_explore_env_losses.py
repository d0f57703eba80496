"""Exploration policies, the synthetic vector env and the head-loss kernels."""
import numpy as np
import pytest

from oracle import explore as E
from oracle import losses as L
from oracle import nn as N
from oracle import synth_env as S
from tests.util import dev_tensor, status_tensor


# ------------------------------------------------------------------------------------ CPU
def test_oracle_exploration_matches_reference(golden):
    g = golden("explore")
    acts = [E.categorical_choice(p, u) for p, u in zip(g["cat_probs"], g["cat_u"])]
    assert acts == g["cat_actions"].tolist()
    acts = [E.egreedy_choice(q, u, ra, tie, eps) for q, u, ra, tie, eps in
            zip(g["eg_q"], g["eg_explore_u"], g["eg_rand_act"], g["eg_tie"], g["eg_eps"])]
    assert acts == g["eg_actions"].tolist()
    a = E.gaussian_action(g["an_mean"], g["an_std"], g["an_z"])
    assert np.array_equal(a, g["an_actions"])


def test_product_egreedy_host_draws_match_reference(golden):
    """coach_amd EGreedy.draw() — the HOST half of the device policy (the order in which the legacy
    np.random stream is consumed, the epsilon schedule stepping) — replayed against the reference's
    EGreedy.get_action run under the same seed (fixture `explore`): epsilon, the explore draw, the random
    action / tie-break randoms of every call are bit-identical, hence so are the actions (the device half,
    rlx_egreedy, is checked against the same fixture in the GPU test below)."""
    from coach_amd.core_types import RunPhase
    from coach_amd.exploration_policies.e_greedy import EGreedy
    from coach_amd.schedules import LinearSchedule
    g = golden("explore")
    A = g["eg_q"].shape[1]
    pol = EGreedy.__new__(EGreedy)                       # no device members: draw() is host-only
    pol.A, pol.n_env, pol.phase = A, 1, RunPhase.TRAIN
    pol.epsilon_schedule, pol.evaluation_epsilon = LinearSchedule(0.5, 0.1, 100), 0.05
    np.random.seed(4)
    pol.current_random_value = np.array([np.random.rand()])
    acts = []
    for i in range(len(g["eg_q"])):
        eps, u, ra, tie = pol.draw()
        eps = float(eps[0])                                # one epsilon per env (per get_action call)
        assert eps == g["eg_eps"][i] and u[0] == g["eg_explore_u"][i], i
        if u[0] < eps:
            assert ra[0] == g["eg_rand_act"][i], i
        else:
            assert np.array_equal(tie[0], g["eg_tie"][i]), i
        acts.append(E.egreedy_choice(g["eg_q"][i], u[0], ra[0], tie[0], eps))
    assert acts == g["eg_actions"].tolist()
    # an evaluation right after: phase TEST acts with evaluation_epsilon, the schedule stands still (step_epsilon,
    # e_greedy.py:112-117), the explore / tie-break draws go on from the same stream
    pol.phase = RunPhase.TEST
    acts = []
    for i in range(len(g["egt_q"])):
        eps, u, ra, tie = pol.draw()
        eps = float(eps[0])
        assert eps == 0.05 == g["egt_eps"][i] and u[0] == g["egt_explore_u"][i], i
        if u[0] < eps:
            assert ra[0] == g["egt_rand_act"][i], i
        else:
            assert np.array_equal(tie[0], g["egt_tie"][i]), i
        acts.append(E.egreedy_choice(g["egt_q"][i], u[0], ra[0], tie[0], eps))
    assert acts == g["egt_actions"].tolist()
    assert (g["egt_explore_u"] < 0.05).sum() > 0                      # the evaluation did explore a few times
    pol.phase = RunPhase.TRAIN
    assert pol.epsilon_schedule.current_value == float(g["egt_schedule_after"])      # untouched by the evaluation


def test_product_ou_process_host_noise_matches_reference(golden):
    """coach_amd OUProcess — the HOST half (the correlated noise state, np.random.randn consumption, restart at an
    episode end) — against the reference's OUProcess.get_action under the same seed (fixture `explore`): action =
    policy mean + noise, bit for bit in fp64 (the device half is rlx_gaussian_action with std 1, checked below)."""
    from coach_amd.core_types import RunPhase
    from coach_amd.exploration_policies.ou_process import OUProcess
    g = golden("explore")
    A = g["ou_mean"].shape[1]
    pol = OUProcess.__new__(OUProcess)                  # no device members: noise() is host-only
    pol.A, pol.n_env, pol.phase = A, 1, RunPhase.TRAIN
    pol.mu, pol.theta, pol.sigma, pol.dt = 0.0 * np.ones(A), 0.15, 0.2 * np.ones(A), 0.01
    pol.state = np.zeros((1, A))
    np.random.seed(8)
    for i, m in enumerate(g["ou_mean"]):
        if i == int(g["ou_reset_at"]):
            pol.reset([0])
        noise = pol.noise()[0]
        assert np.array_equal(m + noise, g["ou_actions"][i]), i


def test_product_egreedy_draws_for_several_envs_are_sequential_calls():
    """n_env envs = n_env sequential get_action calls of the reference policy object: every call sees the epsilon the
    previous call's step_epsilon left behind, and consumes the shared np.random stream in call order."""
    from coach_amd.core_types import RunPhase
    from coach_amd.exploration_policies.e_greedy import EGreedy
    from coach_amd.schedules import LinearSchedule
    A, n = 4, 3
    pol = EGreedy.__new__(EGreedy)
    pol.A, pol.n_env, pol.phase = A, n, RunPhase.TRAIN
    pol.epsilon_schedule, pol.evaluation_epsilon = LinearSchedule(0.9, 0.1, 10), 0.05
    np.random.seed(11)
    pol.current_random_value = np.array([np.random.rand() for _ in range(n)])
    first = pol.current_random_value.copy()
    np.random.seed(12)
    got = [pol.draw() for _ in range(4)]
    # by hand: one schedule, one stream, calls in env order
    sched = LinearSchedule(0.9, 0.1, 10)
    np.random.seed(12)
    cur = first.copy()
    for eps, u, ra, tie in got:
        for e in range(n):
            assert eps[e] == sched.current_value and u[e] == cur[e]
            if cur[e] < sched.current_value:
                assert ra[e] == np.random.choice(A)
            else:
                assert np.array_equal(tie[e], np.random.random(A))
            sched.step()
            cur[e] = np.random.rand()
    assert len({float(x) for x in got[0][0]}) == n          # the three calls of a step saw three different epsilons


def test_oracle_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    def run(c, k):
        r = S.philox4x32_10(*[np.array([x], dtype=np.uint64) for x in c], k[0], np.array([k[1]], dtype=np.uint64))
        return [int(x[0]) for x in r]
    assert run((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_oracle_synth_env_statistics():
    env = S.SynthVecEnv(0, 3, 84 * 84, 5, 7)
    first = env.reset()
    assert first.shape == (3, 7056) and first.dtype == np.uint8
    assert 120 < first.mean() < 135                       # uniform bytes
    dones = []
    for _ in range(11):
        nxt, rst, rew, done = env.step()
        dones.append(done.copy())
        assert set(np.unique(rew)) <= {-1.0, 0.0, 1.0}
    assert np.array(dones)[:, 0].tolist() == [False] * 4 + [True] + [False] * 4 + [True] + [False]
    v = S.SynthVecEnv(1, 2, 4096, 4, 7).reset()
    assert abs(v.std() - 1.0) < 0.05 and abs(v.mean()) < 0.05


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_exploration_matches_reference(golden, rlx, dev):
    import torch
    g = golden("explore")
    n, A = g["cat_probs"].shape
    acts = torch.empty(n, dtype=torch.int32, device=dev)
    rlx.categorical_sample(dev_tensor(g["cat_probs"], dev), A, dev_tensor(g["cat_u"], dev), n, A, acts, 0)
    assert acts.cpu().numpy().tolist() == g["cat_actions"].tolist()
    # e-greedy: epsilon changes per call in the trace -> one launch per distinct epsilon value
    q = g["eg_q"]
    n, A = q.shape
    out = np.zeros(n, dtype=np.int32)
    for i in range(n):
        a = torch.empty(1, dtype=torch.int32, device=dev)
        rlx.egreedy(dev_tensor(q[i:i + 1], dev), A, dev_tensor(g["eg_explore_u"][i:i + 1], dev),
                    dev_tensor(np.maximum(g["eg_rand_act"][i:i + 1], 0), dev, np.int32),
                    dev_tensor(g["eg_tie"][i:i + 1], dev), float(g["eg_eps"][i]), 1, A, a, 0)
        out[i] = a.item()
    assert out.tolist() == g["eg_actions"].tolist()
    mean = g["an_mean"]
    B, D = mean.shape
    o = torch.empty(B, D, dtype=torch.float32, device=dev)
    rlx.gaussian_action(dev_tensor(mean, dev), dev_tensor(g["an_std"], dev, np.float32), None,
                        dev_tensor(g["an_z"], dev), None, None, B, D, o, 0)
    np.testing.assert_allclose(o.cpu().numpy(), g["an_actions"].astype(np.float32), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_env,obs_elems,L", [(0, 5, 84 * 84, 4), (1, 7, 17, 3), (1, 3, 376, 1000)])
def test_hip_synth_env_bit_exact_with_oracle(rlx, dev, kind, n_env, obs_elems, L):
    import torch
    dt = torch.uint8 if kind == 0 else torch.float32
    obs = torch.empty(n_env, obs_elems, dtype=dt, device=dev)
    nxt, rst = torch.empty_like(obs), torch.zeros_like(obs)
    rew = torch.empty(n_env, dtype=torch.float32, device=dev)
    done = torch.empty(n_env, dtype=torch.uint8, device=dev)
    ep = torch.empty(n_env, dtype=torch.int32, device=dev)
    t = torch.empty(n_env, dtype=torch.int32, device=dev)
    o = S.SynthVecEnv(kind, n_env, obs_elems, L, 1234, env_id0=64)
    rlx.synth_env_reset(kind, obs, ep, t, n_env, obs_elems, 1234, 64, 0)
    assert np.array_equal(obs.cpu().numpy(), o.reset())
    for s in range(9):
        rlx.synth_env_step(kind, nxt, rst, rew, done, ep, t, n_env, obs_elems, L, 1234, 64, 0)
        on, orst, orew, odone = o.step()
        assert np.array_equal(nxt.cpu().numpy(), on)
        assert np.array_equal(rew.cpu().numpy(), orew)
        assert np.array_equal(done.cpu().numpy().astype(bool), odone)
        d = odone
        assert np.array_equal(rst.cpu().numpy()[d], orst[d])
        assert np.array_equal(ep.cpu().numpy(), o.ep) and np.array_equal(t.cpu().numpy(), o.t)


@pytest.mark.gpu
def test_hip_loss_kernels_vs_oracle_and_reference_kat(rlx, dev):
    import torch
    rng = np.random.RandomState(12)
    # reference KAT (mxnet test_ppo_head.py:363-376): surrogate = -0.142857153
    new = np.array([[0.9, 0.1], [0.2, 0.8], [0.4, 0.6]], dtype=np.float32)
    old = np.array([[0.7, 0.3], [0.2, 0.8], [0.4, 0.6]], dtype=np.float32)
    sc = torch.zeros(4, dtype=torch.float32, device=dev)
    st = status_tensor(dev)
    rlx.ppo_discrete_loss(dev_tensor(np.log(new), dev), 2, dev_tensor([0, 1, 0], dev, np.int32),
                          dev_tensor([-2, 2, 1], dev, np.float32), dev_tensor(old, dev), 2, 3, 2, 0.2, 0.0, 1.0,
                          None, 2, sc, None, None, st, None, 0)
    np.testing.assert_allclose(sc[0].item(), -0.142857153, rtol=2e-6)
    for B, A, beta in ((64, 6, 0.01), (1, 2, 0.0), (1000, 18, 0.05)):
        logits = rng.randn(B, A).astype(np.float32)
        oldp = N.softmax(logits + 0.5 * rng.randn(B, A).astype(np.float32))
        acts = rng.randint(0, A, size=B)
        adv = rng.randn(B).astype(np.float32)
        ref = L.ppo_discrete_loss(logits, acts, adv, oldp, 0.2, beta)
        dl = torch.empty(B, A, dtype=torch.float32, device=dev)
        ratio = torch.empty(B, dtype=torch.float32, device=dev)
        clipped = torch.empty(B, dtype=torch.float32, device=dev)
        # clip range as epsilon x a DEVICE rescaler (0.4 x 0.5 = 0.2 exactly in fp32)
        rlx.ppo_discrete_loss(dev_tensor(logits, dev), A, dev_tensor(acts, dev, np.int32), dev_tensor(adv, dev),
                              dev_tensor(oldp, dev), A, B, A, 0.4, beta, 1.0, dl, A, sc, ratio, clipped, st,
                              dev_tensor(np.array([0.5], dtype=np.float32), dev), 0)
        np.testing.assert_allclose(sc.cpu().numpy(), [ref["surrogate"], ref["entropy"], ref["kl"], ref["total"]],
                                   rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(dl.cpu().numpy(), ref["dlogits"], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(ratio.cpu().numpy(), ref["ratio"], rtol=1e-5)
        np.testing.assert_allclose(clipped.cpu().numpy(), ref["clipped"], rtol=1e-5)
        probs = torch.empty(B, A, dtype=torch.float32, device=dev)
        rlx.softmax(dev_tensor(logits, dev), A, B, A, probs, A, 0)
        np.testing.assert_allclose(probs.cpu().numpy(), N.softmax(logits), rtol=1e-5, atol=1e-8)
    for kind, name in ((0, "mse"), (1, "huber")):
        B, D = 33, 5
        out = (rng.randn(B, D) * 2).astype(np.float32)
        tgt = rng.randn(B, D).astype(np.float32)
        w = rng.rand(B).astype(np.float32)
        ref_l, ref_g = L.regression_head_loss(out, tgt, w, name, 0.5)
        g = torch.empty(B, D, dtype=torch.float32, device=dev)
        l = torch.zeros(1, dtype=torch.float32, device=dev)
        rlx.regression_loss(dev_tensor(out, dev), D, dev_tensor(tgt, dev), D, dev_tensor(w, dev), B, D, kind,
                            0.5, 1.0, g, D, l, 0)
        np.testing.assert_allclose(l.item(), ref_l, rtol=1e-5)
        np.testing.assert_allclose(g.cpu().numpy(), ref_g, rtol=1e-5, atol=1e-8)
    assert int(st.item()) == 0
    rlx.ppo_discrete_loss(dev_tensor(np.log(new), dev), 2, dev_tensor([0, 5, 0], dev, np.int32),
                          dev_tensor([-2, 2, 1], dev, np.float32), dev_tensor(old, dev), 2, 3, 2, 0.2, 0.0, 1.0,
                          None, 2, sc, None, None, st, None, 0)
    assert int(st.item()) == 1


def test_oracle_ppo_continuous_loss_matches_torch_autograd():
    """The MultivariateNormalDiag surrogate / entropy / KL and their gradients against torch
    distributions + autograd (an independent implementation of ppo_head.py:58-98,118-144)."""
    import torch
    from oracle.losses import ppo_continuous_loss
    rng = np.random.RandomState(3)
    B, A = 12, 4
    mean = rng.randn(B, A).astype(np.float32) * 0.5
    ls = (rng.randn(A) * 0.3).astype(np.float32)
    x = rng.randn(B, A).astype(np.float32)
    adv = rng.randn(B).astype(np.float32)
    om = mean + rng.randn(B, A).astype(np.float32) * 0.1
    os_ = np.exp(ls + rng.randn(A).astype(np.float32) * 0.05)[None].repeat(B, 0).astype(np.float32)
    r = ppo_continuous_loss(mean, ls, x, adv, om, os_, 0.2, 0.01)
    eps = float(np.finfo(np.float32).eps)
    tm = torch.tensor(mean, requires_grad=True)
    tl = torch.tensor(ls, requires_grad=True)
    new = torch.distributions.Normal(tm, torch.exp(tl) + eps)
    old = torch.distributions.Normal(torch.tensor(om), torch.tensor(os_) + eps)
    tx = torch.tensor(x)
    ratio = torch.exp(new.log_prob(tx).sum(1) - old.log_prob(tx).sum(1))
    ta = torch.tensor(adv)
    sur = -torch.minimum(ratio * ta, torch.clamp(ratio, 0.8, 1.2) * ta).mean()
    ent = new.entropy().sum(1).mean()
    kl = torch.distributions.kl_divergence(old, new).sum(1).mean()
    (sur - 0.01 * ent).backward()
    np.testing.assert_allclose(r["surrogate"], float(sur.detach()), rtol=2e-5)
    np.testing.assert_allclose(r["entropy"], float(ent.detach()), rtol=2e-6)
    np.testing.assert_allclose(r["kl"], float(kl.detach()), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(r["dmean"], tm.grad.numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(r["dlog_std"], tl.grad.numpy(), rtol=2e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(64, 6), (100, 17), (7, 1)])
def test_ppo_continuous_loss_kernel_matches_oracle(rlx, dev, B, A):
    import torch
    from oracle.losses import ppo_continuous_loss
    from tests.util import dev_tensor
    rng = np.random.RandomState(B + A)
    mean = rng.randn(B, A).astype(np.float32) * 0.5
    ls = (rng.randn(A) * 0.3).astype(np.float32)
    x = rng.randn(B, A).astype(np.float32)
    adv = rng.randn(B).astype(np.float32)
    om = mean + rng.randn(B, A).astype(np.float32) * 0.1
    os_ = np.exp(ls + rng.randn(A).astype(np.float32) * 0.05)[None].repeat(B, 0).astype(np.float32)
    r = ppo_continuous_loss(mean, ls, x, adv, om, os_, 0.2, 0.01)
    dm = torch.zeros(B, A, device=dev); dl = torch.zeros(A, device=dev)
    sc = torch.zeros(4, device=dev); ra = torch.zeros(B, device=dev); cl = torch.zeros(B, device=dev)
    rlx.ppo_continuous_loss(dev_tensor(mean, dev), A, dev_tensor(ls, dev), dev_tensor(x, dev), dev_tensor(adv, dev),
                            dev_tensor(om, dev), dev_tensor(os_, dev), A, B, A, 0.2, 0.01, 1.0, dm, A, dl, sc, ra, cl, None, 0)
    np.testing.assert_allclose(sc.cpu().numpy(), [r["surrogate"], r["entropy"], r["kl"], r["total"]], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(ra.cpu().numpy(), r["ratio"], rtol=2e-4)
    np.testing.assert_allclose(cl.cpu().numpy(), r["clipped"], rtol=2e-4)
    np.testing.assert_allclose(dm.cpu().numpy(), r["dmean"], rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(dl.cpu().numpy(), r["dlog_std"], rtol=2e-3, atol=2e-6)
