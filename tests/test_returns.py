"""K8/K12: GAE, discounted returns, standardisation, episode statistics."""
import numpy as np
import pytest

from oracle import returns as R
from tests.util import dev_tensor

CASES = ["small", "ppo2048", "ragged"]


# ------------------------------------------------------------------------------------ CPU
def test_oracle_gae_appendix_b(golden):
    g = golden("gae")
    gae, vt = R.gae_episode([1, 0, 2, -1], [.5, .6, .7, .2, 0], 0.99, 0.95)
    np.testing.assert_allclose(gae, g["appB_gae"], rtol=1e-15)
    np.testing.assert_allclose(vt, g["appB_vt"], rtol=1e-15)
    np.testing.assert_allclose(gae, [1.5082156683, 0.4404207, 0.3694, -1.2], rtol=1e-9)
    np.testing.assert_allclose(R.discounted_returns_episode([1, 0, 2, -1], 0.99), g["appB_returns"],
                               rtol=0, atol=0)
    np.testing.assert_allclose(g["appB_returns"], [1.989901, 0.9999, 1.01, -1.0], rtol=1e-12)


@pytest.mark.parametrize("name", CASES)
def test_oracle_fill_advantages_matches_reference(golden, name):
    g = golden("gae")
    disc, lam = g[name + "_hp"]
    adv, vt, _ = R.fill_advantages(g[name + "_rewards"], g[name + "_values"], g[name + "_go"], disc, lam)
    assert np.array_equal(adv, g[name + "_adv"])            # bit-exact fp64
    assert np.array_equal(vt, g[name + "_vt"])
    assert np.array_equal(R.discounted_returns(g[name + "_rewards"], g[name + "_go"], disc),
                          g[name + "_returns"])


# ------------------------------------------------------------------------------------ GPU
def _hip_fill_advantages(rlx, dev, rewards, values, go, disc, lam, n_seq=1):
    import torch
    T = len(rewards)
    adv = torch.empty(T, dtype=torch.float64, device=dev)
    vt = torch.empty(T, dtype=torch.float32, device=dev)
    std = torch.empty(T, dtype=torch.float64, device=dev)
    std32 = torch.empty(T, dtype=torch.float32, device=dev)
    ms = torch.empty(2, dtype=torch.float64, device=dev)
    rlx.gae(dev_tensor(rewards, dev, np.float32), dev_tensor(values, dev, np.float32),
            dev_tensor(go, dev, np.uint8), None, n_seq, T // n_seq, disc, lam, adv, vt, 0)
    rlx.standardize(adv, T, std32, std, ms, 0)
    return adv.cpu().numpy(), vt.cpu().numpy(), std.cpu().numpy(), std32.cpu().numpy(), ms.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_fill_advantages_matches_reference(golden, rlx, dev, name):
    g = golden("gae")
    disc, lam = g[name + "_hp"]
    rewards, values, go = g[name + "_rewards"], g[name + "_values"], g[name + "_go"]
    raw, vt, std, std32, ms = _hip_fill_advantages(rlx, dev, rewards, values, go, disc, lam)
    _, ref_vt, ref_raw = R.fill_advantages(rewards, values, go, disc, lam)
    # fp64 scan re-associates the recurrence: tolerance, not bit-exact (stated in DESIGN.md)
    np.testing.assert_allclose(raw, ref_raw, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(vt, ref_vt.astype(np.float32), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(std, g[name + "_adv"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(std32, g[name + "_adv"].astype(np.float32), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ms, [ref_raw.mean(), ref_raw.std()], rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_hip_gae_env_major_layout_equals_flat(golden, rlx, dev):
    """64 trajectories of 32 steps (the C2 rollout) as n_seq=64 gives the same numbers as one flat
    sequence with game_over at every trajectory end."""
    g = golden("gae")
    disc, lam = g["ppo2048_hp"]
    a = _hip_fill_advantages(rlx, dev, g["ppo2048_rewards"], g["ppo2048_values"], g["ppo2048_go"], disc, lam, 1)
    b = _hip_fill_advantages(rlx, dev, g["ppo2048_rewards"], g["ppo2048_values"], g["ppo2048_go"], disc, lam, 64)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-13, atol=1e-14)
    assert np.array_equal(a[1], b[1]) or np.allclose(a[1], b[1], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("T,n_seq", [(1, 1), (1023, 1), (1024, 1), (1025, 1), (65536, 1), (1000, 256), (1 << 20, 1)])
def test_hip_gae_and_returns_vs_oracle_sizes(rlx, dev, T, n_seq):
    """Edge sizes: single step, tile boundaries (tile = 1024), BASELINE rollout sizes, 1M steps."""
    import torch
    rng = np.random.RandomState(T % 9973 + n_seq)
    total = T * n_seq if n_seq > 1 else T
    rewards = rng.randn(total).astype(np.float32)
    values = rng.randn(total).astype(np.float32)
    go = rng.rand(total) < (0.01 if total > 4096 else 0.15)
    go[T - 1::T] = True                                  # every trajectory ends an episode
    adv = torch.empty(total, dtype=torch.float64, device=dev)
    vt = torch.empty(total, dtype=torch.float32, device=dev)
    ret = torch.empty(total, dtype=torch.float64, device=dev)
    r_d, v_d, g_d = dev_tensor(rewards, dev), dev_tensor(values, dev), dev_tensor(go, dev, np.uint8)
    rlx.gae(r_d, v_d, g_d, None, n_seq, T, 0.99, 0.95, adv, vt, 0)
    rlx.discounted_returns(r_d, g_d, n_seq, T, 0.99, ret, None, 0)
    # vectorised fp64 reference of the same recurrences (the O(T^2) oracle is too slow at 1M)
    r64, v64 = rewards.astype(np.float64), values.astype(np.float64)
    ref_adv, ref_ret = np.zeros(total), np.zeros(total)
    acc_a = acc_r = 0.0
    vnext = 0.0
    for t in range(total - 1, -1, -1):
        if go[t]:
            acc_a = acc_r = 0.0
            vnext = 0.0
        delta = r64[t] + 0.99 * vnext - v64[t]
        acc_a = delta + (0.99 * 0.95) * acc_a
        acc_r = r64[t] + 0.99 * acc_r
        ref_adv[t], ref_ret[t] = acc_a, acc_r
        vnext = v64[t]
    np.testing.assert_allclose(adv.cpu().numpy(), ref_adv, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ret.cpu().numpy(), ref_ret, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(vt.cpu().numpy(), (ref_adv + v64).astype(np.float32), rtol=1e-6, atol=1e-6)
    if total <= 2048:
        np.testing.assert_allclose(ret.cpu().numpy(), R.discounted_returns(rewards, go, 0.99),
                                   rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_hip_gae_bootstrap_for_truncated_rollout(rlx, dev):
    """Not in the reference (it drops incomplete episodes): bootstrap_values continue a truncated
    trajectory.  Property: splitting a trajectory and bootstrapping with V of the cut state gives
    the same value TARGET recursion as lambda=1 returns."""
    import torch
    rng = np.random.RandomState(1)
    T = 50
    r = rng.randn(T).astype(np.float32)
    v = rng.randn(T).astype(np.float32)
    go = np.zeros(T, dtype=np.uint8)
    boot = np.array([0.7], dtype=np.float32)
    adv = torch.empty(T, dtype=torch.float64, device=dev)
    vt = torch.empty(T, dtype=torch.float32, device=dev)
    rlx.gae(dev_tensor(r, dev), dev_tensor(v, dev), dev_tensor(go, dev), dev_tensor(boot, dev), 1, T,
            0.9, 1.0, adv, vt, 0)
    ref = np.zeros(T)
    acc = 0.7
    for t in range(T - 1, -1, -1):
        acc = r[t] + 0.9 * acc
        ref[t] = acc
    np.testing.assert_allclose(vt.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n_env", [1, 64, 512, 3000])
def test_hip_episode_stats_vs_oracle(rlx, dev, n_env):
    import torch
    rng = np.random.RandomState(n_env)
    o = R.EpisodeStatsOracle(n_env)
    ep_ret = torch.empty(n_env, dtype=torch.float64, device=dev)
    ep_len = torch.empty(n_env, dtype=torch.int32, device=dev)
    acc = torch.empty(8, dtype=torch.float64, device=dev)
    last_ret = torch.zeros(n_env, dtype=torch.float64, device=dev)
    last_len = torch.zeros(n_env, dtype=torch.int32, device=dev)
    rlx.episode_stats_init(ep_ret, ep_len, n_env, acc, 0)
    assert np.array_equal(acc.cpu().numpy(), o.acc())
    for s in range(25):
        rew = rng.randn(n_env).astype(np.float32)
        done = (rng.rand(n_env) < 0.2).astype(np.uint8)
        o.step(rew, done)
        rlx.episode_stats_step(dev_tensor(rew, dev), dev_tensor(done, dev), ep_ret, ep_len, n_env, acc,
                               last_ret, last_len, 0)
    np.testing.assert_allclose(acc.cpu().numpy(), o.acc(), rtol=1e-12)
    np.testing.assert_allclose(ep_ret.cpu().numpy(), o.ep_return, rtol=1e-12, atol=1e-12)
    assert np.array_equal(ep_len.cpu().numpy(), o.ep_len)
