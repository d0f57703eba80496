"""Episodic replay with episodes of DIFFERENT lengths, whole-episode eviction and n-step returns:
oracle vs traces of the reference's EpisodicExperienceReplay + Episode classes (tests/golden/episodic.npz, CPU);
the device memory vs the same traces (one env: every draw selects what the reference selects; returns bit-exact)
and vs the oracle for several envs finishing on different steps (GPU)."""
import numpy as np
import pytest

from oracle.replay import EpisodicReplayOracle

VARIANTS = ["to_end", "n3", "episodes"]
DISCOUNT = 0.97


@pytest.mark.parametrize("name", VARIANTS)
def test_oracle_matches_reference_trace(golden, name):
    g = golden("episodic")
    n_step, by_ep, size, seed = (int(x) for x in g[name + "_meta"])
    o = EpisodicReplayOracle(size, n_step, DISCOUNT, by_episodes=bool(by_ep))
    rewards, k = g[name + "_rewards"], 0
    np.random.seed(seed)
    for i, L in enumerate(g[name + "_lens"]):
        r = rewards[k:k + L]
        o.store_episode(list(r), r)
        k += L
        c = g[name + "_counters"][i]
        assert [o.num_transitions(), o.num_transitions(), o.num_complete_episodes(), o.num_complete_episodes()] == c.tolist()
        assert np.array_equal(np.array(o.rows), g["%s_held_r_%d" % (name, i)])
        assert np.array_equal(np.array(o.nsr), g["%s_held_nsr_%d" % (name, i)])          # fp64, bit-exact
        idx = o.sample_indices(6)
        assert np.array_equal(np.array(o.rows)[idx], g[name + "_sampled"][i])


def _memory(dev, size, by_ep, n_step, n_env, tmax):
    from coach_amd.memories.episodic.episodic_experience_replay import EpisodicExperienceReplay
    from coach_amd.memories.memory import MemoryGranularity
    unit = MemoryGranularity.Episodes if by_ep else MemoryGranularity.Transitions
    return EpisodicExperienceReplay((unit, size), n_step=n_step, discount=DISCOUNT, max_episode_length=tmax, device=dev,
                                    n_env=n_env, observation_shape=(1,), min_episode_length=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", VARIANTS)
def test_device_memory_matches_reference_trace(golden, dev, name):
    import torch
    g = golden("episodic")
    n_step, by_ep, size, seed = (int(x) for x in g[name + "_meta"])
    lens = g[name + "_lens"]
    m = _memory(dev, size, bool(by_ep), n_step, 1, int(lens.max()))
    rewards, k = g[name + "_rewards"], 0
    m.reset(torch.zeros(1, 1, device=dev))
    np.random.seed(seed)
    t = lambda v, dt: torch.tensor([v], dtype=dt, device=dev)
    for i, L in enumerate(lens):
        for j in range(L):
            done = j == L - 1
            # rewards travel as fp32 on the device; the fixture's are fp64 — store the fp32 value, compare with
            # the oracle re-run on the same fp32 values below
            m.store(t(0, torch.int32).view(1), t(np.float32(rewards[k]), torch.float32).view(1),
                    t(int(done), torch.uint8).view(1), t(float(k + 1), torch.float32).view(1, 1),
                    t(0.0, torch.float32).view(1, 1), dones_host=np.array([done]))
            k += 1
        c = g[name + "_counters"][i]
        assert [m.num_transitions(), m.num_transitions_in_complete_episodes(), m.num_complete_episodes(), m.length()] \
            == c.tolist()
        n = m.num_transitions()
        b = m.gather(m.physical_rows(np.arange(n)), n)
        m.check_status()
        held = g["%s_held_r_%d" % (name, i)]
        assert np.array_equal(b["reward"].cpu().numpy(), held.astype(np.float32))          # same transitions, same order
        ref32 = np.concatenate([EpisodicReplayOracle.n_step_returns(e, DISCOUNT, n_step) for e in
                                np.split(held.astype(np.float32), np.cumsum(m.episode_lengths())[:-1])])
        assert np.array_equal(b["n_step_discounted_rewards"].cpu().numpy(), ref32)          # fp64 bit-exact
        np.testing.assert_allclose(ref32, g["%s_held_nsr_%d" % (name, i)], rtol=1e-6, atol=1e-6)
        idx = m.sample_indices(6)
        bs = m.gather(m.physical_rows(idx), 6)
        assert np.array_equal(bs["reward"].cpu().numpy(), g[name + "_sampled"][i].astype(np.float32))


@pytest.mark.gpu
def test_envs_finishing_on_different_steps(dev):
    """3 envs with time limits 5 / 3 / 4 and a 20-transition memory: listed transitions, their order (completion
    order, ties in env order), eviction and returns follow the oracle fed the same episodes; a reset in the middle
    of the running episodes drops them."""
    import torch
    lens = [5, 3, 4]
    m = _memory(dev, 20, False, -1, 3, 5)
    o = EpisodicReplayOracle(20, -1, DISCOUNT)
    rng = np.random.RandomState(3)
    t_in = [0, 0, 0]
    open_rows = [[], [], []]
    m.reset(torch.zeros(3, 1, device=dev))
    for step in range(40):
        r = rng.uniform(-1, 1, 3).astype(np.float32)
        done = np.array([t_in[e] + 1 >= lens[e] for e in range(3)])
        m.store(torch.zeros(3, dtype=torch.int32, device=dev), torch.from_numpy(r).to(dev),
                torch.from_numpy(done.astype(np.uint8)).to(dev), torch.full((3, 1), float(step), device=dev),
                torch.zeros(3, 1, device=dev), dones_host=done)
        for e in range(3):
            open_rows[e].append(r[e])
            t_in[e] += 1
            if done[e]:
                o.store_episode(list(open_rows[e]), np.array(open_rows[e]))
                open_rows[e], t_in[e] = [], 0
        if step == 25:                                    # forced reset: the running episodes never reach the memory
            m.drop_open_episode()
            open_rows, t_in = [[], [], []], [0, 0, 0]
        assert m.num_transitions() == o.num_transitions() <= 20
        assert m.num_complete_episodes() == o.num_complete_episodes()
        if o.num_transitions():
            n = o.num_transitions()
            b = m.gather(m.physical_rows(np.arange(n)), n)
            assert np.array_equal(b["reward"].cpu().numpy(), np.array(o.rows, dtype=np.float32))
            assert np.array_equal(b["n_step_discounted_rewards"].cpu().numpy(), np.array(o.nsr))
    m.check_status()
    assert m.episode_lengths() == o.episodes


@pytest.mark.gpu
def test_dropped_open_episodes_never_overrun_the_ring(dev):
    """Every drop_open_episode() (a forced reset in the middle of an episode: evaluations, period starts) leaves written
    but never-listed steps in the ring.  However many of those accumulate, the memory keeps working: the slot a new step
    needs is freed by evicting the oldest listed episode BEFORE the write, the listed episodes are always intact."""
    import torch
    m = _memory(dev, 24, False, -1, 1, 6)
    m.reset(torch.zeros(1, 1, device=dev))
    t = lambda v, dt: torch.tensor([v], dtype=dt, device=dev)
    k, listed = 0, []
    for rnd in range(60):
        for j in range(4):                                   # four steps of an episode that is then abandoned
            m.store(t(0, torch.int32).view(1), t(-1.0, torch.float32).view(1), t(0, torch.uint8).view(1),
                    t(0.0, torch.float32).view(1, 1), t(0.0, torch.float32).view(1, 1), dones_host=np.array([False]))
        m.drop_open_episode()
        ep = []
        for j in range(3):                                   # ... and a complete one
            k += 1
            ep.append(float(k))
            m.store(t(0, torch.int32).view(1), t(float(k), torch.float32).view(1), t(int(j == 2), torch.uint8).view(1),
                    t(0.0, torch.float32).view(1, 1), t(0.0, torch.float32).view(1, 1), dones_host=np.array([j == 2]))
        listed.append(ep)
        n = m.num_transitions()
        assert 3 <= n <= 24 and n % 3 == 0
        held = m.gather(m.physical_rows(np.arange(n)), n)["reward"].cpu().numpy()
        want = np.array([r for e in listed[-(n // 3):] for r in e], dtype=np.float32)
        assert np.array_equal(held, want), rnd                # the newest n / 3 complete episodes, untouched
    m.check_status()
