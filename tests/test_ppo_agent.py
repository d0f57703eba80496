"""End-to-end Clipped-PPO iteration: HIP agent vs the CPU oracle agent from identical weights,
identical synthetic env bytes and identical host RNG streams."""
import random

import numpy as np
import pytest

from tolerances import LOSS, WEIGHTS


def _make(dev, n_env, L, playing, batch, epochs, seed=0, lengths=None, kind="image"):
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    shape = (84, 84) if kind == "image" else (7,)
    ep = SyntheticVectorEnvironmentParameters(kind, n_env, shape, 6, episode_length=L, seed=99)
    if lengths is not None:
        ep.episode_lengths = list(lengths)
    env = SyntheticVectorEnvironment(ep, dev)
    ap = ClippedPPOAgentParameters()
    ap.seed = seed
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(playing)
    ap.algorithm.optimization_epochs = epochs
    ap.network_wrappers["main"].batch_size = batch
    return ClippedPPOAgent(ap, env, dev)


@pytest.mark.gpu
def test_ppo_iteration_matches_oracle(rlx, dev):
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, L, playing, batch, epochs = 4, 6, 24, 8, 2
    agent = _make(dev, n_env, L, playing, batch, epochs, seed=0)
    arrays = agent.networks["main"].params.named_arrays()
    oenv = SynthVecEnv(0, n_env, 84 * 84, L, 99)
    o = ClippedPPOAgentOracle(arrays, oenv, 6, batch_size=batch, playing_steps=playing, epochs=epochs)
    o.reset((84, 84))
    state = (random.getstate(), np.random.get_state())
    for it in range(2):
        # ---- HIP
        random.setstate(state[0]); np.random.set_state(state[1])
        hip_actions = []
        while True:
            agent.act()
            hip_actions.append(agent.actions.cpu().numpy().copy())
            res = agent.train()
            if res is not None:
                break
        hip_state = (random.getstate(), np.random.get_state())
        # ---- oracle, same RNG streams
        random.setstate(state[0]); np.random.set_state(state[1])
        for s in range(len(hip_actions)):
            oa, _ = o.act()
            assert oa == hip_actions[s].tolist(), "action selection differs at step %d" % s
        ores = o.train()
        assert random.getstate() == hip_state[0]               # identical host RNG consumption
        state = hip_state
        hres = np.array([r.cpu().numpy()[:5] for r in res], dtype=np.float64)
        np.testing.assert_allclose(hres, np.array(ores), **LOSS)
        hw = agent.networks["main"].params.named_arrays()
        for name, per_tower in o.net.weights().items():
            for t, ref in per_tower.items():
                np.testing.assert_allclose(hw[name][t], ref, err_msg=name, **WEIGHTS)
    st = agent.episode_statistics()
    assert st["episodes"] == 2 * n_env and st["mean_length"] == L


@pytest.mark.gpu
@pytest.mark.parametrize("kind,lengths,playing", [("vector", (5, 8, 13, 8), 40), ("image", (3, 7, 4), 18),
                                                  ("vector", (9,), 20)])
def test_ppo_with_envs_ending_on_different_steps_matches_oracle(rlx, dev, kind, lengths, playing):
    """Per-env time limits: episodes end on different steps.  Training starts once the COMPLETE episodes hold
    `playing` transitions, the dataset lists them in completion order, open tails are dropped — the oracle agent in
    ragged mode (pinned for one env to the real reference agent's loop, tests/test_update_pins.py) must see the same
    actions, train at the same steps and reach the same weights."""
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, batch, epochs = len(lengths), 8, 2
    agent = _make(dev, n_env, max(lengths), playing, batch, epochs, seed=0, lengths=lengths, kind=kind)
    assert agent.ragged
    arrays = agent.networks["main"].params.named_arrays()
    elems = 84 * 84 if kind == "image" else 7
    oenv = SynthVecEnv(0 if kind == "image" else 1, n_env, elems, max(lengths), 99, episode_lengths=list(lengths))
    o = ClippedPPOAgentOracle(arrays, oenv, 6, batch_size=batch, playing_steps=playing, epochs=epochs, ragged=True)
    o.reset((84, 84) if kind == "image" else None)
    state = (random.getstate(), np.random.get_state())
    for it in range(3):
        random.setstate(state[0]); np.random.set_state(state[1])
        hip_actions = []
        while True:
            agent.act()
            hip_actions.append(agent.actions.cpu().numpy().copy())
            n_complete = agent.memory.num_transitions_in_complete_episodes()
            res = agent.train()
            if res is not None:
                break
        hip_state = (random.getstate(), np.random.get_state())
        random.setstate(state[0]); np.random.set_state(state[1])
        for s_ in range(len(hip_actions)):
            assert not o.should_train(), "the oracle would have trained at step %d" % s_
            oa, _ = o.act()
            assert oa == hip_actions[s_].tolist(), "action selection differs at step %d" % s_
        assert o.should_train() and o.complete_transitions() == n_complete
        ores = o.train()
        assert random.getstate() == hip_state[0]
        assert np.array_equal(np.random.get_state()[1], hip_state[1][1])     # per-step draws: same np.random stream
        state = hip_state
        hres = np.array([r.cpu().numpy()[:5] for r in res], dtype=np.float64)
        np.testing.assert_allclose(hres, np.array(ores), **LOSS)
        hw = agent.networks["main"].params.named_arrays()
        for name, per_tower in o.net.weights().items():
            for t, ref in per_tower.items():
                np.testing.assert_allclose(hw[name][t], ref, err_msg=name, **WEIGHTS)


@pytest.mark.gpu
def test_ppo_advantages_match_oracle_on_c2_shape(rlx, dev):
    """The BASELINE C2 rollout (64 envs x 32 steps): V, GAE and standardised advantages of the HIP
    agent against oracle.returns on the same V predictions."""
    from oracle import returns as R
    agent = _make(dev, 64, 32, 2048, 64, 1, seed=1)
    for _ in range(32):
        agent.act()
    assert agent._should_train()
    agent.fill_advantages()
    n = agent.memory.num_transitions()
    assert n == 2048
    rew = agent.ds_reward[:n].cpu().numpy()
    val = agent.ds_value[:n].cpu().numpy()
    done = agent.ds_done[:n].cpu().numpy().astype(bool)
    assert done.reshape(64, 32)[:, -1].all() and done.sum() == 64
    adv, vt, _ = R.fill_advantages(rew, val, done, 0.99, 0.95)
    np.testing.assert_allclose(agent.ds_adv[:n].cpu().numpy(), adv, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(agent.ds_vtarget[:n].cpu().numpy(), vt, rtol=1e-5, atol=1e-6)
    assert set(np.unique(rew)) <= {-1.0, 0.0, 1.0}


@pytest.mark.gpu
def test_continuous_ppo_with_observation_normalization(dev):
    """Mujoco_ClippedPPO-style agent: continuous head, AdditiveNoise sampling from the policy std,
    ObservationNormalizationFilter as pre-network filter.  hipGraph replay must equal eager execution
    bit for bit; the running statistics must have seen the state and the next state of every transition of every rollout once."""
    import random
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)

    def run(use_graphs):
        ep = SyntheticVectorEnvironmentParameters("vector", 8, (11,), None, action_dim=3, episode_length=8, seed=3)
        env = SyntheticVectorEnvironment(ep, dev)
        ap = ClippedPPOAgentParameters()
        ap.seed = 6
        ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(64)
        ap.algorithm.optimization_epochs = 3
        ap.algorithm.reward_clipping = None
        ap.algorithm.normalize_observations = True
        net = ap.network_wrappers["main"]
        net.batch_size, net.embedder_scheme, net.middleware_scheme = 16, [64], [64]
        agent = ClippedPPOAgent(ap, env, dev, use_graphs=use_graphs)
        random.seed(1); np.random.seed(1)
        out = []
        for _ in range(3):
            res = None
            while res is None:
                agent.act()
                res = agent.train()
            out.append(torch.stack(res).cpu().numpy())
        agent.networks["main"].check_status()
        return agent, np.array(out)

    a, ra = run(True)
    b, rb = run(False)
    np.testing.assert_array_equal(ra, rb)
    assert np.isfinite(ra).all()
    assert torch.equal(a.networks["main"].params.weights, b.networks["main"].params.weights)
    # states AND next states of every transition enter the statistics (InputFilter.filter, filters/filter.py:314-333)
    assert abs(float(a.norm.count.item()) - (3 * 2 * 64 + 1e-2)) < 1e-9
    ls = a.networks["main"].params.w("main/ppo_head/policy_log_std").cpu().numpy()
    assert np.abs(ls).max() > 0 and np.isfinite(ls).all()           # the log-std variable is being trained
    assert a.actions.dtype == torch.float32 and bool(torch.isfinite(a.actions).all())


@pytest.mark.gpu
def test_decaying_clip_schedule_reuses_the_captured_graphs(rlx, dev):
    """clipping_decay_schedule = LinearSchedule (presets/Mujoco_ClippedPPO.py): the clip range changes every phase.
    It reaches the loss kernel through a device scalar, so the epoch / act graphs captured in the first phases are
    replayed afterwards (no new graph per value), and the results equal the eager run of the same schedule."""
    import torch
    from coach_amd.schedules import LinearSchedule
    runs = {}
    for graphs in (True, False):
        random.seed(3); np.random.seed(3)
        agent = _make(dev, 4, 6, 24, 8, 2, seed=0, kind="vector")
        agent.use_graphs = graphs
        agent.ap.algorithm.clipping_decay_schedule = LinearSchedule(1.0, 0.1, 40)
        sizes, res = [], []
        for it in range(5):
            while True:
                agent.act()
                r = agent.train()
                if r is not None:
                    break
            res.append(np.array([x.cpu().numpy()[:5] for x in r]))
            sizes.append(len(agent._graphs))
        runs[graphs] = (np.array(res), agent.networks["main"].params.weights.clone(), sizes,
                        float(agent.ap.algorithm.clipping_decay_schedule.current_value))
    assert runs[True][3] < 0.9                                        # the schedule did decay
    assert runs[True][2][2] == runs[True][2][-1] > 0                  # no graphs added after the third phase
    np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=1e-6, atol=1e-7)
    assert torch.equal(runs[True][1], runs[False][1])


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [False, True])
def test_continuous_ppo_iteration_matches_oracle(rlx, dev, normalize):
    """Mujoco_ClippedPPO-style loop (BoxActionSpace, PPOHead mean / std, actions = clip(N(mean, std))) against the oracle
    agent in continuous mode — whose train() is pinned to the REAL reference agent's train on a BoxActionSpace
    (tests/test_update_pins.py, fixture ppoc).  Same host streams: the sampled actions, the per-epoch results and the
    weights (policy_log_std included) after two iterations.  normalize: with the ObservationNormalizationFilter as
    pre-network filter — the oracle's loop in that mode is pinned to the REAL reference agent's loop
    (tests/golden/ppoc_loop.npz): acting on the statistics as they are, the dataset's states pushed and normalised, then
    its next states pushed; the running statistics are compared after every phase."""
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, L, D, A, playing, batch, epochs = 4, 6, 7, 3, 24, 8, 2
    ep = SyntheticVectorEnvironmentParameters("vector", n_env, (D,), None, action_dim=A, episode_length=L, seed=99)
    env = SyntheticVectorEnvironment(ep, dev)
    ap = ClippedPPOAgentParameters()
    ap.seed = 0
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(playing)
    ap.algorithm.optimization_epochs = epochs
    ap.algorithm.reward_clipping = None
    ap.algorithm.normalize_observations = normalize
    net = ap.network_wrappers["main"]
    net.batch_size, net.embedder_scheme, net.middleware_scheme = batch, [32], [32]
    agent = ClippedPPOAgent(ap, env, dev)
    assert agent.continuous
    arrays = agent.networks["main"].params.named_arrays()
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(1, n_env, D, L, 99), A, batch_size=batch, playing_steps=playing,
                              epochs=epochs, reward_clip=None, continuous=True, normalize=normalize)
    o.reset()
    state = (random.getstate(), np.random.get_state())
    for it in range(2):
        random.setstate(state[0]); np.random.set_state(state[1])
        hip_actions = []
        while True:
            agent.act()
            hip_actions.append(agent.actions.cpu().numpy().copy())
            res = agent.train()
            if res is not None:
                break
        hip_state = (random.getstate(), np.random.get_state())
        random.setstate(state[0]); np.random.set_state(state[1])
        for s_ in range(len(hip_actions)):
            oa, _ = o.act()
            np.testing.assert_allclose(hip_actions[s_], np.stack(oa), rtol=2e-5, atol=2e-6,
                                       err_msg="sampled actions differ at step %d" % s_)
        ores = o.train()
        assert random.getstate() == hip_state[0]
        state = hip_state
        hres = np.array([r.cpu().numpy()[:5] for r in res], dtype=np.float64)
        np.testing.assert_allclose(hres, np.array(ores), **LOSS)
        hw = agent.networks["main"].params.named_arrays()
        for name, per_tower in o.net.weights().items():
            for t, ref in per_tower.items():
                np.testing.assert_allclose(hw[name][t], ref, err_msg=name, **WEIGHTS)
        if normalize:                            # states + next states of the phase's transitions, fp64 on both sides
            assert abs(float(agent.norm.count.item()) - o.stats._count) < 1e-9
            assert abs(o.stats._count - ((it + 1) * 2 * playing + 1e-2)) < 1e-9
            np.testing.assert_allclose(agent.norm.mean.cpu().numpy(), o.stats._mean, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(agent.norm.std.cpu().numpy(), o.stats._std, rtol=1e-12, atol=1e-12)
