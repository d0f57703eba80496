"""The DEVICE DQN agent against the REAL reference DQNAgent's recorded loop (tests/golden/loop.npz:
`rl_coach.agents.dqn_agent.DQNAgent` built by its own __init__ with its own ExperienceReplay /
PrioritizedExperienceReplay / EGreedy, stepped through LevelManager.step's cycle — observe the previous
response, act, env.step, terminal responses observed at once — with train() after every step; generated
by tests/golden/make_golden.py::gen_loop).

No oracle in between: every action, the number of transitions visible at every train(), every sampled
transition and the final weights of the HIP engine are compared with what the reference itself did, for
uniform and for prioritized replay (p**alpha on the device, csrc/libm_pow.hpp).  This is the
"bit-exact for replay index selection" clause of BASELINE.json's north_star at loop level.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _agent(dev, fx, variant):
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    from coach_amd.memories.non_episodic.prioritized_experience_replay import \
        PrioritizedExperienceReplayParameters
    from coach_amd.schedules import LinearSchedule
    D, A, L, B, CAP, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    env = SyntheticVectorEnvironment(
        SyntheticVectorEnvironmentParameters("vector", 1, (D,), A, episode_length=L, seed=99), dev)
    ap = DQNAgentParameters()
    ap.seed = SEED
    if variant == "per":
        ap.memory = PrioritizedExperienceReplayParameters()              # alpha .6, constant beta .4
    ap.memory.max_size = (MemoryGranularity.Transitions, CAP)
    net = ap.network_wrappers["main"]
    net.batch_size, net.learning_rate, net.replace_mse_with_huber_loss = B, 1e-3, False
    net.embedder_scheme, net.middleware_scheme = [16], [12]
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
    ap.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(10)
    ap.exploration.epsilon_schedule = LinearSchedule(1.0, 0.1, 50)
    agent = DQNAgent(ap, env, dev)
    params = agent.networks["main"].params
    import torch
    for k in fx.files:
        if k.startswith("init|"):
            params.w(k[len("init|"):]).copy_(torch.from_numpy(fx[k]).to(dev))
    agent.networks["main"].update_target(1.0)
    return agent, (HEATUP, TRAIN, SEED, CAP)


@pytest.mark.parametrize("variant", ["uniform", "per"])
def test_device_dqn_loop_equals_real_reference_agent_loop(dev, variant):
    from coach_amd.core_types import RunPhase
    fx = np.load(os.path.join(HERE, "golden", "loop.npz"))
    agent, (HEATUP, TRAIN, SEED, CAP) = _agent(dev, fx, variant)
    random.seed(SEED)
    np.random.seed(SEED)
    agent.exploration_policy.current_random_value[:] = np.random.rand()     # e_greedy.py:82, after seeding
    actions, visible, keys = [], [], []
    collate = agent.memory.collate

    def logged(d, B):
        batch = collate(d, B)
        keys.append(batch._states["observation"][:, 0].cpu().numpy().astype(np.float64).tolist())
        visible.append(agent.memory.num_transitions())
        return batch
    agent.memory.collate = logged
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.act()
        actions.append(int(agent.actions.cpu()[0]))
        if step >= HEATUP:
            agent.train()
    agent.check_status()
    np.testing.assert_array_equal(visible, fx[variant + "|visible"])      # transitions visible at train()
    assert max(visible) == CAP                                            # the FIFO / leaf ring did wrap
    ref_keys = fx[variant + "|keys"]
    if variant == "uniform":
        np.testing.assert_array_equal(actions, fx[variant + "|actions"])
        np.testing.assert_array_equal(np.array(keys), ref_keys.astype(np.float32).astype(np.float64))
    else:
        # priorities are |TD errors| of fp32 networks whose accumulation order differs between the GPU
        # GEMMs and numpy, so a leaf can differ in its last bits and — rarely — move a stratified draw
        # across a leaf boundary; everything before such an event must be identical
        same = 0
        for a, b in zip(keys, ref_keys.astype(np.float32).astype(np.float64).tolist()):
            if a != b:
                break
            same += 1
        assert same >= 40, "PER samples diverged from the reference after %d batches" % same
        np.testing.assert_array_equal(actions[:HEATUP + same], fx[variant + "|actions"][:HEATUP + same])
        if same < len(keys):
            return
    w = agent.networks["main"].params.named_arrays()
    n = 0
    for k in fx.files:
        if k.startswith(variant + "|final|"):
            _, _, name, t = k.split("|")
            np.testing.assert_allclose(w[name][int(t)], fx[k], rtol=0, atol=3e-5, err_msg=name)
            n += 1
    assert n > 0


@pytest.mark.parametrize("name", ["td3", "ddpg"])
def test_device_td3_and_ddpg_loops_equal_real_reference_agent_loops(dev, name):
    """The DEVICE TD3 / DDPG agents against the recorded loops of the REAL reference `TD3Agent` / `DDPGAgent`
    (tests/golden/{td3,ddpg}_loop.npz: own __init__, EpisodicExperienceReplay, AdditiveNoise / OUProcess, observe /
    act / train; tests/golden/make_golden.py::gen_td3_loop / gen_ddpg_loop), no oracle in between: every RECORDED
    (unclipped, noisy) action, the training iteration after every step, every sampled transition and the final
    weights."""
    import torch
    from coach_amd.agents.ddpg_agent import DDPGAgent, DDPGAgentParameters
    from coach_amd.agents.td3_agent import TD3Agent, TD3AgentParameters
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    fx = np.load(os.path.join(HERE, "golden", name + "_loop.npz"))
    D, A, L, B, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    env = SyntheticVectorEnvironment(
        SyntheticVectorEnvironmentParameters("vector", 1, (D,), None, action_dim=A, episode_length=L,
                                             seed=55 if name == "td3" else 56), dev)
    p = TD3AgentParameters() if name == "td3" else DDPGAgentParameters()
    p.seed = 3
    for n in p.network_wrappers.values():
        n.batch_size = B
    an, cn = p.network_wrappers["actor"], p.network_wrappers["critic"]
    an.observation_embedder_scheme, an.middleware_scheme = (20,), (12,)
    if name == "td3":
        cn.middleware_scheme = (20, 12)
    else:
        cn.observation_embedder_scheme, cn.middleware_scheme = (18,), (12,)
    agent = (TD3Agent if name == "td3" else DDPGAgent)(p, env, dev)
    for k in fx.files:
        if k.startswith("init|"):
            _, pname, t = k.split("|")
            net = agent.networks[pname.split("/")[0]]
            net.params.w(pname, int(t)).copy_(torch.from_numpy(fx[k]).to(dev))
    for net in agent.networks.values():
        net.update_target(1.0)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, iters, keys = [], [], []
    collate = agent.memory.collate

    def logged(d, B_):
        batch = collate(d, B_)
        keys.append(batch._states["observation"][:, 0].cpu().numpy().astype(np.float64).tolist())
        return batch
    agent.memory.collate = logged
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.act()
        actions.append(agent.actions.cpu().numpy()[0].astype(np.float64))
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    agent.check_status()
    np.testing.assert_array_equal(iters, fx["iters"])
    assert iters[-1] > 0
    np.testing.assert_array_equal(np.array(keys), fx["keys"].astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(np.array(actions), fx["actions"], rtol=2e-5, atol=2e-6)
    n = 0
    for k in fx.files:
        if k.startswith("final|"):
            _, net_name, pname, t = k.split("|")
            w = agent.networks[net_name].params.named_arrays()
            np.testing.assert_allclose(w[pname][int(t)], fx[k], rtol=2e-3, atol=6e-5, err_msg=pname)
            n += 1
    assert n > 0


def test_device_sac_loop_equals_real_reference_agent_loop(dev):
    """The DEVICE Soft Actor-Critic agent against the recorded loop of the REAL reference `SoftActorCriticAgent`
    (tests/golden/sac_loop.npz, make_golden.py::gen_sac_loop; the TF sampling op stood in by np.random.standard_normal,
    one draw per policy pass): every recorded action (heat-up samples, then squashed policy samples), the training
    iteration after every step, the number of transitions visible at every train(), every sampled transition and the
    final weights of the policy, twin-Q and V networks."""
    import torch
    from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent, SoftActorCriticAgentParameters
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    fx = np.load(os.path.join(HERE, "golden", "sac_loop.npz"))
    D, A, L, B, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    env = SyntheticVectorEnvironment(
        SyntheticVectorEnvironmentParameters("vector", 1, (D,), None, action_dim=A, episode_length=L, seed=57), dev)
    p = SoftActorCriticAgentParameters()
    p.seed = 3
    p.algorithm.reward_rescale = 1.0                                     # the golden run has no reward filter
    for n in p.network_wrappers.values():
        n.batch_size = B
    p.network_wrappers["policy"].embedder_scheme, p.network_wrappers["policy"].middleware_scheme = (20,), (12,)
    p.network_wrappers["v"].embedder_scheme, p.network_wrappers["v"].middleware_scheme = (20,), (12,)
    p.network_wrappers["q"].network_layers_sizes = (14, 14)
    agent = SoftActorCriticAgent(p, env, dev)
    for k in fx.files:
        if k.startswith("init|"):
            _, pname, t = k.split("|")
            agent.networks[pname.split("/")[0]].params.w(pname, int(t)).copy_(torch.from_numpy(fx[k]).to(dev))
    agent.networks["v"].update_target(1.0)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, iters, keys, visible = [], [], [], []
    collate = agent.memory.collate

    def logged(d, B_):
        batch = collate(d, B_)
        keys.append(batch._states["observation"][:, 0].cpu().numpy().astype(np.float64).tolist())
        visible.append(agent.memory.num_transitions())
        return batch
    agent.memory.collate = logged
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.act()
        actions.append(agent.actions.cpu().numpy()[0].astype(np.float64))
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    agent.check_status()
    np.testing.assert_array_equal(iters, fx["iters"])
    np.testing.assert_array_equal(visible, fx["visible"])
    np.testing.assert_array_equal(np.array(keys), fx["keys"].astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(np.array(actions), fx["actions"], rtol=1e-4, atol=1e-5)
    n = 0
    for k in fx.files:
        if k.startswith("final|"):
            _, net_name, pname, t = k.split("|")
            w = agent.networks[net_name].params.named_arrays()
            np.testing.assert_allclose(w[pname][int(t)], fx[k], rtol=2e-3, atol=6e-5, err_msg=pname)
            n += 1
    assert n > 0
