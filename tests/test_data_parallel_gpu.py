"""The data-parallel paths end to end on ONE GPU box: two ranks (both on cuda:0, gloo backend because
RCCL refuses two ranks on one device) with rank-sharded envs.
Clipped PPO: weights stay bit-identical across ranks without a broadcast; the overlapped two-bucket
all-reduce (FC + heads first, convs later) gives exactly the weights of the single blocking
all-reduce; splitting the backward pass does not change the gradients; observation running statistics
are shared between the ranks.  DQN / TD3 / SAC: the update captured as hipGraph segments cut at every
gradient all-reduce equals eager execution bit for bit.  Real RCCL (backend "nccl") at world size 1."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir, sync_mode, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.distributed import GradientSync
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    dev = torch.device("cuda:0")
    dist = GradientSync(backend=backend, force=True)
    assert dist.enabled and dist.world_size == world
    if sync_mode == "s":                                # collectives between graph segments even on RCCL
        dist.graph_collectives = False
    if backend == "nccl":
        # RCCL collectives are recorded into the update's hipGraph unless switched off
        assert dist.capturable() == (sync_mode != "s")
    ep = SyntheticVectorEnvironmentParameters("image", 8, (44, 44), 4, episode_length=8, seed=21)
    env = SyntheticVectorEnvironment(ep, dev, rank=dist.rank)
    ap = ClippedPPOAgentParameters()
    ap.seed = 4
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(64)
    ap.algorithm.optimization_epochs = 2
    ap.network_wrappers["main"].batch_size = 16
    agent = ClippedPPOAgent(ap, env, dev, dist=dist)
    if sync_mode != "a":                                # "a": the agent decides from its own measurements
        agent.overlap_allreduce = sync_mode == "0"      # "1" / "s": one blocking all-reduce per minibatch
    for _ in range(2):
        res = None
        while res is None:
            agent.act()
            res = agent.train()
    agent.networks["main"].check_status()
    if sync_mode == "a":
        rec = agent.overlap_decision
        assert rec is not None and agent.overlap_allreduce is rec["overlap"]
        if backend == "gloo":                           # not graph-resident: the single blocking all-reduce
            assert rec["overlap"] is False and rec["graph_resident"] is False
        else:                                           # RCCL: both terms were measured on this box
            assert rec["graph_resident"] and rec["conv_backward_us"] > 0 and rec["late_bucket_allreduce_us"] > 0
            assert rec["overlap"] == (min(rec["conv_backward_us"], rec["late_bucket_allreduce_us"]) > rec["edge_pair_us"])
    w = agent.networks["main"].params.weights.cpu().numpy()
    np.save(os.path.join(out_dir, "w_%s_%d.npy" % (sync_mode, rank)), w)
    np.save(os.path.join(out_dir, "obs_%s_%d.npy" % (sync_mode, rank)), env.obs.cpu().numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_ppo_iteration(tmp_path):
    import torch.multiprocessing as mp
    for mode in ("1", "0", "a"):                             # blocking all-reduce, overlapped, the agent's own choice
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    w = {(m, r): np.load(tmp_path / ("w_%s_%d.npy" % (m, r))) for m in "10a" for r in (0, 1)}
    np.testing.assert_array_equal(w[("a", 0)], w[("1", 0)])  # the default (decided at the first phase) == blocking
    np.testing.assert_array_equal(w[("a", 0)], w[("a", 1)])
    assert np.isfinite(w[("0", 0)]).all()
    np.testing.assert_array_equal(w[("1", 0)], w[("1", 1)])  # ranks agree bit for bit
    np.testing.assert_array_equal(w[("0", 0)], w[("0", 1)])
    np.testing.assert_array_equal(w[("0", 0)], w[("1", 0)])  # overlap == blocking
    o0, o1 = np.load(tmp_path / "obs_0_0.npy"), np.load(tmp_path / "obs_0_1.npy")
    assert not np.array_equal(o0, o1)                        # different env shards per rank


def _norm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.distributed import GradientSync
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    dev = torch.device("cuda:0")
    dist = GradientSync(backend="gloo")
    ep = SyntheticVectorEnvironmentParameters("vector", 8, (11,), None, action_dim=3, episode_length=8, seed=3)
    env = SyntheticVectorEnvironment(ep, dev, rank=dist.rank)
    ap = ClippedPPOAgentParameters()
    ap.seed = 6
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(64)
    ap.algorithm.optimization_epochs = 2
    ap.algorithm.reward_clipping = None
    ap.algorithm.normalize_observations = True
    net = ap.network_wrappers["main"]
    net.batch_size, net.embedder_scheme, net.middleware_scheme = 16, [64], [64]
    agent = ClippedPPOAgent(ap, env, dev, dist=dist)
    raw = []
    for _ in range(2):
        res = None
        while res is None:
            agent.act()
            if agent.total_steps_counter - agent.last_training_phase_step >= 64 and agent.memory.steps % 8 == 0:
                n = agent.memory.num_transitions()
                raw.append(agent.memory.gather_states(agent.memory.dataset_rows(), n,
                                                      torch.empty_like(agent.ds_obs_raw[:n])).cpu().numpy())
                # the filter walks the next states of the same transitions too (filters/filter.py:314-333)
                raw.append(agent.memory.gather_next_states(agent.memory.dataset_rows(), n,
                                                           torch.empty_like(agent.ds_obs_raw[:n])).cpu().numpy())
            res = agent.train()
    agent.networks["main"].check_status()
    np.savez(os.path.join(out_dir, "norm_%d.npz" % rank), raw=np.concatenate(raw),
             sum=agent.norm.sum.cpu().numpy(), sq=agent.norm.sum_squares.cpu().numpy(),
             count=agent.norm.count.cpu().numpy(), mean=agent.norm.mean.cpu().numpy(),
             std=agent.norm.std.cpu().numpy(), w=agent.networks["main"].params.weights.cpu().numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_shared_observation_statistics(tmp_path):
    """the observation running statistics are shared between ranks (2*dim+1 doubles
    all-reduced per rollout).  Both ranks must hold the statistics of BOTH shards, and identical weights."""
    import torch.multiprocessing as mp
    mp.spawn(_norm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / ("norm_%d.npz" % r)) for r in (0, 1))
    for k in ("sum", "sq", "count", "mean", "std", "w"):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    allx = np.concatenate([r0["raw"], r1["raw"]]).astype(np.float64)
    assert not np.array_equal(r0["raw"], r1["raw"])
    n = len(allx) + 1e-2
    assert abs(float(r0["count"][0]) - n) < 1e-9
    np.testing.assert_allclose(r0["sum"], allx.sum(0), rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(r0["sq"], (allx ** 2).sum(0) + 1e-2, rtol=1e-12, atol=1e-10)
    mean = r0["sum"] / n
    std = np.sqrt(np.maximum((r0["sq"] - n * mean ** 2) / max(n - 1, 1), 1e-2))
    np.testing.assert_allclose(r0["mean"], mean, rtol=1e-13)
    np.testing.assert_allclose(r0["std"], std, rtol=1e-12)


def _off_policy_worker(rank, world, port, out_dir, name, graphs):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import random
    import torch
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.distributed import GradientSync
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters as EP)
    from coach_amd.memories.memory import MemoryGranularity
    dev = torch.device("cuda:0")
    dist = GradientSync(backend="gloo")
    if name == "dqn":
        from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
        ap = DQNAgentParameters()
        ap.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(40)
        env = SyntheticVectorEnvironment(EP("vector", 4, (4,), 2, episode_length=10, seed=5), dev, rank=rank)
        cls, nets = DQNAgent, ("main",)
    elif name == "td3":
        from coach_amd.agents.td3_agent import TD3Agent, TD3AgentParameters
        ap = TD3AgentParameters()
        env = SyntheticVectorEnvironment(EP("vector", 4, (17,), None, action_dim=6, episode_length=10, seed=5),
                                         dev, rank=rank)
        cls, nets = TD3Agent, ("actor", "critic")
    else:
        from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent, SoftActorCriticAgentParameters
        ap = SoftActorCriticAgentParameters()
        env = SyntheticVectorEnvironment(EP("vector", 4, (23,), None, action_dim=5, episode_length=10, seed=5),
                                         dev, rank=rank)
        cls, nets = SoftActorCriticAgent, ("policy", "v", "q")
    ap.seed = 3
    ap.memory.max_size = (MemoryGranularity.Transitions, 4096)
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
    from coach_amd.core_types import RunPhase
    agent = cls(ap, env, dev, dist=dist, use_graphs=graphs)
    agent.phase = RunPhase.HEATUP
    for _ in range(80):
        agent.act()
    agent.phase = RunPhase.TRAIN
    for _ in range(12):
        agent.act()
        agent.train()
    agent.check_status()
    assert agent.training_iteration > 20
    w = np.concatenate([agent.networks[n].params.weights.cpu().numpy().ravel() for n in nets])
    np.save(os.path.join(out_dir, "%s_%d_%d.npy" % (name, int(graphs), rank)), w)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["dqn", "td3", "sac"])
def test_two_rank_off_policy_graph_segments(tmp_path, name):
    """Data-parallel off-policy updates: the update is captured as hipGraph segments cut at every
    gradient all-reduce.  Segmented replay == eager execution bit for bit; the ranks (different env
    shards, different replay draws) hold identical weights after every update without a broadcast."""
    import torch.multiprocessing as mp
    for graphs in (True, False):
        mp.spawn(_off_policy_worker, args=(2, _free_port(), str(tmp_path), name, graphs), nprocs=2, join=True)
    w = {(g, r): np.load(tmp_path / ("%s_%d_%d.npy" % (name, g, r))) for g in (0, 1) for r in (0, 1)}
    assert np.isfinite(w[(1, 0)]).all()
    np.testing.assert_array_equal(w[(1, 0)], w[(1, 1)])
    np.testing.assert_array_equal(w[(0, 0)], w[(0, 1)])
    np.testing.assert_array_equal(w[(1, 0)], w[(0, 0)])


@pytest.mark.timeout(600)
def test_rccl_path_world_size_one(tmp_path):
    """The real RCCL (backend "nccl") collectives at world size 1, where the sum is the identity: the all-reduce as a
    NODE of the captured epoch graph — one blocking bucket, or two overlapped buckets on RCCL's stream — must give
    exactly the weights of eager collectives between graph segments (stream ordering between torch's stream, the
    graphs and RCCL's stream)."""
    import torch.multiprocessing as mp
    for mode in ("1", "0", "s", "a"):                   # "a": the agent measures and chooses (checked in the worker)
        mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), mode, "nccl"), nprocs=1, join=True)
    w1, w0, ws, wa = (np.load(tmp_path / ("w_%s_0.npy" % m)) for m in "10sa")
    np.testing.assert_array_equal(wa, w1)               # the decision's dry forward / backward changes no weight
    assert np.isfinite(w0).all()
    np.testing.assert_array_equal(w0, w1)               # two overlapped buckets == one blocking all-reduce (both IN the graph)
    np.testing.assert_array_equal(ws, w1)               # collectives as graph nodes == eager collectives between segments


def test_split_backward_equals_full(dev):
    import torch
    from coach_amd.nn.networks import ClippedPPONet
    B, A, shape = 8, 5, (44, 44, 4)
    rng = np.random.RandomState(0)
    obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
    acts = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
    adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
    grads = []
    for split in (False, True):
        np.random.seed(1)
        net = ClippedPPONet(dev, shape, A, seed=2)
        net.update_target(1.0)
        old = net.policy_probs(obs, B, use_target=True, tag="old")
        net.forward_backward(obs, B, acts, adv, vt, old, stop_after_dense=split)
        if split:
            off = net.late_gradient_offset()
            assert 0 < off < net.params.size
            late = net.params.grads[off:].clone()
            net.backward_rest()
            assert torch.equal(late, net.params.grads[off:])   # the late bucket was already final
        grads.append(net.params.grads.clone())
    assert torch.equal(grads[0], grads[1])


# ------------------------------------------------- a data-parallel update against the ORACLE on the union of the shards
def _shard(kind, rank, shape, A, B):
    rng = np.random.RandomState(500 + rank)
    obs = rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)
    if kind == "ppo":
        return dict(obs=obs, actions=rng.randint(0, A, size=B).astype(np.int32), adv=rng.randn(B).astype(np.float32),
                    vt=rng.randn(B).astype(np.float32))
    return dict(obs=obs, next_obs=rng.randint(0, 256, size=(B,) + shape).astype(np.uint8),
                actions=rng.randint(0, A, size=B).astype(np.int32), rewards=rng.randn(B).astype(np.float32),
                game_overs=(rng.rand(B) < 0.2).astype(np.uint8))


def _oracle_worker(rank, world, port, out_dir, kind, scale_down):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from coach_amd.distributed import GradientSync
    from coach_amd.nn.networks import ClippedPPONet, DQNNet
    dev = torch.device("cuda:0")
    dist = GradientSync(backend="gloo")
    shape, A, B = (44, 44, 4), 4, 16
    d = {k: torch.from_numpy(v).to(dev) for k, v in _shard(kind, rank, shape, A, B).items()}
    scale = dist.grad_scale(scale_down)
    assert scale == (0.5 if scale_down else 1.0)
    np.random.seed(3)          # the VHead initializer draws from the GLOBAL numpy stream (heads/head.py:27-33)
    if kind == "ppo":
        net = ClippedPPONet(dev, shape, A, seed=3)
        net.update_target(1.0)
        old = net.policy_probs(d["obs"], B, use_target=True, tag="old")
        for _ in range(2):                                   # two synchronous steps: the second starts from shared weights
            net.forward_backward(d["obs"], B, d["actions"], d["adv"], d["vt"], old)
            dist.all_reduce_sum(net.params.grads)
            net.finish_update(scale)
    else:
        net = DQNNet(dev, shape, A, seed=3)
        for _ in range(2):
            net.learn_from_batch(d["obs"], d["next_obs"], B, d["actions"], d["rewards"], d["game_overs"], 0.99,
                                 grad_scale=scale, sync=dist)
    net.check_status()
    np.save(os.path.join(out_dir, "wo_%d.npy" % rank), net.params.weights.cpu().numpy())
    np.save(os.path.join(out_dir, "mo_%d.npy" % rank), net.adam.m.cpu().numpy())     # first moments: LINEAR in the scale
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind,scale_down", [("ppo", True), ("dqn", True), ("dqn", False)])
def test_two_rank_update_equals_oracle_on_the_union_of_the_shards(tmp_path, dev, kind, scale_down):
    """Two ranks, each with ITS OWN minibatch: gradients all-reduced (sum), scaled by 1 / num_workers where
    scale_down_gradients_by_number_of_workers_for_sync_training says so (architecture.py:485-488: on for DQN / Clipped
    PPO, off for DDPG / TD3), identical Adam step on every rank — against the oracle network that accumulates the
    gradient of both shards from the same weights, sums them and applies ONE step."""
    import torch
    import torch.multiprocessing as mp
    from coach_amd.nn.networks import ClippedPPONet, DQNNet
    from oracle.agents import ClippedPPOOracle, DQNOracle
    mp.spawn(_oracle_worker, args=(2, _free_port(), str(tmp_path), kind, scale_down), nprocs=2, join=True)
    w = [np.load(tmp_path / ("wo_%d.npy" % r)) for r in (0, 1)]
    np.testing.assert_array_equal(w[0], w[1])
    shape, A, B = (44, 44, 4), 4, 16
    shards = [_shard(kind, r, shape, A, B) for r in (0, 1)]
    np.random.seed(3)
    if kind == "ppo":
        net = ClippedPPONet(dev, shape, A, seed=3)            # same seeds -> the workers' initial weights
        o = ClippedPPOOracle(net.params.named_arrays(), shape, A)
        frozen = o.clone_policy()
        old = [o.policy_probs(s["obs"], frozen) for s in shards]
        for _ in range(2):
            grads = []
            for s, op in zip(shards, old):
                o.train_minibatch(s["obs"], s["actions"], s["adv"], s["vt"], op, apply=False)
                grads.append(o.snapshot_grads())
            o.apply_summed_gradients(grads, 2, scale_down)
    else:
        net = DQNNet(dev, shape, A, seed=3)
        o = DQNOracle(net.params.named_arrays(), shape, A)
        for _ in range(2):
            grads = []
            for s in shards:
                o.learn_from_batch(s["obs"], s["next_obs"], s["actions"], s["rewards"], s["game_overs"].astype(bool),
                                   0.99, apply=False)
                grads.append(o.snapshot_grads())
            o.apply_summed_gradients(grads, 2, scale_down)
    net.params.weights.copy_(torch.from_numpy(w[0]).to(dev))
    hw = net.params.named_arrays()
    for name, per_tower in o.weights().items():
        for t, ref in per_tower.items():
            np.testing.assert_allclose(hw[name][t], ref, rtol=2e-3, atol=5e-5, err_msg=name)
    # Adam's step is nearly invariant to a gradient scale; its first moment is linear in it: the 1 / num_workers rule
    # (or its absence) shows there
    m = net.params.named_arrays(torch.from_numpy(np.load(tmp_path / "mo_0.npy")).to(dev))
    head = "main/ppo_head/policy_fc" if kind == "ppo" else "main/q_head/dense"
    key = [k for k in o.adam.slots if k[0] == head and k[2] == "k"][0]
    om = o.adam.slots[key].m.reshape(m[head + "/kernel"][0].shape)
    np.testing.assert_allclose(m[head + "/kernel"][0], om, rtol=2e-3, atol=1e-7)
    assert np.abs(om).max() > 1e-6
