"""PrioritizedExperienceReplay.update_priorities (prioritized_experience_replay.py:203-217) riding as one more workgroup on
the backward pass's fused input-gradient launch (rlx_conv32_input_grad_per_update, coach_amd/csrc/conv_bwd_fused.hip; or
on its deferred split-K reduction launch, rlx_splitk_reduce_jobs_per_update, where that launch is not taken) against the
same update as a launch of its own behind learn_from_batch (agents/dqn_agent.py:106-109): the three trees, the sampled
leaves of every update and the weights must be bit-identical — it is the same device code (per_update_body.hpp) on the
same inputs, dispatched earlier."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ENV, FRAME, A, B, CAP, EP_LEN = 8, (84, 84), 4, 32, 256, 50


def _run(dev, rides, zero_copy=True, one_launch=True):
    import torch
    from coach_amd import _rlx
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    from coach_amd.memories.non_episodic.experience_replay import ExperienceReplay
    from coach_amd.memories.non_episodic.prioritized_experience_replay import (
        PrioritizedExperienceReplay, PrioritizedExperienceReplayParameters)
    saved = (DQNAgent.PER_UPDATE_RIDES, PrioritizedExperienceReplay.ZERO_COPY_DRAWS, ExperienceReplay.GATHER_ONE_LAUNCH)
    DQNAgent.PER_UPDATE_RIDES, PrioritizedExperienceReplay.ZERO_COPY_DRAWS, ExperienceReplay.GATHER_ONE_LAUNCH = \
        rides, zero_copy, one_launch
    try:
        random.seed(3)
        np.random.seed(3)
        env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("image", N_ENV, FRAME, A, episode_length=EP_LEN,
                                                                              seed=99), dev)
        ap = DQNAgentParameters()
        ap.seed = 0
        ap.memory = PrioritizedExperienceReplayParameters()
        ap.memory.max_size = (MemoryGranularity.Transitions, CAP)
        ap.network_wrappers["main"].batch_size = B
        agent = DQNAgent(ap, env, dev)
        agent.debug_draws = []
        agent.phase = RunPhase.HEATUP
        for _ in range(CAP // N_ENV + 3):
            agent.act()
        agent.phase = RunPhase.TRAIN
        names = []
        for step in range(6):                           # 6 vector steps x 2 updates
            agent.act()
            if step == 4:
                with _rlx.KernelTimer(256) as timer:
                    agent.train()
                names = [n for n, _ in timer.records]
            else:
                agent.train()
        agent.check_status()
        mem = agent.memory
        torch.cuda.synchronize()
        return ([t.cpu().numpy().copy() for t in (mem.sum_tree, mem.min_tree, mem.max_tree)], mem.maximal_priority,
                [np.array(d) for d in agent.debug_draws], agent.networks["main"].params.weights.cpu().numpy().copy(), names)
    finally:
        DQNAgent.PER_UPDATE_RIDES, PrioritizedExperienceReplay.ZERO_COPY_DRAWS, ExperienceReplay.GATHER_ONE_LAUNCH = saved


def test_priority_update_on_the_deferred_reduction_launch_is_bit_identical(rlx, dev):
    own = _run(dev, rides=False)
    ride = _run(dev, rides=True)
    # launches outside the update's hipGraph, per update: sample, gather (+ the priority update: launch_update's kernel is
    # recorded under the name of its function-pointer variable, "kernel")
    assert [n for n in own[4] if n == "kernel"] == ["kernel", "kernel"], own[4]
    assert not any(n == "kernel" for n in ride[4]) and len(ride[4]) == len(own[4]) - 2, (ride[4], own[4])
    assert len(own[2]) == len(ride[2]) == 12
    for a, b in zip(own[2], ride[2]):
        np.testing.assert_array_equal(a, b)             # the sampled leaves of every update
    for a, b in zip(own[0], ride[0]):
        np.testing.assert_array_equal(a, b)             # sum / min / max trees
    assert own[1] == ride[1]
    np.testing.assert_array_equal(own[3], ride[3])


def test_pinned_draws_and_columns_on_the_gather_launch_change_nothing(rlx, dev):
    """rlx_per_sample reading its uniforms from the pinned host slot (no blit) and the batch's small columns gathered by
    one more workgroup of the frame gather (rlx_imgreplay_gather_columns) against the blit + rlx_copy_columns path."""
    new = _run(dev, rides=True, zero_copy=True, one_launch=True)
    old = _run(dev, rides=True, zero_copy=False, one_launch=False)
    assert any("img_gather4_kernel<true>" in n for n in new[4]) and not any("copy_columns" in n for n in new[4]), new[4]
    assert any("copy_columns" in n for n in old[4]), old[4]
    for a, b in zip(old[2], new[2]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(old[0], new[0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(old[3], new[3])


@pytest.mark.parametrize("where", ["conv32", "reduce"])
def test_rider_launches_leave_the_trees_of_the_stand_alone_update(rlx, dev, where):
    """library level: rlx_conv32_input_grad_per_update / rlx_splitk_reduce_jobs_per_update against rlx_per_update on copies
    of the same 2^12-leaf trees, 32 leaves with a duplicate; the launch's own results against the launch without a rider."""
    import ctypes
    import torch
    from coach_amd import _rlx
    cap, n = 1 << 12, 32
    rng = np.random.RandomState(5)
    s_ = _rlx.current_stream()

    def trees():
        t = [torch.empty(2 * cap - 1, dtype=torch.float64, device=dev) for _ in range(3)]
        maxp = torch.zeros(1, dtype=torch.float64, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        rlx.per_init(t[0], t[1], t[2], cap, maxp, s_)
        rlx.per_store(t[0], t[1], t[2], cap, 0, 64, 0.6, maxp, st, s_)
        for i in range(0, cap, 256):
            rlx.per_update(t[0], t[1], t[2], cap, torch.arange(i, i + 256, dtype=torch.int32, device=dev),
                           torch.from_numpy(np.abs(np.random.RandomState(i).randn(256)) + 0.01).to(dev), 256, 0.6, 1e-6, maxp,
                           st, s_)
        return t, maxp, st
    idx_np = rng.randint(0, cap, size=n).astype(np.int32)
    idx_np[7] = idx_np[3]                                   # the last occurrence wins (:214-215)
    idx = torch.from_numpy(idx_np).to(dev)
    err = torch.from_numpy(np.abs(rng.randn(n))).to(dev)
    (ta, ma, sa), (tb, mb, sb) = trees(), trees()
    rlx.per_update(ta[0], ta[1], ta[2], cap, idx, err, n, 0.6, 1e-6, ma, sa, s_)
    args = (tb[0], tb[1], tb[2], cap, idx, err, n, 0.6, 1e-6, mb, sb)
    if where == "conv32":
        B, T = 6, 1
        g = lambda *shape: torch.from_numpy(rng.randn(*shape).astype(np.float32)).to(dev)
        dz3, w3, y2, w2, y1 = g(T, B * 49, 64), g(T, 576, 64), g(T, B * 81, 64), g(T, 512, 64), g(T, B * 400, 32)
        outs = []
        for per in (None, _rlx.per_update_desc(args)):
            dz2, dz1 = torch.zeros(T, B * 81, 64, device=dev), torch.zeros(T, B * 400, 32, device=dev)
            a = (dz3, B * 49 * 64, w3, 576 * 64, y2, B * 81 * 64, dz2, B * 81 * 64, w2, 512 * 64, y1, B * 400 * 32, dz1,
                 B * 400 * 32, B, T, _rlx.ACT["relu"])
            if per is None:
                rlx.conv32_input_grad(*a, s_)
            else:
                rlx.conv32_input_grad_per_update(*a, ctypes.byref(per), s_)
            outs.append((dz2, dz1))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert float(outs[0][1].abs().max()) > 0
    else:
        K, N, splits = 64, 32, 4
        part = torch.from_numpy(rng.randn(splits, K, N).astype(np.float32)).to(dev)
        out = torch.zeros(K, N, device=dev)
        job = _rlx.SplitkJob()
        job.partials, job.C, job.ldc, job.M, job.N, job.batch, job.splits = part.data_ptr(), out.data_ptr(), N, K, N, 1, splits
        _rlx.splitk_reduce_jobs([job], s_, per_tail=args)
        np.testing.assert_allclose(out.cpu().numpy(), part.cpu().numpy().astype(np.float64).sum(0), rtol=1e-6, atol=1e-6)
    torch.cuda.synchronize()
    for x, y in zip(ta, tb):
        assert torch.equal(x, y)
    assert float(ma.item()) == float(mb.item()) and int(sa.item()) == int(sb.item()) == 0
