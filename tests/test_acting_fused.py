"""An acting step's small launches merged (ClippedPPOAgent.FUSE_ACTING_LAUNCHES): softmax + categorical draw as one launch
(rlx_softmax_categorical_sample; heads/ppo_head.py:108 + exploration_policies/categorical.py:45-48) and reward filter +
episode totals + the step's action / reward / game_over columns as one launch (rlx_rollout_observe_step; agents/agent.py:905-973
+ the episodic memory's store).  Same per-element arithmetic as the launches they replace: a whole rollout — actions, stored
columns, episode statistics — and the iteration trained on it must be BIT-IDENTICAL to the unmerged form (which the other
tests compare with the oracle; they now run through the merged form)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(dev, fused, kind, n_env, L, playing, batch):
    import torch
    from coach_amd import _rlx
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent
    from test_ppo_agent import _make
    ClippedPPOAgent.FUSE_ACTING_LAUNCHES = fused
    try:
        random.seed(3); np.random.seed(3)
        agent = _make(dev, n_env, L, playing, batch, 1, seed=0, kind=kind)
        actions, res = [], None
        while res is None:
            agent.act()
            actions.append(agent.actions.clone())
            res = agent.train()
        m = agent.memory
        out = dict(actions=torch.stack(actions), action_col=m.action.clone(), reward_col=m.reward.clone(),
                   done_col=m.game_over.clone(), ep_return=agent.ep_return.clone(), ep_len=agent.ep_len.clone(),
                   acc=agent.ep_acc.clone(), weights=agent.networks["main"].params.weights.clone(),
                   losses=torch.stack([r.clone() for r in res]))
        agent.networks["main"].check_status()
        assert int(m.status.item()) == 0
        return out
    finally:
        ClippedPPOAgent.FUSE_ACTING_LAUNCHES = True


@pytest.mark.parametrize("kind,n_env,L,playing,batch", [("image", 8, 5, 40, 8), ("vector", 6, 7, 42, 6), ("image", 64, 4, 256, 64)])
def test_merged_acting_launches_are_bit_identical(rlx, dev, kind, n_env, L, playing, batch):
    import torch
    a = _run(dev, False, kind, n_env, L, playing, batch)
    b = _run(dev, True, kind, n_env, L, playing, batch)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a["done_col"].sum()) > 0 and float(a["reward_col"].abs().max()) > 0


def test_the_two_merged_launches_run(rlx, dev):
    from coach_amd import _rlx
    from test_ppo_agent import _make
    agent = _make(dev, 8, 5, 40, 8, 1, seed=0)
    agent.use_graphs = False
    agent.act()
    with _rlx.KernelTimer(256) as timer:
        agent.act()
    names = [n for n, _ in timer.records]
    assert sum("softmax_categorical_sample_kernel" in n for n in names) == 1, names
    assert sum("rollout_observe_kernel" in n for n in names) == 1, names
    assert not any(n.startswith(("softmax_kernel", "categorical_sample_kernel", "reward_filter_kernel", "episode_stats_kernel"))
                   for n in names), names


def test_the_sampling_launch_alone(rlx, dev):
    """random logits: the actions of the merged launch = rlx_softmax + rlx_categorical_sample, its optional probabilities =
    rlx_softmax's, bit for bit"""
    import torch
    from coach_amd import _rlx
    n, A = 300, 6
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(n, A, generator=g) * 3).to(dev)
    u = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
    s = _rlx.current_stream()
    probs = torch.empty(n, A, device=dev)
    a0 = torch.empty(n, dtype=torch.int32, device=dev)
    rlx.softmax(logits, A, n, A, probs, A, s)
    rlx.categorical_sample(probs, A, u, n, A, a0, s)
    p1 = torch.empty(n, A, device=dev)
    a1 = torch.empty(n, dtype=torch.int32, device=dev)
    rlx.softmax_categorical_sample(logits, A, u, n, A, p1, A, a1, s)
    a2 = torch.empty(n, dtype=torch.int32, device=dev)
    rlx.softmax_categorical_sample(logits, A, u, n, A, None, 0, a2, s)
    assert torch.equal(a0, a1) and torch.equal(a0, a2) and torch.equal(probs, p1)
    assert len(set(a0.tolist())) == A
