"""VectorOffPolicyAgent.HOST_DRAWS_AHEAD: the host-RNG side of Agent.train (agent.py:701-770 — per phase the index draws of
every batch, :726, then each update's own draws: TD3's smoothing noise td3_agent.py:162, SAC's normals
soft_actor_critic_agent.py:190-230) on a producer thread, against the training loop making the same draws itself: the same
values in the same order — weights, targets and the state of both host generators afterwards are bit-identical."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,L", [("td3", 10), ("sac", 5), ("ddpg", 5)])
def test_producer_thread_changes_nothing(dev, name, L):
    import torch
    from test_agent_loops import _drive, _mk
    from coach_amd.agents.vector_agent import VectorOffPolicyAgent
    res = {}
    saved = VectorOffPolicyAgent.HOST_DRAWS_AHEAD
    try:
        for ahead in (False, True):
            VectorOffPolicyAgent.HOST_DRAWS_AHEAD = ahead
            a = _mk(dev, name, True, n_env=8, L=L)
            made = []
            if ahead:
                import coach_amd.agents.vector_agent as V
                orig = V._HostDrawsAhead.__init__

                def counting(self, items, n, depth=32, _o=orig):
                    made.append(n)
                    _o(self, items, n, depth)
                V._HostDrawsAhead.__init__ = counting
            try:
                ups = _drive(a, 5, 22)
            finally:
                if ahead:
                    V._HostDrawsAhead.__init__ = orig
            res[ahead] = (ups, {k: a.networks[k].params.weights.clone() for k in a.networks},
                          {k: a.networks[k].target.clone() for k in a.networks if a.networks[k].target is not None},
                          np.random.get_state()[1].copy(), random.getstate(), made)
    finally:
        VectorOffPolicyAgent.HOST_DRAWS_AHEAD = saved
    off, on = res[False], res[True]
    assert sum(off[0]) > 0 and off[0] == on[0]
    assert on[5] and sum(on[5]) == sum(on[0]), (on[5], on[0])
    for k in off[1]:
        assert torch.equal(off[1][k], on[1][k]), k
    for k in off[2]:
        assert torch.equal(off[2][k], on[2][k]), k
    assert np.array_equal(off[3], on[3]) and off[4] == on[4]


@pytest.mark.parametrize("name,L", [("td3", 10), ("td3", 7), ("sac", 5), ("ddpg", 5), ("ddpg_bn", 5)])
def test_chunks_of_updates_in_one_graph_change_nothing(dev, name, L):
    """VectorOffPolicyAgent.UPDATE_CHUNK: K consecutive updates of a train() call as one captured graph over one staged record
    (one blit, one replay) against one record, one gather and one or two replays per update: weights, targets and both host
    generators bit-identical; TD3 at L = 7: 56 updates per call."""
    import torch
    from test_agent_loops import _drive, _mk
    from coach_amd.agents.vector_agent import VectorOffPolicyAgent
    res = {}
    saved = VectorOffPolicyAgent.UPDATE_CHUNK
    try:
        for chunk in (1, 8):
            VectorOffPolicyAgent.UPDATE_CHUNK = chunk
            a = _mk(dev, name, True, n_env=8, L=L)
            ups = _drive(a, 5, 30)
            keys = [k for k in a._graphs if k and k[0] == "chunk"]
            res[chunk] = (ups, {k: a.networks[k].params.weights.clone() for k in a.networks},
                          {k: a.networks[k].target.clone() for k in a.networks if a.networks[k].target is not None},
                          np.random.get_state()[1].copy(), random.getstate(), keys, a.training_iteration)
    finally:
        VectorOffPolicyAgent.UPDATE_CHUNK = saved
    one, eight = res[1], res[8]
    assert sum(one[0]) > 16 and one[0] == eight[0] and one[6] == eight[6]
    assert not one[5] and eight[5], (one[5], eight[5])
    if name == "td3":
        # the delayed actor step and the target mixes are part of a chunk's key: every second update mixes (:186-209)
        assert all(k[1] == 8 and len(k[2]) == 8 and any(k[2]) and not all(k[2]) for k in eight[5]), eight[5]
    for k in one[1]:
        assert torch.equal(one[1][k], eight[1][k]), k
    for k in one[2]:
        assert torch.equal(one[2][k], eight[2][k]), k
    assert np.array_equal(one[3], eight[3]) and one[4] == eight[4]
