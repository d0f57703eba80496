"""Host-side bookkeeping of the on-policy rollout buffer for envs that end their episodes on different steps
(DeviceEpisodicRolloutBuffer.note_episode_ends / dataset_rows): pure host logic, runs without a GPU."""
import numpy as np
import torch


def test_complete_episode_dataset_lists_episodes_in_completion_order():
    from coach_amd.memories.episodic.episodic_rollout_buffer import DeviceEpisodicRolloutBuffer
    n_env, T = 3, 12
    buf = DeviceEpisodicRolloutBuffer(torch.device("cpu"), n_env, T, obs_dim=2)
    lengths = [3, 5, 4]                                   # per-env time limits
    t = np.zeros(n_env, dtype=int)
    expect = []                                           # (env, first step, end step) as store_episode calls would arrive
    start = [0] * n_env
    for step in range(1, 11):
        buf.steps = step                                  # the step's rows are stored, then its episode ends are noted
        t += 1
        dones = t >= np.array(lengths)
        buf.note_episode_ends(dones)
        for e in np.nonzero(dones)[0]:
            expect.append((int(e), start[e], step))
            start[e] = step
        t[dones] = 0
    assert buf.ragged and buf._episodes == expect
    n = buf.num_transitions()
    assert n == sum(b - a for _, a, b in expect) == buf.num_transitions_in_complete_episodes()
    assert buf.length() == len(expect)
    rows = buf.dataset_rows().numpy()
    want = np.concatenate([np.arange(a, b) * n_env + e for e, a, b in expect])
    np.testing.assert_array_equal(rows[:n], want)
    assert (rows[n:] == 0).all()
    # open tails are not part of the dataset: env 1 (limit 5) finished 2 episodes in 10 steps, env 0 three, env 2 two
    per_env = {e: sum(b - a for ee, a, b in expect if ee == e) for e in range(n_env)}
    assert per_env == {0: 9, 1: 10, 2: 8}
    # the row list is one static buffer: a later phase rewrites it in place
    ptr = buf.dataset_rows().data_ptr()
    buf.clean()
    assert buf.num_transitions() == 0 and buf._episodes == [] and buf.steps == 0
    buf.steps = 4
    buf.note_episode_ends(np.array([True, False, True]))
    assert buf.dataset_rows().data_ptr() == ptr
    np.testing.assert_array_equal(buf.dataset_rows().numpy()[:8], np.concatenate([np.arange(4) * 3 + 0, np.arange(4) * 3 + 2]))


def test_lockstep_dataset_is_env_major_and_unchanged():
    from coach_amd.memories.episodic.episodic_rollout_buffer import DeviceEpisodicRolloutBuffer
    buf = DeviceEpisodicRolloutBuffer(torch.device("cpu"), 2, 6, obs_dim=1)
    buf.steps = 3
    assert not buf.ragged and buf.num_transitions() == 6
    np.testing.assert_array_equal(buf.dataset_rows().numpy(), [0, 2, 4, 1, 3, 5])
