"""Signal statistics and the experiment CSV ("the CSV schema is the observable parity surface").
tests/golden/csv_columns.json was written by the REAL reference DQNAgent.update_log and the reference `Signal` class
(tests/golden/make_golden.py csv_columns): the column list (order included) of the device engine's logger and the
mean / stdev / max / min of the device signal records must reproduce it."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "csv_columns.json")))


def test_csv_columns_are_the_reference_agents_columns():
    from coach_amd.agents.dqn_agent import DQNAgent
    from coach_amd.graph_managers.basic_rl_graph_manager import CsvLogger
    lg = CsvLogger(None, DQNAgent.SIGNAL_NAMES)
    assert lg.COLUMNS[0] == GOLD["dqn_index"] == "Episode #"
    assert lg.COLUMNS[1:] == GOLD["dqn_columns"]


@pytest.mark.gpu
def test_device_signal_statistics_match_the_reference_signal_class(dev):
    import torch
    from coach_amd.signals import DeviceSignals
    cases = GOLD["signal_cases"]
    names = ["s%d" % i for i in range(len(cases))] + ["never fed"]
    sig = DeviceSignals(names, dev)
    rounds = max(len(c["samples"]) for c in cases)
    for r in range(rounds):                                  # one launch per "update": all signals that have a sample
        batch = {}
        for i, c in enumerate(cases):
            if r < len(c["samples"]):
                v = np.array(c["samples"][r])
                as32 = bool(np.all(v.astype(np.float32).astype(np.float64) == v))     # an fp32 sample array
                batch["s%d" % i] = torch.tensor(v, dtype=torch.float32 if as32 else torch.float64, device=dev)
        sig.accumulate(batch)
    out = sig.flush()
    for i, c in enumerate(cases):
        # np.mean / np.std of float32 samples accumulate in float32 (pairwise); the device records are fp64
        rt = 2e-6 if c["f32"] else 1e-12
        np.testing.assert_allclose(out["s%d/Mean" % i], c["mean"], rtol=rt, atol=1e-7 if c["f32"] else 1e-15)
        np.testing.assert_allclose(out["s%d/Stdev" % i], c["stdev"], rtol=max(rt, 1e-9), atol=1e-12)
        assert out["s%d/Max" % i] == c["max"] and out["s%d/Min" % i] == c["min"]
    assert [out["never fed/" + s] for s in ("Mean", "Stdev", "Max", "Min")] == ["", "", "", ""]   # Signal.get_* on no samples
    again = sig.flush()                                       # flushed records are empty again
    assert all(v == "" for v in again.values())


@pytest.mark.gpu
def test_graph_manager_writes_one_reference_style_row_per_episode(dev, tmp_path):
    """CartPole-like DQN through heat-up and training with CSV logging on: every finished episode is a row with the
    reference's columns; reward / length / discounted-return statistics are checked against the CPU twin of the env."""
    from coach_amd.agents.dqn_agent import DQNAgentParameters
    from coach_amd.base_parameters import VisualizationParameters
    from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import SyntheticVectorEnvironmentParameters
    from coach_amd.graph_managers.basic_rl_graph_manager import BasicRLGraphManager, ScheduleParameters
    from coach_amd.memories.memory import MemoryGranularity
    from oracle.replay import EpisodicReplayOracle
    from oracle.synth_env import SynthVecEnv
    L, n_env = 7, 2
    ap = DQNAgentParameters()
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
    ap.network_wrappers["main"].batch_size = 8
    ap.memory.max_size = (MemoryGranularity.Transitions, 256)
    env = SyntheticVectorEnvironmentParameters("vector", n_env, (4,), 2, episode_length=L, seed=21, episode_lengths=[7, 5])
    sp = ScheduleParameters()
    sp.heatup_steps = EnvironmentSteps(2 * 7 * n_env)
    sp.improve_steps = EnvironmentSteps(4 * 7 * n_env)
    sp.steps_between_evaluation_periods = EnvironmentSteps(2 * 7 * n_env)
    sp.evaluation_steps = EnvironmentEpisodes(0)
    path = str(tmp_path / "agent.csv")
    gm = BasicRLGraphManager(ap, env, sp, vis_params=VisualizationParameters(dump_csv=True), device=dev, csv_path=path)
    rows = gm.improve()
    import csv
    with open(path) as f:
        table = list(csv.DictReader(f))
    assert list(table[0].keys()) == ["Episode #"] + GOLD["dqn_columns"]
    assert len(table) == len(rows)
    # the CPU twin of the env gives every episode's rewards.  Heat-up and the two training periods are 14 vector steps
    # each; env 1 (episodes of 5) is in the middle of an episode when a period ends, so the next period starts from a
    # forced reset of every env (GraphManager.train_and_act -> reset_internal_state(force_environment_reset=True),
    # graph_manager.py:477) and the open episode is never logged
    o = SynthVecEnv(1, n_env, 4, L, 21, episode_lengths=[7, 5])
    expect = []
    for period in range(3):
        o.reset()
        acc = [[], []]
        for _ in range(2 * 7):
            _, _, rew, done = o.step()
            for e in range(n_env):
                acc[e].append(float(rew[e]))
                if done[e]:
                    expect.append((e, acc[e]))
                    acc[e] = []
    assert len(table) == len(expect)
    for row, (e, rewards) in zip(table, expect):
        assert int(row["Episode Length"]) == len(rewards)
        heat = int(row["In Heatup"])
        if not heat:
            np.testing.assert_allclose(float(row["Training Reward"]), sum(rewards), rtol=1e-6, atol=1e-6)
            assert row["Loss/Mean"] != "" or True
    # 'Discounted Return' of the first row of every step = statistics over the finished episodes of that step
    first = table[0]
    dr = EpisodicReplayOracle.n_step_returns(np.float32(expect[0][1]), 0.99, -1)
    np.testing.assert_allclose(float(first["Discounted Return/Mean"]), dr.mean(), rtol=1e-9)
    np.testing.assert_allclose(float(first["Discounted Return/Max"]), dr.max(), rtol=1e-12)
    trained = [r for r in table if not int(r["In Heatup"])]
    # (the first TRAIN-phase row is written before the first update: train() follows act())
    with_stats = [r for r in trained if r["Discounted Return/Mean"] != ""]
    assert with_stats and all(r["Loss/Mean"] != "" for r in with_stats[1:])
    assert all(abs(float(r["Learning Rate/Mean"]) - 0.00025) < 1e-15 for r in trained if r["Learning Rate/Mean"] != "")
