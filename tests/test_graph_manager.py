"""Schedule driver: the reference's heatup -> (train_and_act -> evaluate)* loop
(graph_manager.py:400-556) on the CartPole_DQN preset with the synthetic env."""
import pytest


def test_schedule_and_step_units_cpu():
    from coach_amd.core_types import EnvironmentSteps, TrainingSteps
    from coach_amd.graph_managers.basic_rl_graph_manager import ScheduleParameters, dynamic_import
    sp = ScheduleParameters()
    assert isinstance(sp.improve_steps, TrainingSteps) and sp.heatup_steps == EnvironmentSteps(0)
    cls = dynamic_import("coach_amd.core_types:EnvironmentSteps")
    assert cls is EnvironmentSteps
    assert EnvironmentSteps(10) / EnvironmentSteps(4) == 3                     # core_types.py:75-83 (ceil)


def test_csv_columns_are_reference_column_names_cpu():
    """The fixed experiment-CSV columns, in Agent.update_log's order (agents/agent.py:520-546); the full list incl. the
    per-signal statistics is pinned to the real reference agent in tests/test_signals_csv.py."""
    from coach_amd.graph_managers.basic_rl_graph_manager import CsvLogger
    lg = CsvLogger(None, ["Loss"])
    assert lg.COLUMNS[0] == "Episode #" and lg.COLUMNS[1:4] == ["Training Iter", "Epoch", "In Heatup"]
    assert lg.COLUMNS[-4:] == ["Loss/Mean", "Loss/Stdev", "Loss/Max", "Loss/Min"]


@pytest.mark.gpu
def test_cartpole_dqn_preset_improve(dev, tmp_path):
    import importlib
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    preset = importlib.reload(importlib.import_module("coach_amd.presets.CartPole_DQN"))
    gm = preset.make(synthetic=True)
    gm.device = dev
    gm.logger.__init__(str(tmp_path / "exp.csv"))
    gm.schedule.heatup_steps = EnvironmentSteps(400)
    gm.schedule.improve_steps = EnvironmentSteps(1200)
    gm.schedule.steps_between_evaluation_periods = EnvironmentSteps(600)
    rows = gm.improve()
    assert gm.total_steps_counters[RunPhase.HEATUP] == 400
    assert gm.total_steps_counters[RunPhase.TRAIN] == 1200
    assert gm.total_steps_counters[RunPhase.TEST] >= 2 * 200                 # two evaluation periods
    assert gm.agent.training_iteration == 1200                                # 1 update per TRAIN env-step
    assert gm.agent.memory.num_transitions() == 1600                          # evaluation is not stored
    train_rows = [r for r in rows if r["Training Reward"] != ""]
    eval_rows = [r for r in rows if r["Evaluation Reward"] != ""]
    assert len(train_rows) == 8 and len(eval_rows) == 2
    assert [r["Episode #"] for r in train_rows] == list(range(1, 9))
    assert all(r["Episode Length"] == 200 for r in train_rows)
    assert (tmp_path / "exp.csv").read_text().splitlines()[0].startswith("Episode #,Training Iter")


@pytest.mark.gpu
def test_atari_dueling_ddqn_preset_improve(dev, tmp_path):
    """presets/Atari_Dueling_DDQN.py end to end on a small vector: dueling head, Empty middleware,
    gradient clipping and the head-gradient rescale inside the captured update; graph replay equals
    eager execution bit for bit."""
    import random
    import numpy as np
    import torch
    from coach_amd.core_types import RunPhase
    from coach_amd.presets import Atari_Dueling_DDQN as preset
    weights = []
    for graphs in ("0", "1"):
        random.seed(2); np.random.seed(2)
        gm = preset.make(num_envs=8, replay_transitions=2048, heatup_steps=256, improve_steps=256,
                         episode_length=16)
        gm.device = dev
        gm.use_graphs = graphs == "1"
        gm.logger.__init__(str(tmp_path / ("exp%s.csv" % graphs)))
        gm.improve()
        net = gm.agent.networks["main"]
        assert net.dueling and net.clip_gradients == 10 and abs(net.head_gradient_rescale - 2 ** -0.5) < 1e-12
        assert gm.total_steps_counters[RunPhase.TRAIN] == 256
        assert gm.agent.training_iteration == 256 // 4                        # one update per 4 env-steps
        assert bool(gm.agent.use_graphs) == (graphs == "1")
        net.check_status()
        w = net.params.weights.cpu().numpy()
        assert np.isfinite(w).all()
        weights.append(w)
    np.testing.assert_array_equal(weights[0], weights[1])
