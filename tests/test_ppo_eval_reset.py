"""Clipped PPO on IMAGE observations through evaluation periods and forced resets — the headline configuration's
schedule (rl_coach/graph_managers/graph_manager.py:411-424 reset_internal_state, :463-485 train_and_act, :491-523
evaluate; agents/clipped_ppo_agent.py:346-350 inference-time filter).

A period that ends in the middle of an episode abandons that episode (it lived in current_episode_buffer, agent.py:619):
its transitions never reach the memory, its frames leave the stacking filter, and the np.random draws the abandoned steps
consumed stay consumed.  The device agent rewinds its lockstep rollout buffer and frame ring and must then sample the
same actions, train at the same steps on the same data and evaluate to the same rewards as the oracle agent (whose loop
is pinned to the real reference agent's, tests/test_update_pins.py)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tolerances import LOSS, WEIGHTS


def _make(dev, kind, n_env, L, playing, batch, epochs, seed=0):
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    shape = (84, 84) if kind == "image" else (7,)
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters(kind, n_env, shape, 6, episode_length=L, seed=99), dev)
    ap = ClippedPPOAgentParameters()
    ap.seed = seed
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(playing)
    ap.algorithm.optimization_epochs = epochs
    ap.network_wrappers["main"].batch_size = batch
    return ClippedPPOAgent(ap, env, dev)


def _rollout_snapshot(agent):
    """every row of the rollout's dataset as the training phase would read it (stacked states included)."""
    import torch
    mem = agent.memory
    n = mem.num_transitions()
    if n == 0:
        return {"n": 0}
    rows = mem.dataset_rows()
    out = torch.empty((n,) + tuple(mem.cur_state.shape[1:]), dtype=mem.cur_state.dtype, device=agent.device)
    mem.gather_states(rows[:n], n, out)
    r = rows[:n].long()
    return {"n": n, "state": out.cpu().numpy(), "action": mem.action[r].cpu().numpy(), "reward": mem.reward[r].cpu().numpy(),
            "game_over": mem.game_over[r].cpu().numpy()}


@pytest.mark.parametrize("kind", ["image", "vector"])
def test_periods_that_end_mid_episode_and_evaluations_match_the_oracle(rlx, dev, kind):
    """Three train_and_act periods of 10 vector steps with L = 6 (no period ends on an episode boundary), an
    evaluation of one episode per env after each: actions, training steps, evaluation rewards, losses and weights
    against the oracle; the rollout (rows AND stacked frames) is untouched by the evaluation."""
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    n_env, L, playing, batch, epochs, period = 4, 6, 40, 8, 2, 10
    agent = _make(dev, kind, n_env, L, playing, batch, epochs)
    assert not agent.ragged and agent.steps_per_phase == 12
    arrays = agent.networks["main"].params.named_arrays()
    elems = 84 * 84 if kind == "image" else 7
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(0 if kind == "image" else 1, n_env, elems, L, 99), 6, batch_size=batch,
                              playing_steps=playing, epochs=epochs)
    o.reset((84, 84) if kind == "image" else None)
    start = (random.getstate(), np.random.get_state())

    # ---- device: the whole schedule first (its np.random draws are made a phase at a time)
    dev_actions, dev_train_at, dev_losses, dev_evals, dev_weights = [], [], [], [], []
    t = 0
    for p in range(3):
        agent.reset_internal_state()                          # train_and_act: forced reset at the period start (:477)
        for _ in range(period):
            agent.act()
            dev_actions.append(agent.actions.cpu().numpy().copy())
            res = agent.train()
            if res is not None:
                dev_train_at.append(t)
                dev_losses.append(np.array([r.cpu().numpy()[:5] for r in res], dtype=np.float64))
                dev_weights.append(agent.networks["main"].params.named_arrays())
            t += 1
        assert agent.memory.steps % L != 0                    # the period ended inside an episode
        open_steps = agent.memory.steps % L
        complete = agent.memory.steps - open_steps
        before_counters = (agent.total_steps_counter, agent.training_iteration, agent.last_training_phase_step)
        dev_evals.append(agent.evaluate_episodes(1))
        # evaluation abandoned the open episode (as the forced reset of the next period would) and stored nothing
        assert agent.memory.steps == complete
        snap_a = _rollout_snapshot(agent)
        agent.evaluate_episodes(2)                            # a second evaluation: twice as many scratch frames
        snap_b = _rollout_snapshot(agent)
        assert snap_a["n"] == snap_b["n"] == complete * n_env
        for k in snap_a:
            np.testing.assert_array_equal(snap_a[k], snap_b[k], err_msg=k)
        assert (agent.total_steps_counter, agent.training_iteration, agent.last_training_phase_step) == before_counters
        assert float(agent.ep_return.abs().sum().item()) == 0.0 and int(agent.ep_len.sum().item()) == 0
    agent.check_status()
    dev_random_state = random.getstate()
    assert len(dev_train_at) >= 2

    # ---- oracle: same host streams from the same starting point, draws made one call at a time
    random.setstate(start[0]); np.random.set_state(start[1])
    t, total, last_train, o_train_at, k = 0, 0, 0, [], 0
    for p in range(3):
        o.forced_reset()
        for _ in range(period):
            oa, _ = o.act()
            assert np.array_equal(np.asarray(oa), dev_actions[t]), "sampled action differs at step %d" % t
            total += n_env
            held = len(o.transitions[0])
            if total - last_train >= playing and held > 0 and held % L == 0:      # agent.py:681-699, full episodes
                last_train = total
                ores = np.array(o.train())
                o_train_at.append(t)
                np.testing.assert_allclose(dev_losses[k], ores, **LOSS)
                for name, per_tower in o.net.weights().items():
                    for tw, ref in per_tower.items():
                        np.testing.assert_allclose(dev_weights[k][name][tw], ref, err_msg=name, **WEIGHTS)
                k += 1
            t += 1
        ev = o.evaluate(1)
        assert abs(ev - dev_evals[p]) < 1e-9, (p, ev, dev_evals[p])
        o.evaluate(2)
    assert o_train_at == dev_train_at
    assert random.getstate() == dev_random_state              # the same number of shuffles


def test_atari_clipped_ppo_preset_improves_with_evaluation_periods(rlx, dev, tmp_path):
    """presets/Atari_ClippedPPO.py (a real evaluation period) through BasicRLGraphManager.improve() on a small vector,
    with a period that is NOT a whole number of rollouts: every period starts from a forced reset, every evaluation
    leaves the rollout buffer and the frame ring as they were, and the run equals an eager (no hipGraph) run bit for
    bit."""
    import importlib
    from coach_amd.core_types import EnvironmentEpisodes, EnvironmentSteps, RunPhase
    weights, evals = [], []
    for graphs in (True, False):
        random.seed(3); np.random.seed(3)
        preset = importlib.reload(importlib.import_module("coach_amd.presets.Atari_ClippedPPO"))
        assert preset.schedule_params.evaluation_steps.num_steps > 0          # the shipped preset evaluates
        gm = preset.make(num_envs=4, episode_length=8, playing_steps=64, batch_size=16, optimization_epochs=2,
                         improve_steps=3 * 80, steps_between_evaluation_periods=80)
        gm.device, gm.use_graphs = dev, graphs
        gm.logger.__init__(str(tmp_path / ("exp%d.csv" % graphs)))
        gm.create_graph()
        agent = gm.agent
        seen = []
        orig = agent.evaluate_episodes

        def spy(episodes, agent=agent, orig=orig, seen=seen):
            agent.memory.drop_open_episodes(agent.L)          # what the evaluation's own forced reset keeps
            before = _rollout_snapshot(agent)
            r = orig(episodes)
            after = _rollout_snapshot(agent)
            assert before["n"] == after["n"]
            for k in before:
                np.testing.assert_array_equal(before[k], after[k], err_msg=k)
            seen.append(before["n"])
            return r
        agent.evaluate_episodes = spy
        rows = gm.improve()
        # 80 env-steps = 20 vector steps = 2.5 episodes per period: every period ends mid-episode
        assert gm.total_steps_counters[RunPhase.TRAIN] == 240
        assert gm.total_steps_counters[RunPhase.TEST] == 3 * 4 * 8            # one episode per env per evaluation
        assert len(seen) == 3
        eval_rows = [r for r in rows if r["Evaluation Reward"] != ""]
        assert len(eval_rows) == 3
        assert agent.training_iteration >= 2
        agent.check_status()
        weights.append(agent.networks["main"].params.weights.cpu().numpy().copy())
        evals.append([float(r["Evaluation Reward"]) for r in eval_rows])
    np.testing.assert_array_equal(weights[0], weights[1])
    assert evals[0] == evals[1]
