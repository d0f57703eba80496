"""Host logic of tools/loss_curve_c2.py (CPU): the pairwise report of runs trained on an identical action history and the
ensemble statistic, on synthetic run files — what the numbers in DESIGN.md §6 (`profiles/r05_loss_curve_c2_identical_histories_10ep.json`,
`…_device_vs_itself_…json`) are computed with."""
import argparse
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("loss_curve_c2", os.path.join(ROOT, "tools", "loss_curve_c2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write(path, results, actions, own=None):
    np.savez_compressed(path, results=results, actions=actions, own_actions=np.zeros((0,)) if own is None else own,
                        seconds=1.0)


def test_pairs_of_runs_on_an_identical_history(tmp_path):
    T = _tool()
    rng = np.random.RandomState(0)
    it, ep = 14, 10
    steps = it * (T.PLAYING // T.N_ENV)
    for sd in (0, 1):
        d = tmp_path / ("seed%d" % sd)
        d.mkdir()
        acts = rng.randint(0, T.A, size=(steps, T.N_ENV)).astype(np.int8)
        base = 1.0 + rng.rand(it, ep, 5)
        _write(d / "hip_forced.npz", base, acts, own=acts.copy())
        _write(d / "oracle_forced.npz", base * 1.01, acts, own=acts.copy())         # 1 % apart everywhere
        own = acts.copy()
        own[0, :3] = (own[0, :3] + 1) % T.A                                           # three of its own samples differ
        _write(d / "hip_forced_ulp.npz", base * (1 + 1e-6), acts, own=own)
    out = tmp_path / "report.json"
    T.forced_report(argparse.Namespace(seeds="0,1,2", window=7, dir=str(tmp_path), out=str(out)))
    rep = json.load(open(out))
    assert set(rep["pairs"]) == {"engines_final_tree", "device_ulp"}                 # only the pairs whose files exist
    eng = rep["pairs"]["engines_final_tree"]
    assert sorted(eng["seeds"]) == ["0", "1"] and eng["seeds"]["0"]["iterations"] == it
    for nm in T.NAMES:
        w = eng["over_seeds"][nm]["median_over_seeds_per_window"]
        assert len(w) == 2 and all(abs(x - 0.01 / 1.01) < 1e-4 for x in w)           # |a - b| / |b|, b = the second run
        assert rep["pairs"]["device_ulp"]["over_seeds"][nm]["worst_window_of_any_seed"] < 2e-6
    k = "own_samples_of_the_first_run_differing_from_the_history"
    assert rep["pairs"]["device_ulp"]["seeds"]["0"][k].startswith("3 / ")


def test_runs_on_different_histories_are_refused(tmp_path):
    T = _tool()
    d = tmp_path / "seed0"
    d.mkdir()
    steps = 7 * (T.PLAYING // T.N_ENV)
    a = np.zeros((steps, T.N_ENV), dtype=np.int8)
    b = a.copy()
    b[5, 5] = 1
    r = np.ones((7, 2, 5))
    _write(d / "hip_forced.npz", r, a)
    _write(d / "oracle_forced.npz", r, b)
    try:
        T.forced_report(argparse.Namespace(seeds="0", window=7, dir=str(tmp_path), out=str(tmp_path / "r.json")))
    except AssertionError:
        return
    raise AssertionError("two runs with different recorded actions were compared as an identical-history pair")


def test_ensemble_statistic_of_two_run_sets(tmp_path):
    T = _tool()
    rng = np.random.RandomState(1)
    it, ep, K = 14, 2, 4
    steps = it * (T.PLAYING // T.N_ENV)
    for sd in range(K):
        d = tmp_path / ("seed%d" % sd)
        d.mkdir()
        acts = rng.randint(0, T.A, size=(steps, T.N_ENV)).astype(np.int8)
        base = 1.0 + rng.rand(it, ep, 5)
        _write(d / "hip.npz", base, acts)
        _write(d / "hip_ulp.npz", base * 1.02, acts)
    out = tmp_path / "ens.json"
    T.ensemble(argparse.Namespace(seeds="0,1,2,3", window=7, dir=str(tmp_path), out=str(out), first="hip.npz",
                                  second="hip_ulp.npz"))
    rep = json.load(open(out))
    for nm in T.NAMES:
        s = rep["signals"][nm]
        assert abs(s["max_rel_diff_of_ensemble_means"] - 0.02 / 1.02) < 1e-4 and not s["within_1_percent_in_every_window"]
    assert rep["first_vector_step_with_a_different_sampled_action"] == {str(i): None for i in range(K)}
