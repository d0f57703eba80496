"""Loss curves of the C2 configuration (BASELINE.json: Atari-like Clipped PPO, 64 vectorized envs, 84x84x4 uint8
observations, GAE) — the HIP engine against the CPU oracle agent over >= 100 k env-steps, from the same initial weights,
the same synthetic env bytes and the same host RNG streams.  The two sides run SEPARATELY, because one oracle update of
this network is ~2 s of numpy on the build container and ~0.4 s on the GPU box's host, against 0.2 ms on the device:

    python tools/loss_curve_c2.py --side hip    --dir gpurun_out/lc_c2 --iterations 49 --epochs 2     # MI355X, seconds
    python tools/loss_curve_c2.py --side oracle --dir gpurun_out/lc_c2                                # CPU, hours
    python tools/loss_curve_c2.py --compare     --dir gpurun_out/lc_c2 --out profiles/r03_loss_curve_c2.json

    python tools/loss_curve_c2.py --side oracle --perturb-ulp --dir gpurun_out/lc_c2                  # the noise floor:
    python tools/loss_curve_c2.py --compare --first oracle_ulp.npz --dir gpurun_out/lc_c2 --out profiles/...noise_floor.json

Round 5 — what "within 1 %" can mean at the real schedule (10 epochs): the same statistic for the device against ITSELF started one ulp
away, and runs that train on an IDENTICAL action history:

    python tools/loss_curve_c2.py --side hip --hip-seeds 0,..,15 --epochs 10 --no-init [--perturb-ulp]   # hip.npz / hip_ulp.npz
    python tools/loss_curve_c2.py --ensemble --seeds 0,..,15 --first hip.npz --second hip_ulp.npz --out ...   # device vs device + 1 ulp
    python tools/loss_curve_c2.py --side hip    --follow-hip-actions [--perturb-ulp | --perturb-all | --split-cap 16] ...   # followers of
    python tools/loss_curve_c2.py --side oracle --follow-hip-actions [--perturb-ulp] --dir <dir>/seed<s>                     # hip.npz's actions
    python tools/loss_curve_c2.py --forced --seeds 0,..,7 --dir <dir> --out ...        # pair by pair: engines; each engine vs itself

The hip side writes the initial weights, the host RNG state after construction and its results; the oracle side
starts from exactly those.  The synthetic env ignores the actions, so both sides see the same observations for the
whole run even after the first differently sampled action (reported).  Reduced against the preset in ONE respect,
stated in the output: `epochs` optimisation epochs per 2048-step rollout instead of 10 (the oracle's cost)."""
import argparse
import json
import os
import pickle
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

N_ENV, L, A, PLAYING, B, FRAME = 64, 32, 6, 2048, 64, (84, 84)
NAMES = ["surrogate", "entropy", "kl", "policy_total", "value_loss"]


def hip_side(args):
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    dev = torch.device("cuda:0")
    if args.split_cap:
        # another valid fp32 engine: products cut into at most this many K slices (default 64) — the weight gradients of
        # the convolutions are then summed in other groups, every update, everywhere (a CONTINUOUS source of last-bit
        # differences, like the one between two engines; --perturb-* only move the starting point)
        from coach_amd import _rlx
        _rlx.lib().gemm_split_cap(args.split_cap)
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("image", N_ENV, FRAME, A, episode_length=L,
                                                                         seed=1234 + args.seed), dev)
    p = ClippedPPOAgentParameters()
    p.seed = args.seed
    p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAYING)
    p.algorithm.optimization_epochs = args.epochs
    p.network_wrappers["main"].batch_size = B
    if args.clip_eps:
        # not C2: another clipping range of the surrogate.  (Tried with 10, where the clip almost never binds, to see whether a
        # one-ulp perturbation still grows as fast: ten unclipped epochs per rollout collapse the policy to determinism within
        # the first seven iterations — entropy 0.00 in 4 of 5 seeds — so that run says nothing about C2: call 30.)
        p.algorithm.clip_likelihood_ratio_using_epsilon = args.clip_eps
    agent = ClippedPPOAgent(p, env, dev)
    os.makedirs(args.dir, exist_ok=True)
    forced = None
    if args.follow_hip_actions:
        # the device trains on the action history of <dir>/hip.npz (an earlier run of this side) instead of its own
        # samples: the step's action column is overwritten before anything reads it (the synthetic env ignores actions;
        # the recorded probabilities are whole rows).  With --perturb-ulp: the DEVICE's own noise floor under an
        # identical history (hip_forced_ulp.npz against hip_forced.npz), to set beside engine against engine
        # (hip_forced.npz against oracle_forced.npz) and the oracle's (oracle_forced_ulp.npz against oracle_forced.npz).
        forced = torch.from_numpy(np.load(os.path.join(args.dir, "hip.npz"))["actions"].astype(np.int32)).to(dev)
    if args.perturb_all:
        # EVERY weight one ulp up or down (seeded coin per weight): the size of difference two engines have everywhere
        # (another summation order in every product), where --perturb-ulp moves one weight per tower
        w = agent.networks["main"].params.weights
        g = torch.Generator().manual_seed(777 + args.seed)
        up = (torch.rand(w.numel(), generator=g) < 0.5).to(dev)
        inf = torch.full_like(w, float("inf"))
        w.copy_(torch.nextafter(w, torch.where(up, inf, -inf)))
        print("perturbed by one ulp: every weight (%d), random direction" % w.numel(), flush=True)
    elif args.perturb_ulp:
        params = agent.networks["main"].params
        k = sorted(n for n in params.entries if n.endswith("kernel"))[0]       # the oracle side's choice (same names)
        towers = params.entries[k][2]
        for t in range(towers):
            flat = params.w(k, t).view(-1)
            flat[0] = float(np.nextafter(np.float32(flat[0].item()), np.float32(np.inf), dtype=np.float32))
        t = towers
        print("perturbed by one ulp: %s[0] of %d tower(s)" % (k, t), flush=True)
    if not args.no_init and forced is None:           # (13.5 MB per seed; the oracle side can rebuild the same arrays on the CPU: rebuild_init)
        arrays = agent.networks["main"].params.named_arrays()
        np.savez_compressed(os.path.join(args.dir, "init.npz"), **{"%s|%d" % (k, t): a for k, v in arrays.items()
                                                                    for t, a in enumerate(v)})
    if forced is None:
        with open(os.path.join(args.dir, "rng_state.pkl"), "wb") as f:
            pickle.dump({"random": random.getstate(), "numpy": np.random.get_state(), "epochs": args.epochs,
                         "iterations": args.iterations, "seed": args.seed}, f)
    res, acts, own = [], [], []
    t0 = time.perf_counter()
    for it in range(args.iterations):
        while True:
            agent.act()
            if forced is not None:
                own.append(agent.actions.cpu().numpy().astype(np.int8))
                r0 = (agent.memory.steps - 1) * N_ENV
                agent.memory.action[r0:r0 + N_ENV].copy_(forced[len(acts)])
                acts.append(forced[len(acts)].cpu().numpy().astype(np.int8))
            else:
                acts.append(agent.actions.cpu().numpy().astype(np.int8))
            r = agent.train()
            if r is not None:
                break
        res.append(np.array([x.cpu().numpy()[:5] for x in r], dtype=np.float64))
    agent.networks["main"].check_status()
    tag = "_ulpall" if args.perturb_all else ("_ulp" if args.perturb_ulp else "")
    if args.split_cap:
        tag += "_cap%d" % args.split_cap
    if args.clip_eps:
        tag += "_eps%g" % args.clip_eps
    name = ("hip%s.npz" % tag) if forced is None else ("hip_forced%s.npz" % tag)
    np.savez_compressed(os.path.join(args.dir, name), results=np.array(res), actions=np.array(acts),
                        own_actions=np.array(own), seconds=time.perf_counter() - t0)
    print("hip side: %d iterations, %d env-steps, %.1f s" % (args.iterations, args.iterations * PLAYING,
                                                             time.perf_counter() - t0))


def rebuild_init(seed):
    """The initial weights ClippedPPOAgent(seed) builds, rebuilt on the CPU without the GPU library: the agent seeds the
    host generators (agents/agent.py:49-55), ClippedPPONet lays out torso x2 / value head / policy head and initialises
    them from RandomState(seed) — the value head's normalized-columns initialiser from the GLOBAL np.random stream
    (heads/head.py:27-33).  Checked bit for bit against the device side's own init.npz (--check-init)."""
    from coach_amd.nn import graph as G, networks as NW
    random.seed(seed)
    np.random.seed(seed)
    params = G.FlatParams()
    torso, feat = NW.build_torso(params, "main", FRAME + (4,), "tanh", 2, "Medium", "Medium")
    vh = G.Dense(params, "main/v_head/dense", feat, 1, None, 1, init=G.normalized_columns(1.0))
    ph = G.Dense(params, "main/ppo_head/policy_fc", feat, A, None, 1)
    params.finalize("cpu")
    rng = np.random.RandomState(seed)
    for m in (torso, vh, ph):
        m.initialize(rng)
    return params.named_arrays()


def oracle_side(args):
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    with open(os.path.join(args.dir, "rng_state.pkl"), "rb") as f:
        st = pickle.load(f)
    init = os.path.join(args.dir, "init.npz")
    if os.path.exists(init):
        fx = np.load(init)
        arrays = {}
        for k in fx.files:
            name, t = k.rsplit("|", 1)
            arrays.setdefault(name, {})[int(t)] = fx[k]
        arrays = {k: [v[t] for t in sorted(v)] for k, v in arrays.items()}
        if args.check_init:
            mine = rebuild_init(st.get("seed", 0))
            assert sorted(mine) == sorted(arrays)
            for k in arrays:
                for a, b in zip(arrays[k], mine[k]):
                    assert a.dtype == b.dtype and np.array_equal(a, b), k
            print("rebuild_init(%d) == init.npz of the device side, every tensor bit for bit" % st.get("seed", 0))
            return
    else:
        arrays = rebuild_init(st.get("seed", 0))
    if args.perturb_ulp:
        # the noise floor: the SAME oracle from weights that differ in the last bit of one convolution weight per tower
        k = sorted(n for n in arrays if n.endswith("kernel"))[0]
        for a in arrays[k]:
            a.flat[0] = np.nextafter(a.flat[0], np.float32(np.inf), dtype=np.float32)
        print("perturbed by one ulp: %s[0] of every tower" % k, flush=True)
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(0, N_ENV, FRAME[0] * FRAME[1], L, 1234 + st.get("seed", 0)), A, batch_size=B,
                              playing_steps=PLAYING, epochs=st["epochs"])
    o.reset(FRAME)
    random.setstate(st["random"])
    np.random.set_state(st["numpy"])
    res, acts = [], []
    out = os.path.join(args.dir, "oracle_ulp.npz" if args.perturb_ulp else "oracle.npz")
    forced = None
    if args.follow_hip_actions:
        # SAME INPUTS for all 100 k steps: the oracle draws its own uniform numbers (the host stream stays aligned) and
        # samples its own action, but RECORDS the action the device took at that vector step — the synthetic env ignores
        # actions, so the recorded transition is the only place an action enters.  Both engines then train on identical
        # histories to the end, and what is compared is f_device(history) against f_oracle(history), not two samples of
        # a stochastic process.  The oracle's own samples are kept (`own_actions`): where they differ from the device's
        # is where a free run would have forked.
        forced = np.load(os.path.join(args.dir, "hip.npz"))["actions"]
        out = os.path.join(args.dir, "oracle_forced_ulp.npz" if args.perturb_ulp else "oracle_forced.npz")
    own = []
    t0 = time.perf_counter()
    for it in range(st["iterations"]):
        for _ in range(PLAYING // N_ENV):
            a, _ = o.act()
            if forced is not None:
                own.append(np.array(a, dtype=np.int8))
                a = [int(x) for x in forced[len(acts)]]
                for e in range(N_ENV):
                    t = o.transitions[e][-1]
                    o.transitions[e][-1] = (t[0], a[e]) + tuple(t[2:])
            acts.append(np.array(a, dtype=np.int8))
        res.append(np.array(o.train(), dtype=np.float64))
        np.savez_compressed(out, results=np.array(res), actions=np.array(acts), own_actions=np.array(own),
                            seconds=time.perf_counter() - t0)
        print("oracle iteration %d / %d  (%.0f s)  %s" % (it + 1, st["iterations"], time.perf_counter() - t0,
                                                          np.round(res[-1].mean(0), 5)), flush=True)


def compare(args):
    h, o = np.load(os.path.join(args.dir, args.first)), np.load(os.path.join(args.dir, "oracle.npz"))
    with open(os.path.join(args.dir, "rng_state.pkl"), "rb") as f:
        st = pickle.load(f)
    n = min(len(h["results"]), len(o["results"]))
    hip, orc = h["results"][:n].mean(1), o["results"][:n].mean(1)                # per-iteration means over the epochs
    steps = n * (PLAYING // N_ENV)
    ha, oa = h["actions"][:steps], o["actions"][:steps]
    diff = np.nonzero((ha != oa).any(1))[0]
    first = int(diff[0]) if diff.size else None
    W = args.window
    wins = {}
    for j, nm in enumerate(NAMES):
        k = n // W * W
        hw, ow = hip[:k, j].reshape(-1, W).mean(1), orc[:k, j].reshape(-1, W).mean(1)
        rel = np.abs(hw - ow) / np.maximum(np.abs(ow), 1e-12)
        wins[nm] = {"max_window_rel_diff": float(rel.max()), "per_window_rel_diff": [round(float(x), 6) for x in rel],
                    "hip_last_window": float(hw[-1]), "oracle_last_window": float(ow[-1])}
    kk = (first // (PLAYING // N_ENV)) if first is not None else n          # iterations with identical sampled actions
    before = {nm: (float(np.max(np.abs(hip[:kk, j] - orc[:kk, j]) / np.maximum(np.abs(orc[:kk, j]), 1e-12))) if kk else None)
              for j, nm in enumerate(NAMES)}
    out = {"workload": "C2: Clipped PPO, %d vectorized envs, 84x84x4 uint8 observations (synthetic, episodes of %d), %d "
                       "actions, GAE(0.99, 0.95), rollout %d, minibatch %d, fp32 conv torso x2, %d optimisation epochs per "
                       "rollout%s" % (N_ENV, L, A, PLAYING, B, st["epochs"],
                                      "" if st["epochs"] == 10 else " (the C2 configuration has 10)"),
           "iterations": n, "env_steps": n * PLAYING, "updates": n * st["epochs"] * (PLAYING // B),
           "window_iterations": W,
           "first_vector_step_with_a_different_sampled_action": first,
           "first_diverging_action": None if first is None else {
               "vector_step": first, "iteration": first // (PLAYING // N_ENV),
               "envs": np.nonzero(ha[first] != oa[first])[0].tolist(),
               "hip": ha[first][ha[first] != oa[first]].tolist(), "oracle": oa[first][ha[first] != oa[first]].tolist()},
           "identical_sampled_actions": "%d / %d" % (int((ha == oa).sum()), ha.size),
           "max_per_iteration_rel_diff_while_actions_identical": before,
           "compared": "%s vs oracle.npz" % args.first, "signals": wins, "seconds_hip": float(h["seconds"]), "seconds_oracle_cpu": float(o["seconds"]),
           "oracle_host": "build container, numpy"}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "signals"}))
    for nm in NAMES:
        print(nm, wins[nm]["max_window_rel_diff"])


def ensemble(args):
    """K independent experiments (agent seed s: initial weights and host streams; env seed 1234 + s), each run on both
    sides from identical starting points (<dir>/seed<s>/{hip,oracle}.npz).  Per window of `--window` iterations and per
    signal: mean and standard error over the seeds on either side, the relative difference of the two ENSEMBLE MEANS
    and that difference in units of its own standard error (paired over the seeds: the two sides of a seed share
    everything up to the first differently sampled action)."""
    seeds = [int(x) for x in args.seeds.split(",")]
    W = args.window
    per_side = {"hip": [], "oracle": []}
    first_div = {}
    n = None
    for sd in seeds:
        d = os.path.join(args.dir, "seed%d" % sd)
        h, o = np.load(os.path.join(d, args.first)), np.load(os.path.join(d, args.second))
        k = min(len(h["results"]), len(o["results"]))
        n = k if n is None else min(n, k)
        per_side["hip"].append(h["results"].mean(1))
        per_side["oracle"].append(o["results"].mean(1))
        steps = k * (PLAYING // N_ENV)
        diff = np.nonzero((h["actions"][:steps] != o["actions"][:steps]).any(1))[0]
        first_div[sd] = int(diff[0]) if diff.size else None
    k = n // W * W
    hip = np.stack([x[:k] for x in per_side["hip"]])            # [seed, iteration, signal]
    orc = np.stack([x[:k] for x in per_side["oracle"]])
    K = len(seeds)
    out = {"workload": "C2: Clipped PPO, %d vectorized envs, 84x84x4 uint8 observations, rollout %d, minibatch %d; "
                       "%d seeds x %d iterations (%d env-steps each)" % (N_ENV, PLAYING, B, K, k, k * PLAYING),
           "seeds": seeds, "window_iterations": W,
           "compared": "%s ('hip_*' keys below) against %s ('oracle_*' keys below), free-running: each run trains on its "
                       "own sampled actions" % (args.first, args.second),
           "first_vector_step_with_a_different_sampled_action": first_div, "signals": {}}
    for j, nm in enumerate(NAMES):
        hw = hip[:, :, j].reshape(K, -1, W).mean(2)             # [seed, window]
        ow = orc[:, :, j].reshape(K, -1, W).mean(2)
        hm, om = hw.mean(0), ow.mean(0)
        hse, ose = hw.std(0, ddof=1) / np.sqrt(K), ow.std(0, ddof=1) / np.sqrt(K)
        d = hw - ow                                             # paired differences
        dse = d.std(0, ddof=1) / np.sqrt(K)
        rel = np.abs(hm - om) / np.maximum(np.abs(om), 1e-12)
        out["signals"][nm] = {
            "hip_mean": [float(x) for x in hm], "hip_se": [float(x) for x in hse],
            "oracle_mean": [float(x) for x in om], "oracle_se": [float(x) for x in ose],
            "rel_diff_of_ensemble_means": [round(float(x), 6) for x in rel],
            "oracle_relative_se": [round(float(x), 6) for x in ose / np.maximum(np.abs(om), 1e-12)],
            "paired_diff_in_standard_errors": [round(float(x), 3) for x in d.mean(0) / np.maximum(dse, 1e-300)],
            "max_rel_diff_of_ensemble_means": float(rel.max()),
            "within_1_percent_in_every_window": bool((rel <= 0.01).all()),
            "within_two_standard_errors_in_every_window": bool((np.abs(d.mean(0)) <= 2 * np.maximum(dse, 1e-300)).all())}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    for nm in NAMES:
        s_ = out["signals"][nm]
        print("%-14s max rel diff of ensemble means %.4f   (oracle's own relative s.e. up to %.4f)   within 1 %%: %s   "
              "within 2 s.e.: %s" % (nm, s_["max_rel_diff_of_ensemble_means"], max(s_["oracle_relative_se"]),
                                    s_["within_1_percent_in_every_window"], s_["within_two_standard_errors_in_every_window"]))


def forced_report(args):
    """Runs that trained on the SAME action history (the one of <dir>/seed<s>/hip.npz), pair by pair, for each seed in
    --seeds:
        engines      hip_forced.npz (or hip.npz itself)  against  oracle_forced.npz      device engine vs CPU oracle
        device_ulp   hip_forced_ulp.npz                  against  hip_forced.npz         the device against itself, one
                                                                                         initial weight per tower moved by 1 ulp
        device_every_weight_one_ulp  hip_forced_ulpall.npz against hip_forced.npz       ... EVERY initial weight moved by 1 ulp
        device_other_k_split  hip_forced_cap16.npz       against  hip_forced.npz         ... with products cut into <= 16 K slices
                                                                                         instead of <= 64: other summation groups
                                                                                         in every update (a second fp32 engine)
        oracle_ulp   oracle_forced_ulp.npz               against  oracle_forced.npz      the oracle against itself, same move
    Per pair, seed and signal: the relative difference of the per-iteration mean losses and of the window means (the
    statistic of the free-running ensembles); per pair and signal over the seeds: median and maximum per window.  If the
    engines pair is not larger than the two self pairs, the two engines are as close as either is to itself."""
    seeds = [int(x) for x in args.seeds.split(",")]
    W = args.window
    pairs = [("engines_final_tree", "hip_forced.npz", "oracle_forced.npz"),
             ("device_ulp", "hip_forced_ulp.npz", "hip_forced.npz"),
             ("device_every_weight_one_ulp", "hip_forced_ulpall.npz", "hip_forced.npz"),
             ("device_other_k_split", "hip_forced_cap16.npz", "hip_forced.npz"),
             ("oracle_ulp", "oracle_forced_ulp.npz", "oracle_forced.npz")]
    out = {"workload": "C2: Clipped PPO, %d vectorized envs, 84x84x4 uint8 observations, rollout %d, minibatch %d; every run "
                       "of a seed records the action history of that seed's hip.npz (identical histories)"
                       % (N_ENV, PLAYING, B), "window_iterations": W, "pairs": {}}
    for pname, fa, fb in pairs:
        per_seed, wins = {}, {nm: [] for nm in NAMES}
        for sd in seeds:
            d = os.path.join(args.dir, "seed%d" % sd)
            if not (os.path.exists(os.path.join(d, fa)) and os.path.exists(os.path.join(d, fb))):
                continue
            x, y = np.load(os.path.join(d, fa)), np.load(os.path.join(d, fb))
            n = min(len(x["results"]), len(y["results"]))
            if n == 0:
                continue
            steps = n * (PLAYING // N_ENV)
            assert np.array_equal(x["actions"][:steps], y["actions"][:steps]), (pname, sd)
            xr, yr = x["results"][:n], y["results"][:n]                # [iteration, epoch, signal]
            rec = {"iterations": n, "env_steps": n * PLAYING, "updates": n * xr.shape[1] * (PLAYING // B), "signals": {}}
            for side, f in (("first", x), ("second", y)):
                if "own_actions" in f.files and len(f["own_actions"]):
                    own = f["own_actions"][:steps]
                    rec["own_samples_of_the_%s_run_differing_from_the_history" % side] = \
                        "%d / %d" % (int((own != x["actions"][:len(own)]).sum()), own.size)
            k = n // W * W
            for j, nm in enumerate(NAMES):
                xi, yi = xr[:, :, j].mean(1), yr[:, :, j].mean(1)
                rel_i = np.abs(xi - yi) / np.maximum(np.abs(yi), 1e-12)
                sig = {"rel_diff_per_iteration": [float("%.3g" % v) for v in rel_i],
                       "max_rel_diff_per_iteration": float(rel_i.max())}
                if k:
                    xw, yw = xi[:k].reshape(-1, W).mean(1), yi[:k].reshape(-1, W).mean(1)
                    rel_w = np.abs(xw - yw) / np.maximum(np.abs(yw), 1e-12)
                    sig["rel_diff_per_window"] = [float("%.3g" % v) for v in rel_w]
                    wins[nm].append(rel_w)
                rec["signals"][nm] = sig
            per_seed[str(sd)] = rec
        if not per_seed:
            continue
        summ = {}
        for nm in NAMES:
            if wins[nm]:
                m = min(len(w) for w in wins[nm])
                a_ = np.stack([w[:m] for w in wins[nm]])
                summ[nm] = {"seeds": len(wins[nm]), "median_over_seeds_per_window": [float("%.3g" % v) for v in np.median(a_, 0)],
                            "max_over_seeds_per_window": [float("%.3g" % v) for v in a_.max(0)],
                            "worst_window_of_any_seed": float(a_.max())}
        out["pairs"][pname] = {"first": fa, "second": fb, "over_seeds": summ, "seeds": per_seed}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    for pname, p_ in out["pairs"].items():
        print("%s  (%s vs %s), %d seed(s), iterations %s" % (pname, p_["first"], p_["second"], len(p_["seeds"]),
                                                             sorted(set(r["iterations"] for r in p_["seeds"].values()))))
        for nm in NAMES:
            if nm in p_["over_seeds"]:
                o_ = p_["over_seeds"][nm]
                print("   %-13s median per window %s   max %s" % (nm, o_["median_over_seeds_per_window"], o_["max_over_seeds_per_window"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0, help="agent seed (weights, host streams); the env seed is 1234 + seed")
    ap.add_argument("--ensemble", action="store_true", help="compare <dir>/seed<s>/ for s in --seeds")
    ap.add_argument("--seeds", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--side", choices=["hip", "oracle"])
    ap.add_argument("--compare", action="store_true")
    ap.add_argument("--dir", default="gpurun_out/lc_c2")
    ap.add_argument("--iterations", type=int, default=49)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--window", type=int, default=7)
    ap.add_argument("--out", default="profiles/r03_loss_curve_c2.json")
    ap.add_argument("--hip-seeds", default="", help="hip side: run these seeds in turn into <dir>/seed<s>/")
    ap.add_argument("--no-init", action="store_true", help="hip side: do not write init.npz (the oracle side rebuilds it)")
    ap.add_argument("--check-init", action="store_true",
                    help="oracle side: only check that rebuild_init equals the device side's init.npz")
    ap.add_argument("--perturb-ulp", action="store_true",
                    help="oracle side: start from weights one ulp away in one element per tower (-> oracle_ulp.npz)")
    ap.add_argument("--clip-eps", type=float, default=0.0,
                    help="hip side: clip_likelihood_ratio_using_epsilon of the run (-> ..._eps<X>.npz); 0 = the preset's 0.2")
    ap.add_argument("--split-cap", type=int, default=0,
                    help="hip side: rlx_gemm_split_cap(N) for the whole run (-> hip[_forced]_cap<N>.npz): other summation groups")
    ap.add_argument("--perturb-all", action="store_true",
                    help="hip side: start from weights in which EVERY element is one ulp up or down (-> hip[_forced]_ulpall.npz)")
    ap.add_argument("--follow-hip-actions", action="store_true",
                    help="oracle side: record the device's actions (hip.npz) instead of the oracle's own samples "
                         "(-> oracle_forced.npz): identical histories on both sides for the whole run")
    ap.add_argument("--forced", action="store_true", help="compare / ensemble: hip.npz against oracle_forced.npz")
    ap.add_argument("--second", default="oracle.npz", help="ensemble: the second run set (default: the oracle's)")
    ap.add_argument("--first", default="hip.npz",
                    help="compare: the run set against oracle.npz (oracle_ulp.npz = the oracle against itself)")
    args = ap.parse_args()
    if args.forced:
        forced_report(args)
    elif args.ensemble:
        ensemble(args)
    elif args.compare:
        compare(args)
    elif args.side == "hip" and args.hip_seeds:
        base = args.dir
        for sd in (int(x) for x in args.hip_seeds.split(",")):       # several experiments in one process (one import)
            args.seed, args.dir = sd, os.path.join(base, "seed%d" % sd)
            hip_side(args)
    elif args.side == "hip":
        hip_side(args)
    else:
        oracle_side(args)


if __name__ == "__main__":
    main()
