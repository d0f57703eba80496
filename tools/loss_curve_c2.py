"""Loss curves of the C2 configuration (BASELINE.json: Atari-like Clipped PPO, 64 vectorized envs, 84x84x4 uint8
observations, GAE) — the HIP engine against the CPU oracle agent over >= 100 k env-steps, from the same initial weights,
the same synthetic env bytes and the same host RNG streams.  The two sides run SEPARATELY, because one oracle update of
this network is ~2 s of numpy on the build container and ~0.4 s on the GPU box's host, against 0.2 ms on the device:

    python tools/loss_curve_c2.py --side hip    --dir gpurun_out/lc_c2 --iterations 49 --epochs 2     # MI355X, seconds
    python tools/loss_curve_c2.py --side oracle --dir gpurun_out/lc_c2                                # CPU, hours
    python tools/loss_curve_c2.py --compare     --dir gpurun_out/lc_c2 --out profiles/r03_loss_curve_c2.json

    python tools/loss_curve_c2.py --side oracle --perturb-ulp --dir gpurun_out/lc_c2                  # the noise floor:
    python tools/loss_curve_c2.py --compare --first oracle_ulp.npz --dir gpurun_out/lc_c2 --out profiles/...noise_floor.json

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
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("image", N_ENV, FRAME, A, episode_length=L,
                                                                         seed=1234), dev)
    p = ClippedPPOAgentParameters()
    p.seed = 0
    p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAYING)
    p.algorithm.optimization_epochs = args.epochs
    p.network_wrappers["main"].batch_size = B
    agent = ClippedPPOAgent(p, env, dev)
    os.makedirs(args.dir, exist_ok=True)
    arrays = agent.networks["main"].params.named_arrays()
    np.savez_compressed(os.path.join(args.dir, "init.npz"), **{"%s|%d" % (k, t): a for k, v in arrays.items()
                                                                for t, a in enumerate(v)})
    with open(os.path.join(args.dir, "rng_state.pkl"), "wb") as f:
        pickle.dump({"random": random.getstate(), "numpy": np.random.get_state(), "epochs": args.epochs,
                     "iterations": args.iterations}, f)
    res, acts = [], []
    t0 = time.perf_counter()
    for it in range(args.iterations):
        while True:
            agent.act()
            acts.append(agent.actions.cpu().numpy().astype(np.int8))
            r = agent.train()
            if r is not None:
                break
        res.append(np.array([x.cpu().numpy()[:5] for x in r], dtype=np.float64))
    agent.networks["main"].check_status()
    np.savez_compressed(os.path.join(args.dir, "hip.npz"), results=np.array(res), actions=np.array(acts),
                        seconds=time.perf_counter() - t0)
    print("hip side: %d iterations, %d env-steps, %.1f s" % (args.iterations, args.iterations * PLAYING,
                                                             time.perf_counter() - t0))


def oracle_side(args):
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    with open(os.path.join(args.dir, "rng_state.pkl"), "rb") as f:
        st = pickle.load(f)
    fx = np.load(os.path.join(args.dir, "init.npz"))
    arrays = {}
    for k in fx.files:
        name, t = k.rsplit("|", 1)
        arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {k: [v[t] for t in sorted(v)] for k, v in arrays.items()}
    if args.perturb_ulp:
        # the noise floor: the SAME oracle from weights that differ in the last bit of one convolution weight per tower
        k = sorted(n for n in arrays if n.endswith("kernel"))[0]
        for a in arrays[k]:
            a.flat[0] = np.nextafter(a.flat[0], np.float32(np.inf), dtype=np.float32)
        print("perturbed by one ulp: %s[0] of every tower" % k, flush=True)
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(0, N_ENV, FRAME[0] * FRAME[1], L, 1234), A, batch_size=B,
                              playing_steps=PLAYING, epochs=st["epochs"])
    o.reset(FRAME)
    random.setstate(st["random"])
    np.random.set_state(st["numpy"])
    res, acts = [], []
    out = os.path.join(args.dir, "oracle_ulp.npz" if args.perturb_ulp else "oracle.npz")
    t0 = time.perf_counter()
    for it in range(st["iterations"]):
        for _ in range(PLAYING // N_ENV):
            a, _ = o.act()
            acts.append(np.array(a, dtype=np.int8))
        res.append(np.array(o.train(), dtype=np.float64))
        np.savez_compressed(out, results=np.array(res), actions=np.array(acts), seconds=time.perf_counter() - t0)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["hip", "oracle"])
    ap.add_argument("--compare", action="store_true")
    ap.add_argument("--dir", default="gpurun_out/lc_c2")
    ap.add_argument("--iterations", type=int, default=49)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--window", type=int, default=7)
    ap.add_argument("--out", default="profiles/r03_loss_curve_c2.json")
    ap.add_argument("--perturb-ulp", action="store_true",
                    help="oracle side: start from weights one ulp away in one element per tower (-> oracle_ulp.npz)")
    ap.add_argument("--first", default="hip.npz",
                    help="compare: the run set against oracle.npz (oracle_ulp.npz = the oracle against itself)")
    args = ap.parse_args()
    if args.compare:
        compare(args)
    elif args.side == "hip":
        hip_side(args)
    else:
        oracle_side(args)


if __name__ == "__main__":
    main()
