"""Loss-curve match of the HIP Clipped-PPO engine against the CPU oracle agent over many iterations
(vector observations so that the numpy oracle is fast): same initial weights, same synthetic env bytes,
same host RNG streams.  Per-iteration means of [surrogate, entropy, KL, value loss] are compared.

    python tools/loss_curve_ppo.py --iterations 150 --out gpurun_out/loss_curve_ppo.json
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=150)
    ap.add_argument("--out", default="gpurun_out/loss_curve_ppo.json")
    args = ap.parse_args()
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    dev = torch.device("cuda:0")
    n_env, L, D, A, playing, B, epochs = 16, 16, 17, 6, 512, 64, 4
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("vector", n_env, (D,), A, episode_length=L,
                                                                         seed=77), dev)
    p = ClippedPPOAgentParameters()
    p.seed = 0
    p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(playing)
    p.algorithm.optimization_epochs = epochs
    p.algorithm.reward_clipping = None
    p.network_wrappers["main"].batch_size = B
    agent = ClippedPPOAgent(p, env, dev)
    arrays = agent.networks["main"].params.named_arrays()
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(1, n_env, D, L, 77), A, batch_size=B, playing_steps=playing,
                              epochs=epochs, reward_clip=None)
    o.reset()
    state = (random.getstate(), np.random.get_state())
    hip, orc, same_actions, total_actions, first_mismatch = [], [], 0, 0, None
    t_hip = t_cpu = 0.0
    for it in range(args.iterations):
        random.setstate(state[0]); np.random.set_state(state[1])
        t0 = time.perf_counter()
        acts = []
        while True:
            agent.act()
            acts.append(agent.actions.cpu().numpy().copy())
            res = agent.train()
            if res is not None:
                break
        hres = np.array([r.cpu().numpy()[:5] for r in res], dtype=np.float64)
        t_hip += time.perf_counter() - t0
        hip_state = (random.getstate(), np.random.get_state())
        random.setstate(state[0]); np.random.set_state(state[1])
        t0 = time.perf_counter()
        for s in range(len(acts)):
            oa, _ = o.act()
            eq = int(np.sum(np.array(oa) == acts[s]))
            same_actions += eq
            total_actions += n_env
            if eq != n_env and first_mismatch is None:
                first_mismatch = it
        ores = np.array(o.train())
        t_cpu += time.perf_counter() - t0
        assert random.getstate() == hip_state[0]
        state = hip_state
        hip.append(hres.mean(0)); orc.append(ores.mean(0))
    hip, orc = np.array(hip), np.array(orc)
    names = ["surrogate", "entropy", "kl", "policy_total", "value_loss"]
    W = 10
    wins = {}
    for j, nm in enumerate(names):
        hw = hip[:, j].reshape(-1, W).mean(1) if len(hip) % W == 0 else hip[:len(hip) // W * W, j].reshape(-1, W).mean(1)
        ow = orc[:len(hw) * W, j].reshape(-1, W).mean(1)
        wins[nm] = {"max_window_rel_diff": float(np.max(np.abs(hw - ow) / np.maximum(np.abs(ow), 1e-12))),
                    "hip_last": float(hw[-1]), "oracle_last": float(ow[-1])}
    k = first_mismatch if first_mismatch is not None else len(hip)
    before = {nm: (float(np.max(np.abs(hip[:k, j] - orc[:k, j]) / np.maximum(np.abs(orc[:k, j]), 1e-12))) if k else None)
              for j, nm in enumerate(names)}
    out = {"first_iteration_with_a_different_sampled_action": first_mismatch,
           "max_per_iteration_rel_diff_before_it": before,
           "workload": "Clipped PPO, %d envs, obs %d, %d actions, rollout %d, B=%d, %d epochs (tanh 256-512 towers x2)"
                       % (n_env, D, A, playing, B, epochs),
           "iterations": args.iterations, "updates": args.iterations * epochs * (playing // B), "window_iterations": W,
           "identical_sampled_actions": "%d / %d" % (same_actions, total_actions), "signals": wins,
           "seconds_hip": round(t_hip, 2), "seconds_oracle_cpu": round(t_cpu, 2)}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
