"""Loss-curve match of the HIP engine against the CPU oracle (north_star: "loss curves within 1 % of
reference") on the C1 plumbing workload: CartPole-like DQN, 1 env, uniform replay, B = 32, MSE,
1 update per env-step, target copy every 100 steps (presets/CartPole_DQN.py).  Same initial weights,
same host RNG streams; per-update losses are compared in windows.

    python tools/loss_curve.py --steps 20000 --out gpurun_out/loss_curve_c1.json
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
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--heatup", type=int, default=1000)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--out", default="gpurun_out/loss_curve_c1.json")
    args = ap.parse_args()
    from coach_amd.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from coach_amd.core_types import EnvironmentSteps, RunPhase
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    from coach_amd.schedules import LinearSchedule
    from oracle.agents import DQNAgentOracle
    from oracle.synth_env import SynthVecEnv
    dev = torch.device("cuda:0")
    L, A, D = 200, 2, 4
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("vector", 1, (D,), A, episode_length=L,
                                                                         seed=1234), dev)
    p = DQNAgentParameters()
    p.seed = 0
    p.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
    p.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(100)
    p.network_wrappers["main"].replace_mse_with_huber_loss = False
    p.memory.max_size = (MemoryGranularity.Transitions, 40000)
    p.exploration.epsilon_schedule = LinearSchedule(1.0, 0.01, 10000)
    agent = DQNAgent(p, env, dev)
    agent.debug_draws, agent.debug_losses = [], []
    arrays = agent.networks["main"].params.named_arrays()
    random.seed(0); np.random.seed(0)
    o = DQNAgentOracle(arrays, SynthVecEnv(1, 1, D, L, 1234), A, (D,), capacity=40000, batch_size=32,
                       playing_steps=1, target_every=100, huber=False,
                       epsilon_schedule=LinearSchedule(1.0, 0.01, 10000))
    o.reference_order = True          # the store order pinned to the real reference agent (loop.npz)
    o.reset()
    state = (random.getstate(), np.random.get_state())

    t0 = time.perf_counter()
    for _ in range(args.heatup):
        o.heatup_step()
    for _ in range(args.steps):
        o.act(); o.train()
    t_cpu = time.perf_counter() - t0
    random.setstate(state[0]); np.random.set_state(state[1])
    t0 = time.perf_counter()
    agent.phase = RunPhase.HEATUP
    for _ in range(args.heatup):
        agent.act()
    agent.phase = RunPhase.TRAIN
    for _ in range(args.steps):
        agent.act(); agent.train()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    agent.check_status()
    lh, lo = np.array(agent.debug_losses), np.array(o.losses, dtype=np.float64)
    n = min(len(lh), len(lo))
    same_idx = sum(int(np.array_equal(a, b)) for a, b in zip(agent.debug_draws, o.sampled))
    W = args.window
    wins = []
    for i in range(0, n - W + 1, W):
        a, b = lh[i:i + W].mean(), lo[i:i + W].mean()
        wins.append({"updates": [i, i + W], "hip": float(a), "oracle": float(b), "rel_diff": float(abs(a - b) / abs(b))})
    out = {"workload": "C1 CartPole-like DQN (1 env, uniform replay 40k, B=32, MSE, target copy / 100 steps)",
           "updates": int(n), "window": W, "max_window_rel_diff": max(w["rel_diff"] for w in wins),
           "per_update_max_rel_diff_first_1000": float(np.max(np.abs(lh[:1000] - lo[:1000]) / np.abs(lo[:1000]))),
           "batches_with_identical_replay_indices": int(same_idx), "batches": int(len(o.sampled)),
           "seconds_hip_incl_per_update_sync": round(t_gpu, 2), "seconds_oracle_cpu": round(t_cpu, 2),
           "windows": wins}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "windows"}))


if __name__ == "__main__":
    main()
