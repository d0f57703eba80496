"""The two CartPole golden tests (rl_coach/tests/test_golden.py:103-170 on presets/CartPole_DQN.py:47-51 and
CartPole_ClippedPPO.py:66-70) on the MI355X for a range of agent seeds: one line per (preset, seed) with the outcome, the
episode it was decided at and the best averaged evaluation reward.  The reference runs each golden test once, with
`--seed 0`; how often a seed passes says how much of a pass is the seed (oracle on the CPU, tools/cartpole_*_check.py:
DQN 5 of 5 seeds, Clipped PPO 3 of 4).  python tools/cartpole_golden_sweep.py [n_seeds] > profiles/..."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(n_seeds=8):
    dev = torch.device("cuda:0")
    for name in ("CartPole_DQN", "CartPole_ClippedPPO"):
        passed = 0
        for seed in range(n_seeds):
            gm = importlib.import_module("coach_amd.presets." + name).make(agent_seed=seed)
            gm.device = dev
            st = gm.run_preset_validation(time_limit=600)
            passed += int(st["passed"])
            print("%-20s agent_seed %d: %-19s episode %3d of %d, best averaged evaluation reward %6.1f, %3d training "
                  "iterations, %.1f s" % (name, seed, st["reason"], st["episode"], st["max_episodes_to_achieve_reward"],
                                          st["averaged_rewards"].max(), gm.agent.training_iteration, st["wall_s"]), flush=True)
        print("%-20s passed %d of %d seeds" % (name, passed, n_seeds), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
