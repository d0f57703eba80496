"""How far do two CORRECT fp32 implementations of the same DQN loop drift apart?  (CPU only.)

The north-star asks for loss curves within 1 % of the reference over 100k updates.  A DQN loop is a chaotic map of
its rounding errors (Adam divides by sqrt(v) + 1e-4: gradients near zero flip the step's sign), so that bound can
only be judged against the drift between two runs of the ORACLE ITSELF that differ in nothing but the order in
which a matrix product is summed — here: numpy's fp32 BLAS product versus the same product accumulated in fp64 and
rounded once.  Same weights, same host RNG streams, identical replay indices by construction.

    python tools/loss_curve_noise_floor.py --steps 100000 --out profiles/r02_loss_curve_noise_floor_c1.json
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(steps, heatup, exact_products):
    from coach_amd.schedules import LinearSchedule
    from oracle import nn as N
    from oracle.agents import DQNAgentOracle
    from oracle.synth_env import SynthVecEnv
    L, A, D = 200, 2, 4
    fwd, bwd = N.Dense.forward, N.Dense.backward
    if exact_products:
        F32, F64 = np.float32, np.float64

        def forward(self, x):
            self.x = x
            self.y = N.act((x.astype(F64) @ self.W.astype(F64)).astype(F32) + self.b, self.act)
            return self.y

        def backward(self, dy):
            dz = dy * N.act_grad(self.y, self.act)
            self.dW = (self.x.T.astype(F64) @ dz.astype(F64)).astype(F32)
            self.db = dz.sum(0)
            return (dz.astype(F64) @ self.W.T.astype(F64)).astype(F32)
        N.Dense.forward, N.Dense.backward = forward, backward
    try:
        random.seed(0); np.random.seed(0)
        rng, arrays, feat = np.random.RandomState(0), {}, D
        for name, units in (("main/embedder/dense0", 256), ("main/middleware/dense0", 512), ("main/q_head/dense", A)):
            lim = np.sqrt(6.0 / (feat + units))                     # xavier-uniform, as the device networks
            arrays[name + "/kernel"] = [rng.uniform(-lim, lim, (feat, units)).astype(np.float32)]
            arrays[name + "/bias"] = [np.zeros(units, dtype=np.float32)]
            feat = units
        random.seed(0); np.random.seed(0)
        o = DQNAgentOracle(arrays, SynthVecEnv(1, 1, D, L, 1234), A, (D,), capacity=40000, batch_size=32,
                           playing_steps=1, target_every=100, huber=False,
                           epsilon_schedule=LinearSchedule(1.0, 0.01, 10000))
        o.reference_order = True
        o.reset()
        for _ in range(heatup):
            o.heatup_step()
        for _ in range(steps):
            o.act(); o.train()
        return np.array(o.losses, dtype=np.float64), [np.asarray(s).copy() for s in o.sampled]
    finally:
        N.Dense.forward, N.Dense.backward = fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--heatup", type=int, default=1000)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--out", default="profiles/r02_loss_curve_noise_floor_c1.json")
    args = ap.parse_args()
    t0 = time.perf_counter()
    la, sa = run(args.steps, args.heatup, False)
    lb, sb = run(args.steps, args.heatup, True)
    n, W = min(len(la), len(lb)), args.window
    wins = []
    for i in range(0, n - W + 1, W):
        a, b = la[i:i + W].mean(), lb[i:i + W].mean()
        wins.append(float(abs(a - b) / abs(b)))
    same = sum(int(np.array_equal(x, y)) for x, y in zip(sa, sb))
    out = {"what": "numpy oracle (fp32 BLAS products) vs the same oracle with every dense product accumulated in fp64 "
                   "and rounded once; C1 loop of tools/loss_curve.py",
           "updates": int(n), "window": W, "max_window_rel_diff": max(wins), "mean_window_rel_diff": float(np.mean(wins)),
           "windows_above_1_percent": int(sum(w > 0.01 for w in wins)),
           "per_update_max_rel_diff_first_1000": float(np.max(np.abs(la[:1000] - lb[:1000]) / np.abs(lb[:1000]))),
           "batches_with_identical_replay_indices": int(same), "batches": int(len(sa)),
           "seconds": round(time.perf_counter() - t0, 1), "window_rel_diff": wins}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "window_rel_diff"}))


if __name__ == "__main__":
    main()
