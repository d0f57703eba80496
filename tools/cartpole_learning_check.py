"""CPU pre-check of the golden thresholds (build container, no GPU): the oracle's DQN arithmetic (oracle/agents.py
DQNOracle = the TF graph restated in numpy) on the oracle CartPole (oracle/cartpole.py) with the hyper-parameters and
the schedule of rl_coach/presets/CartPole_DQN.py — does the averaged evaluation reward reach 150 within 250 episodes
(the pass rule of rl_coach/tests/test_golden.py:103-170)?  Prints one line per evaluation.  Not a test: a tool used to
choose the seed-independent expectations of tests/test_cartpole.py before spending GPU minutes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.agents import DQNOracle                      # noqa: E402
from oracle.cartpole import CartPole                     # noqa: E402


def xavier(rng, fi, fo):
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32)


def main(seed=0, max_episodes=250):
    rng = np.random.RandomState(seed)
    arrays = {}
    for name, (fi, fo) in (("main/embedder/dense0", (4, 256)), ("main/middleware/dense0", (256, 512)),
                           ("main/q_head/dense", (512, 2))):
        arrays[name + "/kernel"] = [xavier(rng, fi, fo)]
        arrays[name + "/bias"] = [np.zeros(fo, np.float32)]
    net = DQNOracle(arrays, (4,), 2, lr=2.5e-4, huber=False)
    env, ev = CartPole(1234, 0), CartPole(99, 1)
    cap = 40000
    S, S2, A, R, D = np.zeros((cap, 4), np.float32), np.zeros((cap, 4), np.float32), np.zeros(cap, np.int64), \
        np.zeros(cap, np.float32), np.zeros(cap, bool)
    n = 0
    total, last_copy, episodes, evals = 0, 0, 0, []
    s = np.array(env.reset(), np.float32)
    t0 = time.time()
    while episodes < max_episodes:
        heatup = total < 1000
        eps = 1.0 if heatup else max(0.01, 1.0 - (total - 1000) * (0.99 / 10000))
        if heatup or rng.rand() < eps:
            a = int(rng.randint(2))
        else:
            a = int(np.argmax(net.q(s[None])[0]))
        s2, r, d = env.step(a)
        i = n % cap
        S[i], S2[i], A[i], R[i], D[i] = s, s2, a, r, d
        n += 1
        total += 1
        s = np.array(s2, np.float32)
        if d:
            episodes += 1
            s = np.array(env.reset(), np.float32)
            if not heatup and episodes % 10 == 0:
                e = np.array(ev.reset(), np.float32)
                ret, dd = 0.0, False
                while not dd:
                    ea = int(rng.randint(2)) if rng.rand() < 0.05 else int(np.argmax(net.q(e[None])[0]))
                    e2, rr, dd = ev.step(ea)
                    ret += rr
                    e = np.array(e2, np.float32)
                evals.append(ret)
                avg = np.convolve(evals, np.ones(min(len(evals), 10)) / 10, mode='valid')
                print("episode %d steps %d eval %.0f averaged max %.1f (%.0f s)" % (episodes, total, ret, avg.max(), time.time() - t0), flush=True)
                if avg.max() >= 150:
                    print("PASSED at episode", episodes)
                    return True
        if not heatup:
            idx = rng.randint(min(n, cap), size=32)
            net.learn_from_batch(S[idx], S2[idx], A[idx], R[idx], D[idx], 0.99)
            if total - last_copy >= 100:
                net.update_target(1.0)
                last_copy = total
    print("FAILED")
    return False


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
