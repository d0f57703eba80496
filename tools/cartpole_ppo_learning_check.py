"""CPU pre-check of the CartPole_ClippedPPO golden threshold (build container, no GPU): the oracle's Clipped-PPO agent
(oracle/agents.py ClippedPPOAgentOracle in ragged mode, whose loop is pinned to the real reference agent's) on the oracle
CartPole (oracle/cartpole.py) with the preset's hyper-parameters and schedule — evaluation of 5 greedy episodes every
2048 env-steps, every period starting from a forced reset — and the pass rule of rl_coach/tests/test_golden.py:103-170
(150 within 400 episodes).  A tool, not a test: it shows what the DEVICE run of tests/test_cartpole.py should look like."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.agents import ClippedPPOAgentOracle            # noqa: E402
from oracle.cartpole import CartPoleVecEnv                 # noqa: E402
from oracle.explore import categorical_choice              # noqa: E402

F32 = np.float32


class Env(CartPoleVecEnv):
    kind = 1

    def __init__(self, n, seed):
        super().__init__(n, seed)
        self.n_env = n


class Agent(ClippedPPOAgentOracle):
    def act(self):
        states = np.stack(self.cur)
        states = self.stats.normalize(states).astype(F32)
        probs = self.net.policy_probs(states)
        actions = [categorical_choice(probs[e], np.random.random_sample()) for e in range(self.n_env)]
        nxt, rst, rew, done = self.env.step(actions)
        for e in range(self.n_env):
            self.transitions[e].append((self.cur[e], actions[e], float(rew[e]), bool(done[e]), nxt[e].astype(F32)))
            if done[e]:
                self.episodes.append((e, self.ep_start[e], len(self.transitions[e])))
                self.ep_start[e] = len(self.transitions[e])
            self.cur[e] = (np.array(rst[e]) if done[e] else nxt[e]).astype(F32)
        return done

    def forced_reset(self):
        for e in range(self.n_env):                     # the open episode never reaches the memory
            del self.transitions[e][self.ep_start[e]:]
        first = self.env.reset()
        self.cur = [f.astype(F32) for f in first]


def xavier(rng, fi, fo):
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=(fi, fo)).astype(F32)


def main(seed=0):
    random.seed(seed)
    np.random.seed(seed)
    rng = np.random.RandomState(seed + 100)
    arrays = {}
    for name, (fi, fo) in (("main/embedder/dense0", (4, 64)), ("main/middleware/dense0", (64, 64))):
        arrays[name + "/kernel"] = [xavier(rng, fi, fo), xavier(rng, fi, fo)]
        arrays[name + "/bias"] = [np.zeros(fo, F32), np.zeros(fo, F32)]
    w = rng.randn(64, 1).astype(F32)
    arrays["main/v_head/dense/kernel"] = [(w / np.sqrt((w ** 2).sum(0, keepdims=True))).astype(F32)]
    arrays["main/v_head/dense/bias"] = [np.zeros(1, F32)]
    arrays["main/ppo_head/policy_fc/kernel"] = [xavier(rng, 64, 2)]
    arrays["main/ppo_head/policy_fc/bias"] = [np.zeros(2, F32)]
    env = Env(1, 1234)
    a = Agent(arrays, env, 2, batch_size=64, playing_steps=2048, epochs=10, clip_eps=0.2, beta_entropy=0.0, lr=3e-4,
              reward_clip=None, adam=(0.9, 0.999, 1e-5), ragged=True, normalize=True)
    a.reset()
    ev = Env(1, 777)
    episodes, evals, it = 0, [], 0
    while episodes < 400:
        a.forced_reset()
        steps = 0
        while steps < 2048:
            done = a.act()
            steps += 1
            episodes += int(done[0])
            if a.should_train():
                a.train()
                it += 1
        tot = 0.0
        for _ in range(5):
            s = ev.reset()[0].astype(F32)
            d = False
            while not d:
                p = a.net.policy_probs(a.stats.normalize(s[None]).astype(F32))
                nxt, rst, rew, dn = ev.step([int(np.argmax(p[0]))])
                tot += rew[0]
                d = dn[0]
                s = nxt[0].astype(F32)
        evals.append(tot / 5)
        avg = np.convolve(evals, np.ones(min(len(evals), 10)) / 10, mode='valid')
        print("episodes %d iterations %d eval %.1f averaged max %.1f" % (episodes, it, evals[-1], avg.max()), flush=True)
        if avg.max() >= 150:
            print("PASSED at episode", episodes)
            return True
    print("FAILED")
    return False


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
