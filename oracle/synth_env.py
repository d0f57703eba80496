"""Synthetic vector environment — numpy twin of coach_amd/csrc/synth_env.hip (same Philox4x32-10
counters, so CPU and GPU produce identical bytes).  Not a restatement of reference
code (the reference has no synthetic env): this is the shared workload generator; Philox itself is
pinned against the known-answer vectors of Salmon et al.'s Random123 (tests/test_synth_env.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    k0 = np.uint64(k0) & MASK
    k1 = np.asarray(k1, dtype=np.uint64) & MASK
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c0, c1, c2, c3


def _u01(x):
    return (x >> np.uint64(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)


def _irwin_hall(r):
    s = ((_u01(r[0]) + _u01(r[1])) + _u01(r[2])) + _u01(r[3])
    return (s - np.float32(2.0)) * np.float32(1.7320508075688772)


def observation(kind, seed, env, ep, t, obs_elems):
    if kind == 0:
        j = np.arange(obs_elems // 16, dtype=np.uint64)
        r = philox4x32_10(np.full_like(j, ep), np.full_like(j, t), j, np.zeros_like(j), seed,
                          np.full_like(j, env))
        words = np.stack(r, axis=1).astype(np.uint32)          # x,y,z,w little-endian dwords
        return words.reshape(-1).view(np.uint8).copy()
    j = np.arange(obs_elems, dtype=np.uint64)
    r = philox4x32_10(np.full_like(j, ep), np.full_like(j, t), j, np.zeros_like(j), seed,
                      np.full_like(j, env))
    return _irwin_hall(r)


def reward(kind, seed, env, ep, t):
    one = np.ones(1, dtype=np.uint64)
    r = philox4x32_10(one * ep, one * t, one * 0, one * 1, seed, one * env)
    if kind == 0:
        u = _u01(r[0])[0]
        return np.float32(-1.0 if u < np.float32(0.05) else (0.0 if u < np.float32(0.95) else 1.0))
    return _irwin_hall(r)[0]


class SynthVecEnv:
    def __init__(self, kind, n_env, obs_elems, episode_len, seed, env_id0=0, episode_lengths=None):
        self.kind, self.n_env, self.obs_elems, self.L = kind, n_env, obs_elems, episode_len
        self.lengths = [episode_len] * n_env if episode_lengths is None else list(episode_lengths)
        self.seed, self.env_id0 = seed, env_id0
        self.ep = np.zeros(n_env, dtype=np.int64)
        self.t = np.zeros(n_env, dtype=np.int64)

    def reset(self):
        self.ep[:] = 0
        self.t[:] = 0
        return np.stack([observation(self.kind, self.seed, self.env_id0 + e, 0, 0, self.obs_elems)
                         for e in range(self.n_env)])

    def step(self):
        nxt, rst, rew, done = [], [], [], []
        for e in range(self.n_env):
            ep, t = int(self.ep[e]), int(self.t[e])
            d = t + 1 >= self.lengths[e]
            nxt.append(observation(self.kind, self.seed, self.env_id0 + e, ep, t + 1, self.obs_elems))
            rst.append(observation(self.kind, self.seed, self.env_id0 + e, ep + 1, 0, self.obs_elems)
                       if d else nxt[-1])
            rew.append(reward(self.kind, self.seed, self.env_id0 + e, ep, t))
            done.append(d)
            if d:
                self.ep[e] += 1
                self.t[e] = 0
            else:
                self.t[e] += 1
        return np.stack(nxt), np.stack(rst), np.array(rew, dtype=np.float32), np.array(done)
