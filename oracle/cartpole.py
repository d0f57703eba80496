"""CartPole-v0 / -v1 — CPU restatement of the simulator behind the reference's CartPole presets.  TEST INFRASTRUCTURE.

The reference steps it through rl_coach/environments/gym_environment.py:418-474 (`self.env.step(action)`, `self.env.reset()`
on `gym.make('CartPole-v0')`, :285).  gym==0.12.5 (requirements.txt:10) is a third-party dependency that is NOT vendored
under /root/reference and not installable here, so this module restates its PUBLISHED algorithm:
gym/envs/classic_control/cartpole.py (CartPoleEnv.__init__ constants, `step` with kinematics_integrator = 'euler',
`reset`) under gym/wrappers/time_limit.py (`done = True` once max_episode_steps <= elapsed steps; 200 for -v0, 500
for -v1 — gym/envs/__init__.py registrations).  Python floats and math.sin / math.cos, expression by expression in
gym's order, so the arithmetic is what gym executes.

PARITY UNPINNED against gym itself: the reference's tests hold no CartPole vectors and gym cannot run here; what the
reference does pin is the END-TO-END bar — CartPole_DQN reaches an evaluation reward of 150 within 250 episodes
(presets/CartPole_DQN.py:50-51, tests/test_golden.py:103-170) — which tests/test_cartpole.py runs on the device.

One deliberate difference: gym draws reset states with `self.np_random.uniform(-0.05, 0.05, (4,))` from a per-env
MT19937 seeded by `env.seed()`; here every env's reset state is a pure function of (seed, env id, episode) — Philox4x32-10
words turned into 53-bit uniforms the way numpy's random_sample does, then `low + (high - low) * u` like numpy's uniform —
so that N device envs reset on different steps without a host round trip (coach_amd/csrc/cartpole.hip).
"""
import math

import numpy as np

from .synth_env import philox4x32_10

STREAM_RESET = 2
MAX_EPISODE_STEPS = {"CartPole-v0": 200, "CartPole-v1": 500}


def reset_state(seed, env, episode):
    """4 uniforms in [-0.05, 0.05): words (x, y), (z, w) of two Philox calls -> (a >> 5, b >> 6) -> 53-bit uniforms."""
    c0 = np.array([episode, episode], dtype=np.uint64)
    c1 = np.array([0, 1], dtype=np.uint64)
    r = philox4x32_10(c0, c1, np.zeros(2, np.uint64), np.full(2, STREAM_RESET, np.uint64), seed, np.full(2, env, np.uint64))
    words = [(int(r[0][i]), int(r[1][i]), int(r[2][i]), int(r[3][i])) for i in range(2)]
    out = []
    for x, y, z, w in words:
        for a, b in ((x, y), (z, w)):
            u = ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0
            out.append(-0.05 + (0.05 - -0.05) * u)
    return out


class CartPole:
    """One environment: CartPoleEnv under TimeLimit."""

    def __init__(self, seed, env_id, max_episode_steps=200):
        # CartPoleEnv.__init__
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = (self.masspole + self.masscart)
        self.length = 0.5
        self.polemass_length = (self.masspole * self.length)
        self.force_mag = 10.0
        self.tau = 0.02
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        self.seed, self.env_id, self.max_episode_steps = seed, env_id, max_episode_steps
        self.episode = -1
        self.state = None
        self.elapsed = 0

    def reset(self):
        self.episode += 1
        self.state = reset_state(self.seed, self.env_id, self.episode)
        self.elapsed = 0
        return list(self.state)

    def step(self, action):
        assert action in (0, 1)
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot * theta_dot * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / \
            (self.length * (4.0 / 3.0 - self.masspole * costheta * costheta / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = [x, x_dot, theta, theta_dot]
        done = x < -self.x_threshold or x > self.x_threshold or theta < -self.theta_threshold_radians \
            or theta > self.theta_threshold_radians
        self.elapsed += 1                                      # TimeLimit.step
        if self.max_episode_steps <= self.elapsed:
            done = True
        return list(self.state), 1.0, bool(done)


class CartPoleVecEnv:
    """N independent CartPoles stepped together, auto-reset like the device vector env: step() returns the stepped
    states, the new episodes' first states where an episode ended (else None), rewards, dones."""

    kind, takes_actions = 1, True                       # what the oracle agent loops (oracle/agents.py) ask an env

    def __init__(self, n_env, seed, env_id0=0, max_episode_steps=200, f32_obs=False):
        """f32_obs: hand out the observations as the device environment does — the fp64 state rounded to fp32 (the
        width the networks read; the reference's filters see gym's fp64 array, a 1e-8 relative difference)."""
        self.n_env = n_env
        self.dt = np.float32 if f32_obs else np.float64
        self.envs = [CartPole(seed, env_id0 + e, max_episode_steps) for e in range(n_env)]

    def reset(self):
        return np.array([e.reset() for e in self.envs], dtype=np.float64).astype(self.dt)

    def step(self, actions):
        nxt, rst, rew, done = [], [], [], []
        for env, a in zip(self.envs, actions):
            s, r, d = env.step(int(a))
            nxt.append(s)
            rew.append(r)
            done.append(d)
            rst.append(np.array(env.reset(), dtype=np.float64).astype(self.dt) if d else None)
        return np.array(nxt, dtype=np.float64).astype(self.dt), rst, np.array(rew, dtype=np.float64), \
            np.array(done, dtype=bool)
