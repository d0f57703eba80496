"""Auto-pinning hooks for the oracle parts DESIGN.md §6 lists as "parity unpinned": they compare the oracle with the
REAL third-party package the reference calls — gym (CartPole, the Atari wrappers over ALE), scikit-image (the rescale
filter) — whenever that package imports, and skip otherwise.  None of the three is installable in the build container
or on the GPU boxes of this pool (no network), so these tests are skipped there; on the first machine that has the
package they run with no further work and turn SURVEY.md §8 row f3 / the a4 footnote green (or say exactly where the
restatement is off).

What each one pins:
  oracle/cartpole.py      <- gym.envs.classic_control.CartPoleEnv under TimeLimit (the env of presets/CartPole_DQN.py,
                             environments/gym_environment.py:285,418-474): dynamics bit for bit from the same start
                             state, termination step, time limit
  oracle/filters.py       <- skimage.transform.resize(obs, shape, anti_aliasing=False, preserve_range=True)
                             .astype('uint8') — the reference's call, filters/observation/
                             observation_rescale_to_size_filter.py:62-79
  oracle/frontend.py      <- the reference's own wrapper code (environments/gym_environment.py:148-177,418-474) over a
                             real ALE game; needs gym + an Atari ROM + /root/reference (the wrapper classes are the
                             reference's, imported, never copied)"""
import os
import random
import sys

import numpy as np
import pytest


def _gym():
    try:
        import gym
        return gym
    except Exception:
        try:
            import gymnasium as gym          # the maintained fork: same CartPoleEnv arithmetic
            return gym
        except Exception:
            return None


@pytest.mark.parametrize("env_name", ["CartPole-v0", "CartPole-v1"])
def test_cartpole_oracle_against_gym(env_name):
    gym = _gym()
    if gym is None:
        pytest.skip("gym is not installed: oracle/cartpole.py stays unpinned against it (DESIGN.md §6)")
    from oracle.cartpole import MAX_EPISODE_STEPS, CartPole
    env = gym.make(env_name)
    inner = env.unwrapped
    rng = np.random.RandomState(0)
    for episode in range(5):
        out = env.reset()
        obs = out[0] if isinstance(out, tuple) else out                  # gym >= 0.26 returns (obs, info)
        o = CartPole(0, 0, MAX_EPISODE_STEPS[env_name])
        o.reset()
        # gym draws the start state from its own generator (the one deliberate difference, oracle/cartpole.py header):
        # start the oracle from gym's state, at full fp64 where the env exposes it
        start = np.asarray(inner.state if getattr(inner, "state", None) is not None else obs, dtype=np.float64)
        o.state = [float(x) for x in start]
        for t in range(MAX_EPISODE_STEPS[env_name] + 5):
            # a policy that keeps the pole up for a while, so that both the angle limit and the time limit are reached
            a = int(o.state[2] + 0.5 * o.state[3] > 0) if episode % 2 == 0 else int(rng.randint(2))
            res = env.step(a)
            if len(res) == 5:                                            # (obs, reward, terminated, truncated, info)
                g_obs, g_r, g_done = res[0], res[1], bool(res[2] or res[3])
            else:
                g_obs, g_r, g_done = res[0], res[1], bool(res[2])
            s, r, d = o.step(a)
            g_state = np.asarray(inner.state, dtype=np.float64)
            np.testing.assert_array_equal(np.asarray(s, dtype=np.float64), g_state,
                                          err_msg="%s episode %d step %d" % (env_name, episode, t))
            np.testing.assert_array_equal(np.asarray(g_obs, dtype=np.float32), np.asarray(s, dtype=np.float32))
            assert float(g_r) == r == 1.0 and g_done == d, (env_name, episode, t, g_done, d)
            if d:
                break
        assert d


@pytest.mark.parametrize("H,W,C,out", [(210, 160, 1, (84, 84)), (210, 160, 3, (84, 84)), (84, 84, 1, (42, 42)),
                                       (50, 70, 1, (84, 84)), (96, 96, 3, (64, 48)), (7, 9, 1, (3, 5)),
                                       (33, 17, 2, (66, 34)), (84, 84, 4, (84, 84)), (240, 256, 3, (84, 84))])
def test_rescale_oracle_against_scikit_image(H, W, C, out):
    try:
        from skimage.transform import resize
    except Exception:
        pytest.skip("scikit-image is not installed: the rescale filter stays pinned to scipy.ndimage.zoom only")
    from oracle.filters import resize_bilinear_u8
    rng = np.random.RandomState(H * 1000 + W)
    img = rng.randint(0, 256, size=(H, W, C) if C > 1 else (H, W)).astype(np.uint8)
    shape = out + (C,) if C > 1 else out
    ref = resize(img, shape, anti_aliasing=False, preserve_range=True).astype('uint8')     # the reference's call (:76-77)
    got = resize_bilinear_u8(img, out)
    import skimage
    ver = tuple(int(x) for x in skimage.__version__.split(".")[:2])
    if ver >= (0, 19):
        np.testing.assert_array_equal(got, ref)
    else:
        # < 0.19 blends by rows in its own warp path: same sampling rule, 1 LSB on < 0.1 % of the pixels
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.reference
@pytest.mark.parametrize("game,train", [("BreakoutNoFrameskip-v4", True), ("BreakoutNoFrameskip-v4", False),
                                        ("PongNoFrameskip-v4", True)])
def test_atari_front_end_oracle_against_the_reference_wrappers(game, train):
    """oracle.frontend.FrontEndOracle over a real ALE game against the reference's GymEnvironment (its own
    MaxOverFramesAndFrameskipEnvWrapper, lives handling, random no-ops, fire): same emulator seed, same host random
    stream, same actions -> identical frames, rewards and episode ends for 2000 agent steps."""
    gym = _gym()
    if gym is None:
        pytest.skip("gym is not installed: oracle/frontend.py stays unpinned against gym / ALE")
    try:
        probe = gym.make(game)
        probe.close()
    except Exception as e:
        pytest.skip("no ALE / ROM for %s: %s" % (game, str(e).splitlines()[0] if str(e) else type(e).__name__))
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    try:
        from rl_coach.base_parameters import VisualizationParameters
        from rl_coach.core_types import RunPhase
        from rl_coach.environments.gym_environment import Atari, GymEnvironment
    except Exception as e:
        pytest.skip("the reference's environment module does not import here: %s" % str(e).splitlines()[0])
    from oracle.frontend import FrontEndOracle

    class Emulator(object):
        """the interface FrontEndOracle asks of an emulator, over the bare gym env (no wrappers of ours)."""

        def __init__(self, seed):
            self.env = gym.make(game)
            self.env.seed(seed)
            self.ale = self.env.unwrapped.ale

        def reset(self):
            return np.asarray(self.env.reset())

        def step(self, a):
            obs, r, done, _ = self.env.step(a)
            return np.asarray(obs), float(r), bool(done)

        def lives(self):
            return self.ale.lives()

        def action_meanings(self):
            return self.env.unwrapped.get_action_meanings()

    params = Atari(level=game)
    params.seed = 5
    params.frame_skip, params.max_over_num_frames, params.random_initialization_steps = 4, 2, 30
    ref_env = GymEnvironment(**params.__dict__, visualization_parameters=VisualizationParameters())
    ref_env.phase = RunPhase.TRAIN if train else RunPhase.TEST
    o = FrontEndOracle(Emulator(5), frame_skip=4, max_over_num_frames=2, random_initialization_steps=30, train=train)
    state = random.getstate()
    ref_env.reset_internal_state(True)
    after_ref = random.getstate()
    random.setstate(state)
    o.reset(True)
    assert random.getstate() == after_ref                  # the same number of draws for the random no-ops
    np.testing.assert_array_equal(o.state, ref_env.state["observation"])
    rng = np.random.RandomState(1)
    n_actions = len(ref_env.action_space.actions)
    for t in range(2000):
        a = int(rng.randint(n_actions))
        resp = ref_env.step(a)
        s, r, d = o.step(a)
        np.testing.assert_array_equal(s, resp.next_state["observation"], err_msg="frame of agent step %d" % t)
        assert r == resp.reward and d == resp.game_over, (t, r, resp.reward, d, resp.game_over)
        if d:
            state = random.getstate()
            ref_env.reset_internal_state()
            after_ref = random.getstate()
            random.setstate(state)
            o.reset()
            assert random.getstate() == after_ref
            np.testing.assert_array_equal(o.state, ref_env.state["observation"])
