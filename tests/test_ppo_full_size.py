"""Clipped PPO at the FULL size of BASELINE config C2 — 64 envs x 32 steps of 84x84x4 uint8, one rollout of 2048
transitions, minibatches of 64 through the two convolutional towers — device against the numpy oracle agent
(rl_coach/agents/clipped_ppo_agent.py:157-207 fill_advantages, :209-308 train_network): the sampled actions of the
whole rollout, V(s) / GAE / standardised advantages of all 2048 transitions, then the first three minibatch updates
(losses, gradient norm, every weight).  The oracle's layer arithmetic is restated from the TF graph and is unpinned
against TF's own rounding; the tolerances are the ones DESIGN.md §6 states."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ENV, L, PLAYING, BATCH, N_ACTIONS, MINIBATCHES = 64, 32, 2048, 64, 6, 3
from tolerances import ADVANTAGE, LOSS, OUT, WEIGHTS


def _err(name, a, b):
    """worst absolute / relative deviation, printed (pytest -s) so that the stated tolerances can be audited."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), 1e-12)
    print("  %-34s max abs %.3e   max rel (|ref| > 1e-3) %.3e" % (name, d.max(), rel[np.abs(b) > 1e-3].max()
                                                                 if (np.abs(b) > 1e-3).any() else 0.0))


def test_c2_rollout_advantages_and_first_updates_match_the_oracle(rlx, dev):
    import torch
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from coach_amd.core_types import EnvironmentSteps
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    env = SyntheticVectorEnvironment(SyntheticVectorEnvironmentParameters("image", N_ENV, (84, 84), N_ACTIONS,
                                                                          episode_length=L, seed=1234), dev)
    ap = ClippedPPOAgentParameters()
    ap.seed = 0
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAYING)
    ap.network_wrappers["main"].batch_size = BATCH
    agent = ClippedPPOAgent(ap, env, dev, use_graphs=False)
    net = agent.networks["main"]
    o = ClippedPPOAgentOracle(net.params.named_arrays(), SynthVecEnv(0, N_ENV, 84 * 84, L, 1234), N_ACTIONS,
                              batch_size=BATCH, playing_steps=PLAYING, epochs=1)
    o.reset((84, 84))
    start = (random.getstate(), np.random.get_state())

    # ---- device rollout: 32 vector steps = 2048 transitions
    dev_actions = []
    for _ in range(L):
        agent.act()
        dev_actions.append(agent.actions.cpu().numpy().copy())
    assert agent._should_train()
    net.update_target(1.0)                                        # networks['main'].sync() (:326)
    agent.fill_advantages()
    n = agent.memory.num_transitions()
    assert n == PLAYING
    order = list(range(n))
    random.shuffle(order)                                         # shuffle(dataset) (:332)
    batch_order = list(range(n))
    random.shuffle(batch_order)                                   # Batch.shuffle of epoch 0
    order = [order[i] for i in batch_order]
    full = np.zeros(agent.perm_dev.numel(), dtype=np.int32)
    full[:n] = order
    agent._perm.push(full)
    net.set_clip_rescaler(float(ap.algorithm.clipping_decay_schedule.current_value))
    epoch = agent._gather_epoch(n)
    dev_losses, dev_norms, dev_weights = [], [], []
    for i in range(MINIBATCHES):
        agent.scalar_acc.zero_()
        agent._minibatch_fb(BATCH, None, i=i, epoch=epoch)
        agent._minibatch_finish(1.0)
        sc = agent.scalar_acc.cpu().numpy().astype(np.float64)
        dev_losses.append(sc[:5])
        dev_norms.append(sc[5])
        dev_weights.append(net.params.named_arrays())
    agent.check_status()
    dev_random = random.getstate()

    # ---- oracle, same host streams
    random.setstate(start[0]); np.random.set_state(start[1])
    for t in range(L):
        oa, _ = o.act()
        assert np.array_equal(np.asarray(oa), dev_actions[t]), "sampled actions differ at vector step %d" % t
    # the oracle's weights and gradient norm after EACH of its first minibatch updates
    snaps, norms = [], []
    onet = o.net
    orig = onet.train_minibatch

    def recording(*a, **kw):
        r = orig(*a, **kw)
        snaps.append({k: {t: w.copy() for t, w in v.items()} for k, v in onet.weights().items()})
        norms.append(r["norm"])
        return r
    onet.train_minibatch = recording
    ep = o.train(max_minibatches=MINIBATCHES)
    assert random.getstate() == dev_random                        # the same two shuffles

    _err("V(s), 2048 states", agent.ds_value[:n].cpu().numpy(), o.dbg["values"])
    _err("standardised advantages", agent.ds_adv[:n].cpu().numpy(), o.dbg["adv"])
    _err("value targets", agent.ds_vtarget[:n].cpu().numpy(), o.dbg["vt"])
    for i in range(MINIBATCHES):
        _err("losses, minibatch %d" % i, dev_losses[i], np.array(ep[i]))
        _err("gradient norm, minibatch %d" % i, dev_norms[i], norms[i])
        worst = max((float(np.max(np.abs(dev_weights[i][name][tw] - ref))), name) for name, pt in snaps[i].items()
                    for tw, ref in pt.items())
        print("  weights after minibatch %d: worst abs deviation %.3e (%s)" % (i, worst[0], worst[1]))
    # fill_advantages over the whole rollout (dataset order: env-major)
    np.testing.assert_allclose(agent.ds_value[:n].cpu().numpy(), o.dbg["values"], **OUT)
    np.testing.assert_array_equal(agent.ds_reward[:n].cpu().numpy().astype(np.float64), o.dbg["rewards"])
    np.testing.assert_array_equal(agent.ds_done[:n].cpu().numpy().astype(bool), o.dbg["dones"])
    np.testing.assert_array_equal(agent.ds_action[:n].cpu().numpy(), o.dbg["actions"])
    np.testing.assert_allclose(agent.ds_adv[:n].cpu().numpy(), o.dbg["adv"], **ADVANTAGE)
    np.testing.assert_allclose(agent.ds_vtarget[:n].cpu().numpy(), o.dbg["vt"], **OUT)
    for i in range(MINIBATCHES):
        np.testing.assert_allclose(dev_losses[i], np.array(ep[i]), err_msg="losses of minibatch %d" % i, **LOSS)
        np.testing.assert_allclose(dev_norms[i], norms[i], rtol=LOSS["rtol"], err_msg="gradient norm of minibatch %d" % i)
        for name, per_tower in snaps[i].items():
            for tw, ref in per_tower.items():
                np.testing.assert_allclose(dev_weights[i][name][tw], ref, err_msg="%s after minibatch %d" % (name, i), **WEIGHTS)
