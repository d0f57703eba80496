#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own code (/root/reference, read-only)
under the stub-import harness (_refstub.py).  Run from the repo root in the build container:

    python tests/golden/make_golden.py            # regenerates every fixture
    python tests/golden/make_golden.py per gae    # only the named groups

The fixtures are small, committed, and are what travels to the GPU box (the reference does not).
Every group seeds `random` / `np.random` explicitly and stores the seeds it used.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402

_refstub.install()

from rl_coach.core_types import Transition  # noqa: E402
from rl_coach.memories.memory import MemoryGranularity  # noqa: E402
from rl_coach.schedules import ConstantSchedule  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (path, len(arrays), os.path.getsize(path) / 1024))


def _transition(i, game_over=False):
    return Transition(state={'observation': np.array([i])}, action=0, reward=float(i),
                      next_state={'observation': np.array([i + 1])}, game_over=game_over)


# ----------------------------------------------------------------------------------------- PER
def gen_per():
    """PrioritizedExperienceReplay store / sample / update_priorities traces
    (memories/non_episodic/prioritized_experience_replay.py:188-283)."""
    from rl_coach.memories.non_episodic.prioritized_experience_replay import \
        PrioritizedExperienceReplay

    out = {}
    cases = [  # (name, max_size, alpha, beta, batch, n_initial_stores, rounds, stores_per_round)
        ("c8", 8, 0.6, 0.4, 4, 8, 6, 1),
        ("c50", 50, 0.6, 0.4, 8, 40, 12, 3),        # non power of two -> 64; wraps the ring
        ("c1024", 1024, 0.7, 0.5, 32, 1024, 10, 4),
        ("c16k", 1 << 14, 0.6, 0.4, 32, 1 << 14, 6, 4),
    ]
    for name, max_size, alpha, beta, batch, n0, rounds, spr in cases:
        seed = 1000 + max_size
        random.seed(seed)
        np.random.seed(seed)
        m = PrioritizedExperienceReplay((MemoryGranularity.Transitions, max_size), alpha=alpha,
                                        beta=ConstantSchedule(beta))
        cap = m.power_of_2_size
        for i in range(n0):
            m.store(_transition(i))
        # give the leaves non-trivial priorities first
        init_err = np.abs(np.random.randn(min(n0, cap)))
        m.update_priorities(list(range(len(init_err))), list(init_err))
        rec = dict(u=[], idx=[], w=[], err=[], sum_root=[], min_root=[], maxp=[], ntrans=[])
        for r in range(rounds):
            for s in range(spr):
                m.store(_transition(n0 + r * spr + s))
            state = random.getstate()
            u = [random.random() for _ in range(batch)]     # the draws random.uniform will make
            random.setstate(state)
            rec['ntrans'].append(m.num_transitions())
            b = m.sample(batch)
            idx = [t.info['idx'] for t in b]
            w = [t.info['weight'] for t in b]
            err = np.abs(np.random.randn(batch)) * (10.0 if r % 3 == 0 else 1.0)
            if r == 1:
                err[0] = 0.0                                   # epsilon-only priority
            m.update_priorities(idx, list(err))
            rec['u'].append(u); rec['idx'].append(idx); rec['w'].append(w); rec['err'].append(err)
            rec['sum_root'].append(m.sum_tree.total_value())
            rec['min_root'].append(m.min_tree.total_value())
            rec['maxp'].append(m.maximal_priority)
        out[name + "_meta"] = np.array([max_size, cap, batch, n0, rounds, spr, seed], dtype=np.int64)
        out[name + "_ab"] = np.array([alpha, beta, m.epsilon], dtype=np.float64)
        out[name + "_init_err"] = init_err
        for k, v in rec.items():
            out[name + "_" + k] = np.array(v)
        keep = min(len(m.sum_tree.tree), 2047)      # big trees: top 11 levels only
        out[name + "_sum_tree"] = m.sum_tree.tree[:keep].copy()
        out[name + "_min_tree"] = m.min_tree.tree[:keep].copy()
        out[name + "_max_tree"] = m.max_tree.tree[:keep].copy()
    # SURVEY Appendix B vector (random.seed(123), cap 8)
    m = PrioritizedExperienceReplay((MemoryGranularity.Transitions, 8), alpha=0.6,
                                    beta=ConstantSchedule(0.4))
    for i in range(8):
        m.store(_transition(i))
    m.update_priorities(range(8), [.1, .5, 1, 2, 0, .3, 4, .7])
    random.seed(123)
    u = [random.random() for _ in range(4)]
    random.seed(123)
    b = m.sample(4)
    out["appB_u"] = np.array(u)
    out["appB_idx"] = np.array([t.info['idx'] for t in b])
    out["appB_w"] = np.array([t.info['weight'] for t in b])
    out["appB_roots"] = np.array([m.sum_tree.total_value(), m.min_tree.total_value(),
                                  m.maximal_priority])
    # double-store quirk (:271,:280): 5 stores on capacity 8 -> num_transitions() == 8
    m = PrioritizedExperienceReplay((MemoryGranularity.Transitions, 8))
    counts = []
    for i in range(6):
        m.store(_transition(i))
        counts.append(m.num_transitions())
    out["quirk_counts"] = np.array(counts)
    _save("per", **out)


# ------------------------------------------------------------------------------ uniform replay
def gen_er():
    """ExperienceReplay store/sample index traces (memories/non_episodic/experience_replay.py:71-150)."""
    from rl_coach.memories.non_episodic.experience_replay import ExperienceReplay
    out = {}
    # 50 stored (reward = i), np.random.seed(7), sample(8)
    m = ExperienceReplay((MemoryGranularity.Transitions, 1000))
    for i in range(50):
        m.store(_transition(i))
    np.random.seed(7)
    out["appB_rewards"] = np.array([t.reward for t in m.sample(8)])
    # FIFO eviction: capacity 32, 100 stores interleaved with samples
    np.random.seed(11)
    m = ExperienceReplay((MemoryGranularity.Transitions, 32))
    sampled, counts = [], []
    for i in range(100):
        m.store(_transition(i))
        if i % 7 == 6:
            counts.append(m.num_transitions())
            sampled.append([t.reward for t in m.sample(16)])
    out["fifo_meta"] = np.array([32, 100, 16, 7, 11])
    out["fifo_counts"] = np.array(counts)
    out["fifo_rewards"] = np.array(sampled)
    _save("er", **out)


def gen_episodic():
    """EpisodicExperienceReplay.store_episode / close_last_episode / _enforce_max_length / sample
    (memories/episodic/episodic_experience_replay.py:102-130,240-317) with Episode's n-step discounted returns
    (core_types.py:771-820): episodes of different lengths handed over one at a time (what Agent.handle_episode_ended
    does with current_episode_buffer), whole-episode eviction once max_size transitions are exceeded, uniform draws
    over the flat transitions list.  Stored per variant: the rewards, the episode lengths, after EVERY store_episode
    the memory's counters, the rewards / n_step_discounted_rewards of all transitions it holds, and one sampled batch."""
    from rl_coach.core_types import Episode
    from rl_coach.memories.episodic.episodic_experience_replay import EpisodicExperienceReplay
    out = {}
    for name, n_step, max_size, lens in (("to_end", -1, (MemoryGranularity.Transitions, 40), [5, 9, 1, 12, 7, 3, 16, 4, 8]),
                                         ("n3", 3, (MemoryGranularity.Transitions, 30), [6, 2, 11, 4, 9, 5, 13]),
                                         ("episodes", -1, (MemoryGranularity.Episodes, 3), [4, 6, 2, 8, 3])):
        rng = np.random.RandomState(len(lens))
        np.random.seed(100 + len(lens))
        m = EpisodicExperienceReplay(max_size, n_step=n_step)
        rewards = rng.uniform(-2, 2, size=sum(lens))
        counters, held_r, held_nsr, sampled = [], [], [], []
        k = 0
        for L in lens:
            ep = Episode(discount=0.97, n_step=n_step)
            for t in range(L):
                ep.insert(_transition(float(rewards[k]), game_over=(t == L - 1)))
                k += 1
            m.store_episode(ep)
            counters.append([m.num_transitions(), m.num_transitions_in_complete_episodes(), m.num_complete_episodes(),
                             m.length()])
            held_r.append(np.array([t.reward for t in m.transitions]))
            held_nsr.append(np.array([t.n_step_discounted_rewards for t in m.transitions]))
            sampled.append(np.array([t.reward for t in m.sample(6)]))
        out[name + "_meta"] = np.array([n_step, int(max_size[0] == MemoryGranularity.Episodes), max_size[1], 100 + len(lens)])
        out[name + "_lens"], out[name + "_rewards"] = np.array(lens), rewards
        out[name + "_counters"] = np.array(counters)
        out[name + "_sampled"] = np.array(sampled)
        for i, (a, b) in enumerate(zip(held_r, held_nsr)):
            out["%s_held_r_%d" % (name, i)], out["%s_held_nsr_%d" % (name, i)] = a, b
    _save("episodic", **out)


# ------------------------------------------------------------------------------------ stacking
def gen_stack():
    """ObservationStackingFilter traces (filters/observation/observation_stacking_filter.py:89-101)."""
    from rl_coach.filters.observation.observation_stacking_filter import ObservationStackingFilter
    rng = np.random.RandomState(5)
    out = {}
    for name, hw, stack, ep_lens in (("s4", (6, 8), 4, [5, 1, 2, 7]), ("s3", (4, 4), 3, [4, 3])):
        f = ObservationStackingFilter(stack)
        frames, stacks, first = [], [], []
        for L in ep_lens:
            f.reset() if hasattr(f, "reset") else None
            f.stack = []                                     # Filter.reset / InputFilter.reset (:380-390)
            for t in range(L + 1):                           # L transitions -> L+1 observations
                fr = rng.randint(0, 256, size=hw).astype(np.uint8)
                frames.append(fr)
                first.append(t == 0)
                stacks.append(np.array(f.filter(fr)))
        out[name + "_frames"] = np.array(frames)
        out[name + "_first"] = np.array(first)
        out[name + "_stacks"] = np.array(stacks)
        out[name + "_eplens"] = np.array(ep_lens)
    _save("stack", **out)


# ------------------------------------------------------------------------------------- filters
def gen_filters():
    from rl_coach.filters.observation.observation_rgb_to_y_filter import ObservationRGBToYFilter
    from rl_coach.filters.observation.observation_to_uint8_filter import ObservationToUInt8Filter
    from rl_coach.filters.reward.reward_clipping_filter import RewardClippingFilter
    from rl_coach.filters.reward.reward_rescale_filter import RewardRescaleFilter
    from rl_coach.utilities.shared_running_stats import NumpySharedRunningStats
    rng = np.random.RandomState(21)
    out = {}
    rgb = rng.randint(0, 256, size=(3, 84, 84, 3)).astype(np.uint8)
    rgb[0, :4, :4] = 255                                      # saturate: exercises 254.97 -> 254
    rgb[0, 4:8, :4] = 0
    y = np.array([ObservationRGBToYFilter().filter(im) for im in rgb])
    u8 = np.array([ObservationToUInt8Filter(0, 255).filter(v.copy()) for v in y])
    out["rgb"], out["y"], out["y_u8"] = rgb, y, u8
    rewards = np.array([-7.5, -1.0, -0.2, 0.0, 0.3, 1.0, 4.0])
    for name, lo, hi in (("clip11", -1, 1), ("clip0hi", -2, 0), ("clip0lo", 0, 2)):
        flt = RewardClippingFilter(lo, hi)
        out[name] = np.array([flt.filter(r) for r in rewards])
        out[name + "_bounds"] = np.array([lo, hi], dtype=np.float64)
    out["rewards"] = rewards
    out["rescale5"] = np.array([RewardRescaleFilter(5).filter(r) for r in rewards])
    # running stats: Appendix B vector and a 3-push random trace (fp32 observations as the env emits)
    st = NumpySharedRunningStats(name="golden")
    st.set_params(shape=(2,), clip_values=(-5, 5))
    st.push_val(np.array([[1, 2], [3, 6], [5, 10]]))
    out["rsB_mean"], out["rsB_std"] = st.mean.copy(), st.std.copy()
    out["rsB_norm"] = st.normalize(np.array([[1.0, 2.0]]))
    st = NumpySharedRunningStats(name="golden2")
    st.set_params(shape=(17,), clip_values=(-5, 5))
    pushes = [rng.randn(n, 17).astype(np.float32) * (1 + 3 * k) + k for k, n in enumerate((64, 1, 2048))]
    means, stds, sums, sqs, cnts = [], [], [], [], []
    for p_ in pushes:
        st.push_val(p_)
        means.append(st.mean.copy()); stds.append(st.std.copy())
        sums.append(st._sum.copy()); sqs.append(st._sum_squares.copy()); cnts.append(st._count)
    for k, p_ in enumerate(pushes):
        out["rs_push%d" % k] = p_
    out["rs_mean"], out["rs_std"] = np.array(means), np.array(stds)
    out["rs_sum"], out["rs_sq"], out["rs_count"] = np.array(sums), np.array(sqs), np.array(cnts)
    probe = rng.randn(33, 17).astype(np.float32) * 30
    out["rs_probe"] = probe
    out["rs_norm"] = st.normalize(probe)
    _save("filters", **out)


# ----------------------------------------------------------------------------- GAE and returns
class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def gen_gae():
    """ClippedPPOAgent.fill_advantages (agents/clipped_ppo_agent.py:157-207) driven with a stand-in
    value network, ActorCriticAgent.get_general_advantage_estimation_values
    (agents/actor_critic_agent.py:111-125) and Episode.update_discounted_rewards
    (core_types.py:771-801)."""
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent
    from rl_coach.agents.policy_optimization_agent import PolicyGradientRescaler
    from rl_coach.core_types import Batch, Episode
    out = {}

    class Fake(ClippedPPOAgent):
        def __init__(self, values, discount, lam, batch_size):
            self._values = values
            wrapper = _Obj(input_embedders_parameters={'observation': None}, batch_size=batch_size)
            self.ap = _Obj(network_wrappers={'main': wrapper},
                           algorithm=_Obj(discount=discount, gae_lambda=lam,
                                          estimate_state_value_using_gae=True))
            fake = self

            class Net(object):
                def predict(self_inner, inputs):
                    idx = inputs['observation'][:, 0].astype(np.int64)
                    return [fake._values[idx]]
            self.networks = {'main': _Obj(online_network=Net())}
            self.state_values = _Obj(add_sample=lambda v: None)
            self.action_advantages = _Obj(add_sample=lambda v: None)
            self.policy_gradient_rescaler = PolicyGradientRescaler.GAE

    # Appendix B
    f = Fake(None, 0.99, 0.95, 64)
    gae, vt = f.get_general_advantage_estimation_values(np.array([1., 0., 2., -1.]),
                                                        np.array([.5, .6, .7, .2, 0.]))
    out["appB_gae"], out["appB_vt"] = gae, vt[:, 0]
    rng = np.random.RandomState(99)
    cases = (("small", [5, 1, 3], 0.99, 0.95, 4), ("ppo2048", [32] * 64, 0.99, 0.95, 64),
             ("ragged", [1000, 1, 257, 64, 726], 0.995, 0.9, 64))
    for name, ep_lens, disc, lam, bs in cases:
        T = sum(ep_lens)
        rewards = rng.randn(T).astype(np.float32)
        if name == "ppo2048":
            rewards = rng.choice([-1.0, 0.0, 1.0], size=T, p=[.05, .9, .05]).astype(np.float32)
        values = rng.randn(T).astype(np.float32)
        go = np.zeros(T, dtype=bool)
        go[np.cumsum(ep_lens) - 1] = True
        trans = [Transition(state={'observation': np.array([i], dtype=np.float64)}, action=0,
                            reward=float(rewards[i]), next_state={'observation': np.array([i + 1.])},
                            game_over=bool(go[i])) for i in range(T)]
        for t in trans:
            t.n_step_discounted_rewards = 0.0
        f = Fake(values, disc, lam, bs)
        f.fill_advantages(Batch(trans))
        out[name + "_rewards"], out[name + "_values"], out[name + "_go"] = rewards, values, go
        out[name + "_hp"] = np.array([disc, lam])
        out[name + "_adv"] = np.array([t.info['advantage'] for t in trans])
        out[name + "_vt"] = np.array([t.info['gae_based_value_target'] for t in trans])
        # discounted returns per episode from the reference Episode class
        rets, start = [], 0
        for L in ep_lens:
            ep = Episode(discount=disc)
            for i in range(start, start + L):
                ep.insert(trans[i])
            ep.update_discounted_rewards()
            rets += [t.n_step_discounted_rewards for t in ep.transitions]
            start += L
        out[name + "_returns"] = np.array(rets)
    ep = Episode(discount=0.99)
    for r in [1., 0., 2., -1.]:
        ep.insert(Transition(state={'observation': np.zeros(1)}, action=0, reward=r,
                             next_state={'observation': np.zeros(1)}, game_over=False))
    ep.update_discounted_rewards()
    out["appB_returns"] = np.array([t.n_step_discounted_rewards for t in ep.transitions])
    _save("gae", **out)


# ------------------------------------------------------------------------------------- targets
def gen_targets():
    """learn_from_batch target arithmetic of DQN / DDQN / DDPG / TD3 with stand-in networks."""
    from unittest import mock
    from rl_coach.agents.dqn_agent import DQNAgent
    from rl_coach.agents.ddqn_agent import DDQNAgent
    from rl_coach.agents.ddpg_agent import DDPGAgent
    from rl_coach.agents.td3_agent import TD3Agent
    from rl_coach.core_types import Batch
    from rl_coach.spaces import BoxActionSpace
    rng = np.random.RandomState(31)
    out = {}

    def make_batch(B, obs_dim, actions, rewards, go):
        tr = [Transition(state={'observation': rng.randn(obs_dim)}, action=actions[i],
                         reward=float(rewards[i]), next_state={'observation': rng.randn(obs_dim)},
                         game_over=bool(go[i])) for i in range(B)]
        return Batch(tr)

    for cls, name in ((DQNAgent, "dqn"), (DDQNAgent, "ddqn")):
        B, A = 32, 4
        q_next_t = rng.randn(B, A).astype(np.float32)
        q_next_o = rng.randn(B, A).astype(np.float32)
        q_onl = rng.randn(B, A).astype(np.float32)
        q_next_t[3] = q_next_t[3, 0]                          # ties -> first argmax
        actions = rng.randint(0, A, size=B)
        rewards = rng.choice([-1.0, 0.0, 1.0], size=B).astype(np.float32)
        go = rng.rand(B) < 0.2
        captured = {}

        class Fake(cls):
            def __init__(self):
                pass
        f = Fake()
        wrapper = _Obj(input_embedders_parameters={'observation': None})
        f.ap = _Obj(network_wrappers={'main': wrapper}, algorithm=_Obj(discount=0.99))
        f.q_values = _Obj(add_sample=lambda v: None)
        f.memory = object()
        online = _Obj(predict=lambda inputs: q_next_o.copy())     # DDQN: online net on next states
        main = _Obj(target_network=object(), online_network=online,
                    parallel_prediction=lambda pairs: (q_next_t.copy(), q_onl.copy()))

        def train(states, targets, importance_weights=None):
            captured['targets'] = np.array(targets)
            return 0.0, [0.0], 0.0
        main.train_and_sync_networks = train
        f.networks = {'main': main}
        orig = f.update_transition_priorities_and_get_weights

        def upd(errs, batch):
            captured['errors'] = np.array(errs, dtype=np.float64)
            return None
        f.update_transition_priorities_and_get_weights = upd
        f.learn_from_batch(make_batch(B, 4, actions, rewards, go))
        out[name + "_q_next_t"], out[name + "_q_next_o"], out[name + "_q_onl"] = q_next_t, q_next_o, q_onl
        out[name + "_actions"], out[name + "_rewards"], out[name + "_go"] = actions, rewards, go
        out[name + "_targets"], out[name + "_errors"] = captured['targets'], captured['errors']

    # DDPG / TD3 TD targets
    for cls, name in ((DDPGAgent, "ddpg"), (TD3Agent, "td3")):
        B, A = 100, 6
        next_actions = np.tanh(rng.randn(B, A)).astype(np.float32)
        q_next = rng.randn(B, 1).astype(np.float32) * 5
        rewards = rng.randn(B).astype(np.float32)
        go = rng.rand(B) < 0.1
        captured = {}

        class Fake(cls):
            def __init__(self):
                pass
        f = Fake()
        f.ap = _Obj(network_wrappers={'actor': _Obj(input_embedders_parameters={'observation': None}),
                                      'critic': _Obj(input_embedders_parameters={'observation': None,
                                                                                 'action': None})},
                    algorithm=_Obj(discount=0.99, use_non_zero_discount_for_terminal_states=False,
                                   clip_critic_targets=(-3.0, 3.0) if name == "ddpg" else None,
                                   policy_noise=0.2, noise_clipping=0.5,
                                   update_policy_every_x_episode_steps=2))
        f.TD_targets_signal = _Obj(add_sample=lambda v: None)
        f.training_iteration = 1                              # odd -> TD3 skips the actor update
        f.spaces = _Obj(action=BoxActionSpace(A, -0.8, 0.8))
        actor = mock.MagicMock()
        actor.parallel_prediction = lambda pairs: (next_actions.copy(), next_actions.copy())
        actor.has_global = False
        critic = mock.MagicMock()

        def critic_target_predict(inputs):
            captured['critic_next_action'] = np.array(inputs['action'])
            return [q_next, q_next, q_next, q_next]
        critic.target_network.predict = critic_target_predict
        critic.online_network.predict = lambda *a, **k: np.zeros((B, A), dtype=np.float32)

        def train(inputs, targets, **kw):
            captured['targets'] = np.array(targets)
            return 0.0, [0.0], 0.0
        critic.train_and_sync_networks = train
        actor.online_network.predict = lambda *a, **k: []
        f.networks = {'actor': actor, 'critic': critic}
        np.random.seed(77)
        noise = np.random.normal(0, 0.2, next_actions.shape)
        np.random.seed(77)
        actions = rng.uniform(-0.8, 0.8, size=(B, A))
        f.learn_from_batch(make_batch(B, 17, list(actions), rewards, go))
        out[name + "_next_actions"], out[name + "_q_next"] = next_actions, q_next
        out[name + "_rewards"], out[name + "_go"] = rewards, go
        out[name + "_targets"] = captured['targets']
        if name == "td3":
            out["td3_noise"] = noise
            out["td3_smoothed"] = captured['critic_next_action']
    _save("targets", **out)


# --------------------------------------------------------------------------------- exploration
def gen_explore():
    """Categorical / EGreedy / AdditiveNoise .get_action under a seeded np.random, together with the
    raw draws each call consumed (exploration_policies/{categorical,e_greedy,additive_noise}.py)."""
    from rl_coach.core_types import RunPhase
    from rl_coach.exploration_policies.additive_noise import AdditiveNoise
    from rl_coach.exploration_policies.categorical import Categorical
    from rl_coach.exploration_policies.e_greedy import EGreedy
    from rl_coach.schedules import ConstantSchedule, LinearSchedule
    from rl_coach.spaces import BoxActionSpace, DiscreteActionSpace
    out = {}
    rng = np.random.RandomState(8)
    # --- categorical: one np.random.choice per call
    A = 6
    pol = Categorical(DiscreteActionSpace(A))
    pol.change_phase(RunPhase.TRAIN)
    logits = rng.randn(200, A).astype(np.float32)
    probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    probs = probs.astype(np.float32)
    np.random.seed(3)
    u = np.random.random_sample(200)
    np.random.seed(3)
    out["cat_probs"], out["cat_u"] = probs, u
    out["cat_actions"] = np.array([pol.get_action(p)[0] for p in probs])
    # --- e-greedy (discrete): per call: [choice if exploring | A tie-break randoms], then a new rand()
    A = 4
    pol = EGreedy(DiscreteActionSpace(A), LinearSchedule(0.5, 0.1, 100), 0.05)
    pol.change_phase(RunPhase.TRAIN)
    q = rng.randn(300, A).astype(np.float32)
    q[5] = q[5, 0]                      # full tie
    q[9, 2] = q[9].max()                # partial tie
    np.random.seed(4)
    pol.current_random_value = np.random.rand()
    acts, eps, explore_u, rand_act, tie = [], [], [], [], []
    for i in range(300):
        e = pol.epsilon_schedule.current_value
        eps.append(e); explore_u.append(pol.current_random_value)
        st = np.random.get_state()
        if pol.current_random_value < e:
            rand_act.append(np.random.choice(A)); tie.append(np.zeros(A))
        else:
            rand_act.append(-1); tie.append(np.random.random(A))
        np.random.set_state(st)
        acts.append(pol.get_action(q[i])[0])
    out["eg_q"], out["eg_actions"], out["eg_eps"] = q, np.array(acts), np.array(eps)
    out["eg_explore_u"], out["eg_rand_act"], out["eg_tie"] = np.array(explore_u), np.array(rand_act), np.array(tie)
    # ... then an evaluation: phase TEST uses evaluation_epsilon, the schedule stands still, the draws go on
    pol.change_phase(RunPhase.TEST)
    q2 = np.random.RandomState(88).randn(80, A).astype(np.float32)     # (own stream: the fixtures below keep their values)
    acts, eps, explore_u, rand_act, tie = [], [], [], [], []
    for i in range(80):
        e = pol.get_control_param()
        eps.append(e); explore_u.append(pol.current_random_value)
        st = np.random.get_state()
        if pol.current_random_value < e:
            rand_act.append(np.random.choice(A)); tie.append(np.zeros(A))
        else:
            rand_act.append(-1); tie.append(np.random.random(A))
        np.random.set_state(st)
        acts.append(pol.get_action(q2[i])[0])
    out["egt_q"], out["egt_actions"], out["egt_eps"] = q2, np.array(acts), np.array(eps)
    out["egt_explore_u"], out["egt_rand_act"], out["egt_tie"] = np.array(explore_u), np.array(rand_act), np.array(tie)
    pol.change_phase(RunPhase.TRAIN)
    out["egt_schedule_after"] = np.float64(pol.epsilon_schedule.current_value)
    # --- additive gaussian noise (continuous)
    D = 6
    space = BoxActionSpace(D, -1.0, 1.0)
    pol = AdditiveNoise(space, ConstantSchedule(0.1), 0.05)
    pol.change_phase(RunPhase.TRAIN)
    mean = np.tanh(rng.randn(50, D)).astype(np.float32)
    np.random.seed(6)
    z = np.random.standard_normal((50, D))
    np.random.seed(6)
    out["an_mean"], out["an_z"] = mean, z
    out["an_std"] = 0.1 * (space.high - space.low)
    out["an_actions"] = np.array([pol.get_action(m) for m in mean])
    # ---- OUProcess (ou_process.py:41-77, DDPG's default exploration): correlated noise, restart at reset()
    from rl_coach.exploration_policies.ou_process import OUProcess
    space = BoxActionSpace(3, -1.0, 1.0)
    pol = OUProcess(space, mu=0, theta=0.15, sigma=0.2, dt=0.01)
    pol.change_phase(RunPhase.TRAIN)
    ou_mean = np.tanh(rng.randn(40, 3)).astype(np.float32)
    np.random.seed(8)
    acts = []
    for i, m in enumerate(ou_mean):
        if i == 25:
            pol.reset()                                   # an episode ended: Agent.reset_internal_state
        acts.append(pol.get_action(m).copy())
    out["ou_mean"], out["ou_actions"], out["ou_reset_at"] = ou_mean, np.array(acts), np.array(25)
    _save("explore", **out)


# ------------------------------------------------------------------- whole agent updates
def _rand_arrays(rng, spec):
    """{name: (in, out, towers)} -> {name/kernel|bias: [array per tower]}."""
    out = {}
    for name, (i, o, t) in spec.items():
        out[name + "/kernel"] = [(rng.uniform(-1, 1, (i, o)) * np.sqrt(3.0 / i)).astype(np.float32) for _ in range(t)]
        out[name + "/bias"] = [rng.uniform(-0.1, 0.1, (o,)).astype(np.float32) for _ in range(t)]
    return out


def _flat(prefix, weights, out):
    for name, towers in weights.items():
        for t, arr in towers.items():
            out["%s|%s|%d" % (prefix, name, t)] = np.array(arr)


def gen_updates():
    """The REFERENCE's learn_from_batch of DDPG / TD3 / SAC / DQN (agents/{ddpg,td3,soft_actor_critic,
    dqn}_agent.py) executed here against oracle-backed network stand-ins (_oracle_backend.py): stores the
    initial weights, every batch / noise draw and the weights after three updates.
    tests/test_update_pins.py re-runs the oracle's own whole-update functions from the same inputs."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle import ac_nets as O
    from oracle.agents import DQNOracle
    from rl_coach.agents.ddpg_agent import DDPGAgent
    from rl_coach.agents.dqn_agent import DQNAgent
    from rl_coach.agents.soft_actor_critic_agent import SoftActorCriticAgent
    from rl_coach.agents.td3_agent import TD3Agent
    from rl_coach.core_types import Batch
    from rl_coach.spaces import BoxActionSpace
    out = {}
    sink = _Obj(add_sample=lambda v: None)

    def make_batch(rng, B, D, A, discrete=None):
        s, ns = rng.randn(B, D).astype(np.float32), rng.randn(B, D).astype(np.float32)
        a = rng.randint(0, discrete, size=B) if discrete else rng.uniform(-1, 1, (B, A)).astype(np.float32)
        r = rng.randn(B).astype(np.float32)
        done = rng.rand(B) < 0.15
        tr = [Transition(state={'observation': s[i]}, action=a[i], reward=float(r[i]),
                         next_state={'observation': ns[i]}, game_over=bool(done[i])) for i in range(B)]
        return Batch(tr), (s, a, r, done, ns)

    def fake(cls):
        class Fake(cls):
            def __init__(self):
                pass
        return Fake()

    keys = lambda *names: _Obj(input_embedders_parameters={n: None for n in names})

    # ---------------- DDPG and TD3
    for name, cls, streams, D, A, B in (("ddpg", DDPGAgent, 1, 11, 3, 32), ("td3", TD3Agent, 2, 17, 6, 48)):
        rng = np.random.RandomState(41 + streams)
        spec = {"actor/embedder/dense0": (D, 24, 1), "actor/middleware/dense0": (24, 16, 1),
                "actor/ddpg_actor_head/fc_mean": (16, A, 1)}
        a_arrays = _rand_arrays(rng, spec)
        if streams == 1:
            spec = {"critic/embedder/dense0": (D, 20, 1), "critic/middleware/dense0": (20 + A, 16, 1),
                    "critic/v_head/output": (16, 1, 1)}
        else:
            spec = {"critic/middleware/dense0": (D + A, 24, 2), "critic/middleware/dense1": (24, 16, 2),
                    "critic/v_head/output": (16, 1, 2)}
        c_arrays = _rand_arrays(rng, spec)
        actor = O.ActorOracle(copy.deepcopy(a_arrays), 1.0, lr=1e-3)
        critic = O.CriticOracle(copy.deepcopy(c_arrays), streams=streams, lr=1e-3)
        for k, v in list(a_arrays.items()) + list(c_arrays.items()):
            for t, arr in enumerate(v):
                out["%s|init|%s|%d" % (name, k, t)] = arr
        f = fake(cls)
        f.ap = _Obj(network_wrappers={'actor': keys('observation'), 'critic': keys('observation', 'action')},
                    algorithm=_Obj(discount=0.99, use_non_zero_discount_for_terminal_states=False,
                                   clip_critic_targets=None, policy_noise=0.2, noise_clipping=0.5,
                                   update_policy_every_x_episode_steps=2))
        f.TD_targets_signal = sink
        f.spaces = _Obj(action=BoxActionSpace(A, -1.0, 1.0))
        f.networks = {'actor': OB.ActorWrapper(actor), 'critic': OB.CriticWrapper(critic, A)}
        for it in range(1, 4):
            f.training_iteration = it
            batch, arrays = make_batch(rng, B, D, A)
            np.random.seed(500 + it)
            noise = np.random.normal(0, 0.2, (B, A))            # what TD3's learn_from_batch will draw
            np.random.seed(500 + it)
            res = f.learn_from_batch(batch)
            for k, v in zip(("s", "a", "r", "done", "ns"), arrays):
                out["%s|batch%d|%s" % (name, it, k)] = v
            out["%s|noise%d" % (name, it)] = noise
            out["%s|loss%d" % (name, it)] = np.float64(res[0])
            if name == "ddpg" or it % 2 == 0:                   # the agents' target cadence, simplified
                actor.mix_target(0.01); critic.mix_target(0.01)
        _flat(name + "|final|actor", actor.weights(), out)
        _flat(name + "|final|critic", critic.weights(), out)

    # ---------------- SAC
    D, A, B = 13, 4, 40
    rng = np.random.RandomState(47)
    p_arr = _rand_arrays(rng, {"policy/embedder/dense0": (D, 24, 1), "policy/middleware/dense0": (24, 16, 1),
                               "policy/sac_policy_head/policy_mu_logsig": (16, 2 * A, 1)})
    q_arr = _rand_arrays(rng, {"q/q_head/obs_fc": (D, 16, 2), "q/q_head/act_fc": (A, 16, 2),
                               "q/q_head/fc1": (16, 16, 2), "q/q_head/q_output": (16, 1, 2)})
    v_arr = _rand_arrays(rng, {"v/embedder/dense0": (D, 24, 1), "v/middleware/dense0": (24, 16, 1),
                               "v/v_values_head/output": (16, 1, 1)})
    pol, qn, vn = O.SACPolicyOracle(copy.deepcopy(p_arr)), O.SACQOracle(copy.deepcopy(q_arr)), \
        O.SACValueOracle(copy.deepcopy(v_arr))
    for arrs in (p_arr, q_arr, v_arr):
        for k, v in arrs.items():
            for t, arr in enumerate(v):
                out["sac|init|%s|%d" % (k, t)] = arr
    f = fake(SoftActorCriticAgent)
    f.ap = _Obj(network_wrappers={n: keys('observation') for n in ('policy', 'q', 'v')},
                algorithm=_Obj(discount=0.99))
    for sig in ("policy_means", "policy_logsig", "policy_logprob_sampled", "q1_values", "q2_values",
                "policy_grads", "v_onl_ys", "v_tgt_ns", "TD_err1", "TD_err2"):
        setattr(f, sig, sink)
    pw = OB.SACPolicyWrapper(pol, A)
    f.networks = {'policy': pw, 'q': OB.SACQWrapper(qn), 'v': OB.SACValueWrapper(vn)}
    for it in range(1, 4):
        batch, arrays = make_batch(rng, B, D, A)
        np.random.seed(600 + it)
        pw.normals = []
        res = f.learn_from_batch(batch)
        assert len(pw.normals) == 3                              # three policy sess.run passes
        for k, v in zip(("s", "a", "r", "done", "ns"), arrays):
            out["sac|batch%d|%s" % (it, k)] = v
        out["sac|normals%d" % it] = np.stack(pw.normals)
        out["sac|loss%d" % it] = np.float64(res[0])
        vn.mix_target(0.005)
    for nm, net in (("policy", pol), ("q", qn), ("v", vn)):
        _flat("sac|final|" + nm, net.weights(), out)

    # ---------------- DQN (Huber loss, uniform replay: no importance weights)
    D, A, B = 6, 3, 32
    rng = np.random.RandomState(53)
    d_arr = _rand_arrays(rng, {"main/embedder/dense0": (D, 24, 1), "main/middleware/dense0": (24, 16, 1),
                               "main/q_head/dense": (16, A, 1)})
    dq = DQNOracle(copy.deepcopy(d_arr), (D,), A, lr=1e-3)
    for k, v in d_arr.items():
        out["dqn|init|%s|0" % k] = v[0]
    f = fake(DQNAgent)
    f.ap = _Obj(network_wrappers={'main': keys('observation')}, algorithm=_Obj(discount=0.99))
    f.q_values = sink
    f.memory = object()
    f.update_transition_priorities_and_get_weights = lambda errs, batch: None
    f.networks = {'main': OB.DQNWrapper(dq)}
    for it in range(1, 4):
        batch, arrays = make_batch(rng, B, D, A, discrete=A)
        res = f.learn_from_batch(batch)
        for k, v in zip(("s", "a", "r", "done", "ns"), arrays):
            out["dqn|batch%d|%s" % (it, k)] = v
        out["dqn|loss%d" % it] = np.float64(res[0])
        if it == 2:
            dq.update_target(1.0)
    _flat("dqn|final|main", dq.weights(), out)
    _save("updates", **out)


def gen_ppo_update():
    """The REFERENCE's ClippedPPOAgent.train (agents/clipped_ppo_agent.py:314-344: sync, fill_advantages,
    dataset shuffle, train_network with per-epoch Batch.shuffle) executed here on an oracle-backed
    network stand-in; tests/test_update_pins.py re-runs oracle.agents.ClippedPPOAgentOracle.train from
    the same transitions, weights and `random` seed.  Two cases: "ppo" = DiscreteActionSpace (PPOHead softmax),
    "ppoc" = BoxActionSpace (PPOHead mean / std: the old policy travels as two inputs, the actions are vectors)."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import ClippedPPOOracle
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent
    from rl_coach.agents.policy_optimization_agent import PolicyGradientRescaler
    from rl_coach.core_types import EnvironmentSteps
    from rl_coach.spaces import BoxActionSpace, DiscreteActionSpace
    out = {}
    for prefix, continuous, seed0 in (("ppo", False, 61), ("ppoc", True, 62)):
        D, A, B, n_env, L, epochs = (9, 4, 16, 6, 8, 3) if not continuous else (7, 3, 8, 4, 6, 2)
        rng = np.random.RandomState(seed0)
        head = "main/ppo_head/policy_mean" if continuous else "main/ppo_head/policy_fc"
        arrays = _rand_arrays(rng, {"main/embedder/dense0": (D, 24, 2), "main/middleware/dense0": (24, 16, 2),
                                    "main/v_head/dense": (16, 1, 1), head: (16, A, 1)})
        if continuous:
            arrays["main/ppo_head/policy_log_std"] = [rng.uniform(-0.5, 0.1, (A,)).astype(np.float32)]
        net = ClippedPPOOracle(copy.deepcopy(arrays), (D,), A, lr=1e-3, clip_eps=0.2, beta_entropy=0.01,
                               continuous=continuous)
        for k, v in arrays.items():
            for t, arr in enumerate(v):
                out["%s|init|%s|%d" % (prefix, k, t)] = arr
        T = n_env * L
        states = rng.randn(T, D).astype(np.float32)
        actions = rng.uniform(-1, 1, (T, A)).astype(np.float32) if continuous else rng.randint(0, A, size=T)
        rewards = rng.choice([-1.0, 0.0, 1.0], size=T).astype(np.float32)
        go = np.zeros(T, dtype=bool)
        go[np.arange(n_env) * L + L - 1] = True                      # episode-major, one episode per env
        trans = [Transition(state={'observation': states[i]}, action=actions[i] if continuous else int(actions[i]),
                            reward=float(rewards[i]), next_state={'observation': states[min(i + 1, T - 1)]},
                            game_over=bool(go[i]))
                 for i in range(T)]
        for t in trans:
            t.n_step_discounted_rewards = 0.0

        class Fake(ClippedPPOAgent):
            def __init__(self):
                pass
        f = Fake()
        sink = _Obj(add_sample=lambda v: None)
        wrapper = _Obj(input_embedders_parameters={'observation': None}, batch_size=B, learning_rate_decay_rate=0,
                       learning_rate=1e-3)
        f.ap = _Obj(network_wrappers={'main': wrapper},
                    algorithm=_Obj(discount=0.99, gae_lambda=0.95, estimate_state_value_using_gae=True,
                                   update_pre_network_filters_state_on_train=False,
                                   num_consecutive_training_steps=1,
                                   num_consecutive_playing_steps=EnvironmentSteps(T), optimization_epochs=epochs,
                                   clipping_decay_schedule=ConstantSchedule(1)))
        f.policy_gradient_rescaler = PolicyGradientRescaler.GAE
        f.spaces = _Obj(action=BoxActionSpace(A, -1.0, 1.0) if continuous else DiscreteActionSpace(A))
        for sig in ("state_values", "action_advantages", "unclipped_grads", "value_targets", "likelihood_ratio",
                    "clipped_likelihood_ratio", "value_loss", "policy_loss", "loss", "curr_learning_rate", "entropy",
                    "kl_divergence"):
            setattr(f, sig, sink)
        f.memory = _Obj(transitions=trans)
        f.pre_network_filter = _Obj(filter=lambda dataset, deep_copy=False, update_internal_state=False: dataset)
        f._should_train = lambda: True
        f.post_training_commands = lambda: None
        f.update_log = lambda: None
        f.training_iteration = 0
        f.networks = {'main': (OB.PPOContinuousWrapper if continuous else OB.PPOWrapper)(net)}
        random.seed(71)
        f.train()
        out[prefix + "|states"], out[prefix + "|actions"], out[prefix + "|rewards"], out[prefix + "|go"] = \
            states, actions, rewards, go
        out[prefix + "|hp"] = np.array([D, A, B, n_env, L, epochs, 71])
        out[prefix + "|adv"] = np.array([t.info['advantage'] for t in trans])       # (dataset was shuffled in place:
        out[prefix + "|adv_state0"] = np.array([t.state['observation'][0] for t in trans])   # keyed by state[0])
        _flat(prefix + "|final", net.weights(), out)
    _save("ppo_update", **out)


def gen_cadence():
    """Agent.train scheduling (agents/agent.py:640-784: _should_train, num_consecutive_training_steps,
    _should_update_online_weights_to_target) of the REFERENCE, one env stepping: per env-step, how many
    learn_from_batch calls and target copies happen.  tests/test_update_pins.py replays the same step
    sequence through the oracle agent loop (oracle.agents.DQNAgentOracle.train, n_env = 1)."""
    from rl_coach.agents.dqn_agent import DQNAgent
    from rl_coach.core_types import EnvironmentSteps, TrainingSteps
    out = {}
    cases = {"p1_t10": (1, ("env", 10), 1), "p4_t10": (4, ("env", 10), 1), "p3_t7": (3, ("env", 7), 1),
             "p2_tt3_c2": (2, ("train", 3), 2)}
    for name, (playing, (tk, tn), consecutive) in cases.items():
        class Fake(DQNAgent):
            def __init__(self):
                pass

            def call_memory(self, func, args=()):
                return {'num_transitions': self.n_stored, 'sample': [_transition(0)] * 4}[func]

            def learn_from_batch(self, batch):
                self.learned += 1
                return 0.0, [], 0.0
        f = Fake()
        sink = _Obj(add_sample=lambda v: None)
        f.ap = _Obj(is_batch_rl_training=False, visualization=_Obj(dump_csv=False),
                    network_wrappers={'main': _Obj(batch_size=4, learning_rate_decay_rate=0, learning_rate=1e-3)},
                    algorithm=_Obj(act_for_full_episodes=False,
                                   num_consecutive_playing_steps=EnvironmentSteps(playing),
                                   num_consecutive_training_steps=consecutive,
                                   num_steps_between_copying_online_weights_to_target=
                                   EnvironmentSteps(tn) if tk == "env" else TrainingSteps(tn),
                                   rate_for_copying_weights_to_target=1.0,
                                   update_pre_network_filters_state_on_train=False))
        copies = []
        net = _Obj(has_target=True, set_is_training=lambda s: None,
                   update_target_network=lambda rate: copies.append(1))
        f.networks = {'main': net}
        f.pre_network_filter = None
        f.unclipped_grads = f.curr_learning_rate = f.loss = sink
        f.agent_logger = _Obj(create_signal_value=lambda *a, **k: None)
        f.imitation = False
        f.post_training_commands = lambda: None
        f.total_steps_counter = f.last_training_phase_step = f.training_iteration = 0
        f.last_target_network_update_step = 0
        f.learned, f.n_stored = 0, 0
        learned, copied = [], []
        for t in range(1, 61):
            f.total_steps_counter += 1
            f.n_stored += 1
            a, b = f.learned, len(copies)
            f.train()
            learned.append(f.learned - a)
            copied.append(len(copies) - b)
        out[name + "|learned"], out[name + "|copied"] = np.array(learned), np.array(copied)
        out[name + "|cfg"] = np.array([playing, 0 if tk == "env" else 1, tn, consecutive])

    # act_for_full_episodes (Clipped PPO, TD3): a phase opens only when the running episode is complete
    for name, playing, L in (("full_p20_L8", 20, 8), ("full_p16_L16", 16, 16), ("full_p5_L3", 5, 3)):
        class FakeFull(DQNAgent):
            def __init__(self):
                pass

            def call_memory(self, func, args=()):
                return self.n_stored
        f = FakeFull()
        f.ap = _Obj(algorithm=_Obj(act_for_full_episodes=True, num_consecutive_playing_steps=EnvironmentSteps(playing)))
        f.total_steps_counter = f.last_training_phase_step = f.n_stored = 0
        opened = []
        for t in range(1, 81):
            f.total_steps_counter += 1
            f.n_stored += 1
            f.current_episode_buffer = _Obj(is_complete=(t % L == 0))
            opened.append(int(f._should_train()))
        out[name + "|opened"] = np.array(opened)
        out[name + "|cfg"] = np.array([playing, L])

    # episodic memory (DDPG): the memory receives an episode only when it ends, so nothing can be sampled
    # — and no phase opens — before the first episode is complete
    for name, playing, L in (("episodic_p1_L6", 1, 6), ("episodic_p4_L10", 4, 10)):
        class FakeEp(DQNAgent):
            def __init__(self):
                pass

            def call_memory(self, func, args=()):
                return self.n_complete
        f = FakeEp()
        f.ap = _Obj(algorithm=_Obj(act_for_full_episodes=False, num_consecutive_playing_steps=EnvironmentSteps(playing)))
        f.total_steps_counter = f.last_training_phase_step = f.n_complete = 0
        opened = []
        for t in range(1, 41):
            f.total_steps_counter += 1
            f.n_complete = L * (t // L)
            opened.append(int(f._should_train()))
        out[name + "|opened"] = np.array(opened)
        out[name + "|cfg"] = np.array([playing, L])
    _save("cadence", **out)


def gen_defaults():
    """Default hyper-parameters of the reference's AgentParameters classes (algorithm + every network
    wrapper + exploration), instantiated here under the import stubs -> tests/golden/defaults.json.
    tests/test_update_pins.py compares every field the device agents' parameter classes also carry."""
    import json
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgentParameters
    from rl_coach.agents.ddpg_agent import DDPGAgentParameters
    from rl_coach.agents.ddqn_agent import DDQNAgentParameters
    from rl_coach.agents.dqn_agent import DQNAgentParameters
    from rl_coach.agents.soft_actor_critic_agent import SoftActorCriticAgentParameters
    from rl_coach.agents.td3_agent import TD3AgentParameters

    def scalars(o):
        out = {}
        for k, v in (vars(o).items() if hasattr(o, "__dict__") else ()):
            if k.startswith('_'):
                continue
            if isinstance(v, (int, float, bool, str, type(None))):
                out[k] = v
            elif hasattr(v, 'num_steps'):
                out[k] = [type(v).__name__, v.num_steps]
            elif isinstance(v, (tuple, list)) and all(isinstance(x, (int, float)) for x in v):
                out[k] = list(v)
            elif hasattr(v, 'current_value') and hasattr(v, 'initial_value'):
                out[k] = ["schedule", type(v).__name__, float(v.initial_value),
                          float(getattr(v, 'final_value', v.initial_value)),
                          int(getattr(v, 'decay_steps', 0) or 0)]
        return out
    out = {}
    for P in (DQNAgentParameters, DDQNAgentParameters, ClippedPPOAgentParameters, DDPGAgentParameters,
              TD3AgentParameters, SoftActorCriticAgentParameters):
        ap = P()
        out[P.__name__] = {"algorithm": scalars(ap.algorithm),
                           "networks": {n: scalars(w) for n, w in ap.network_wrappers.items()},
                           "exploration": dict(scalars(ap.exploration), **{"class": type(ap.exploration).__name__}),
                           "memory": dict(scalars(ap.memory), **{"class": type(ap.memory).__name__})}
    path = os.path.join(HERE, "defaults.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s" % path)


def gen_presets():
    """The agent hyper-parameters of the reference PRESETS this engine mirrors, read from the preset
    modules imported here under the stubs (rl_coach/presets/{CartPole_DQN, Atari_Dueling_DDQN,
    Mujoco_ClippedPPO, Atari_DQN_with_PER, Mujoco_TD3, Mujoco_SAC}.py) -> tests/golden/presets.json."""
    import importlib
    import json

    def scalars(o):
        out = {}
        for k, v in (vars(o).items() if hasattr(o, "__dict__") else ()):
            if k.startswith('_'):
                continue
            if isinstance(v, (int, float, bool, str, type(None))):
                out[k] = v
            elif hasattr(v, 'num_steps'):
                out[k] = [type(v).__name__, v.num_steps]
            elif isinstance(v, (tuple, list)) and all(isinstance(x, (int, float)) for x in v):
                out[k] = list(v)
            elif hasattr(v, 'current_value') and hasattr(v, 'initial_value'):
                out[k] = ["schedule", type(v).__name__, float(v.initial_value),
                          float(getattr(v, 'final_value', v.initial_value)),
                          int(getattr(v, 'decay_steps', 0) or 0)]
        return out
    out = {}
    for name in ("CartPole_DQN", "Atari_Dueling_DDQN", "Mujoco_ClippedPPO", "Atari_DQN_with_PER", "Mujoco_TD3",
                 "Mujoco_SAC"):
        gm = importlib.import_module("rl_coach.presets." + name).graph_manager
        ap = gm.agent_params
        nets = {}
        for n, w in ap.network_wrappers.items():
            nets[n] = scalars(w)
            heads = getattr(w, "heads_parameters", None) or []
            nets[n]["heads"] = [[type(h).__name__, float(getattr(h, "rescale_gradient_from_head_by_factor", 1.0))]
                                for h in heads]
            def widths(scheme):           # a scheme is an enum member or an explicit list of layers
                if isinstance(scheme, (list, tuple)):
                    return [getattr(l, "units", getattr(l, "num_filters", None)) for l in scheme]
                return str(scheme)
            nets[n]["middleware_layers"] = widths(getattr(getattr(w, "middleware_parameters", None), "scheme", None))
            emb = getattr(w, "input_embedders_parameters", {}) or {}
            nets[n]["embedder_layers"] = {k: widths(getattr(e, "scheme", None)) for k, e in emb.items()}
            nets[n]["embedder_activation"] = {k: getattr(e, "activation_function", None) for k, e in emb.items()}
            nets[n]["middleware_activation"] = getattr(getattr(w, "middleware_parameters", None),
                                                       "activation_function", None)
        out[name] = {"agent_class": type(ap).__name__, "algorithm": scalars(ap.algorithm), "networks": nets,
                     "exploration": dict(scalars(ap.exploration), **{"class": type(ap.exploration).__name__}),
                     "memory": dict(scalars(ap.memory), **{"class": type(ap.memory).__name__}),
                     "schedule": scalars(gm.schedule_params) if hasattr(gm, "schedule_params") else {}}
    path = os.path.join(HERE, "presets.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s" % path)


def gen_loop():
    """The REAL reference agent — `DQNAgent(DQNAgentParameters())` constructed by its own __init__
    (memory, exploration policy, filters, signals), with only `create_networks` overridden to return
    the oracle-backed stand-in — driven through the reference's step cycle (LevelManager.step
    level_manager.py:215-269: observe -> act -> env.step, terminal response observed at once;
    GraphManager.train_and_act: train() after every step) on the synthetic env, uniform and
    prioritized replay.  Stores every action, the number of transitions visible at every train() call,
    a key of every sampled transition and the final weights; tests/test_update_pins.py replays
    oracle.agents.DQNAgentOracle (reference_order=True) under the same seeds."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import DQNOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplayParameters
    from rl_coach.schedules import LinearSchedule
    from rl_coach.spaces import DiscreteActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, CAP, HEATUP, TRAIN, SEED = 4, 3, 7, 8, 32, 12, 120, 5
    out = {"hp": np.array([D, A, L, B, CAP, HEATUP, TRAIN, SEED])}
    rng = np.random.RandomState(0)
    arrays = {}
    for name, (i, o) in {"main/embedder/dense0": (D, 16), "main/middleware/dense0": (16, 12),
                         "main/q_head/dense": (12, A)}.items():
        arrays[name + "/kernel"] = [rng.uniform(-.5, .5, (i, o)).astype(np.float32)]
        arrays[name + "/bias"] = [np.zeros(o, np.float32)]
        out["init|%s/kernel" % name], out["init|%s/bias" % name] = arrays[name + "/kernel"][0], arrays[name + "/bias"][0]
    for variant in ("uniform", "per"):
        ap = DQNAgentParameters()
        ap.task_parameters = TaskParameters()
        ap.name = "agent"
        ap.visualization.dump_csv = False
        ap.is_a_highest_level_agent = False
        ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()
        if variant == "per":
            ap.memory = PrioritizedExperienceReplayParameters()           # alpha .6, constant beta .4
        ap.network_wrappers['main'].batch_size = B
        ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
        ap.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(10)
        ap.memory.max_size = (MemoryGranularity.Transitions, CAP)
        ap.exploration.epsilon_schedule = LinearSchedule(1.0, 0.1, 50)
        holder = {}

        class Agent(DQNAgent):
            def create_networks(self):
                net = DQNOracle(copy.deepcopy(arrays), (D,), A, lr=1e-3, huber=False)
                holder["net"] = net
                w = OB.DQNWrapper(net)
                w.has_target = True
                w.set_is_training = lambda s: None
                w.update_target_network = lambda rate=1.0: net.update_target(rate)
                for n in (w.online_network, w.target_network):
                    n.reset_internal_memory = lambda: None
                return {'main': w}
        agent = Agent(ap)
        agent.set_environment_parameters(SpacesDefinition(
            state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
            action=DiscreteActionSpace(A), reward=RewardSpace(1)))
        env = SynthVecEnv(1, 1, D, L, 99)
        random.seed(SEED)
        np.random.seed(SEED)
        agent.exploration_policy.current_random_value = np.random.rand()      # e_greedy.py:82, after seeding
        actions, visible, keys = [], [], []
        sample = agent.memory.sample

        def logged_sample(size):
            b = sample(size)
            keys.append([float(t.state['observation'][0]) for t in b])
            visible.append(agent.memory.num_transitions())
            return b
        agent.memory.sample = logged_sample
        resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
        agent.reset_internal_state()
        reset_required, first = False, None
        for step in range(HEATUP + TRAIN):
            agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
            if reset_required:
                agent.reset_internal_state()
                resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
                reset_required = False
            agent.observe(resp)
            actions.append(int(agent.act().action))
            nxt, rst, rew, done = env.step()
            resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
            if resp.game_over:
                agent.observe(resp)
                agent.handle_episode_ended()
                reset_required, first = True, rst[0]
            if step >= HEATUP:
                agent.train()
        out[variant + "|actions"], out[variant + "|visible"] = np.array(actions), np.array(visible)
        out[variant + "|keys"] = np.array(keys)
        _flat(variant + "|final", holder["net"].weights(), out)
    _save("loop", **out)


def gen_td3_loop():
    """The REAL reference `TD3Agent(TD3AgentParameters())` — own __init__ (EpisodicExperienceReplay, AdditiveNoise
    exploration, signals), only `create_networks` overridden to return the oracle-backed actor / twin-critic stand-ins —
    driven through the reference's step cycle (LevelManager.step: observe -> act -> env.step, terminal response observed
    at once, handle_episode_ended; train() after every step, which for TD3 runs `episode length` updates when an episode
    has ended) on the synthetic env: heat-up with random actions, then training.  Stores every action the agent
    RECORDED, the training iteration after every step, a key of every sampled transition, the stored game_over flags
    and the final weights; tests/test_update_pins.py replays oracle.agents.TD3AgentOracle under the same seeds."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle import ac_nets as O
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.td3_agent import TD3Agent, TD3AgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.spaces import BoxActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, HEATUP, TRAIN, SEED = 5, 2, 6, 8, 12, 30, 13
    out = {"hp": np.array([D, A, L, B, HEATUP, TRAIN, SEED])}
    rng = np.random.RandomState(3)
    a_arrays = _rand_arrays(rng, {"actor/embedder/dense0": (D, 20, 1), "actor/middleware/dense0": (20, 12, 1),
                                  "actor/ddpg_actor_head/fc_mean": (12, A, 1)})
    c_arrays = _rand_arrays(rng, {"critic/middleware/dense0": (D + A, 20, 2), "critic/middleware/dense1": (20, 12, 2),
                                  "critic/v_head/output": (12, 1, 2)})
    for k, v in list(a_arrays.items()) + list(c_arrays.items()):
        for t, arr in enumerate(v):
            out["init|%s|%d" % (k, t)] = arr
    ap = TD3AgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()
    for n in ap.network_wrappers.values():
        n.batch_size = B
    holder = {}

    class Agent(TD3Agent):
        def create_networks(self):
            actor = O.ActorOracle(copy.deepcopy(a_arrays), 1.0, lr=1e-3)
            critic = O.CriticOracle(copy.deepcopy(c_arrays), streams=2, lr=1e-3)
            holder["actor"], holder["critic"] = actor, critic
            wa, wc = OB.ActorWrapper(actor), OB.CriticWrapper(critic, A)
            for w, net in ((wa, actor), (wc, critic)):
                w.has_target = True
                w.set_is_training = lambda s: None
                w.update_target_network = (lambda rate=1.0, net=net: net.mix_target(rate))
                for nn in (w.online_network, w.target_network):
                    nn.reset_internal_memory = lambda: None
            return {'actor': wa, 'critic': wc}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
        action=BoxActionSpace(A, -1.0, 1.0), reward=RewardSpace(1)))
    agent.parent_level_manager = _Obj(environment=_Obj(env=_Obj(_max_episode_steps=L)))
    env = SynthVecEnv(1, 1, D, L, 55)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, iters, keys, stored_go = [], [], [], []
    sample = agent.memory.sample

    def logged_sample(size):
        b = sample(size)
        keys.append([float(t.state['observation'][0]) for t in b])
        return b
    agent.memory.sample = logged_sample
    resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
    agent.reset_internal_state()
    reset_required, first = False, None
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(np.array(agent.act().action, dtype=np.float64))
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            stored_go.append([bool(t.game_over) for t in agent.current_episode_buffer.transitions])
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    out["actions"], out["iters"] = np.array(actions), np.array(iters)
    out["keys"] = np.array(keys)
    out["stored_game_over"] = np.array(stored_go)
    _flat("final|actor", holder["actor"].weights(), out)
    _flat("final|critic", holder["critic"].weights(), out)
    _save("td3_loop", **out)


def gen_ddpg_loop():
    """The REAL reference `DDPGAgent(DDPGAgentParameters())` (own __init__: EpisodicExperienceReplay, OUProcess
    exploration restarted at every episode start, one update per env-step once a complete episode is stored, target
    networks mixed after every update) with oracle-backed actor / critic stand-ins, stepped like gen_td3_loop.
    tests/test_update_pins.py replays oracle.agents.DDPGAgentOracle under the same seeds."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle import ac_nets as O
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.ddpg_agent import DDPGAgent, DDPGAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.spaces import BoxActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, HEATUP, TRAIN, SEED = 5, 2, 6, 8, 9, 27, 17
    out = {"hp": np.array([D, A, L, B, HEATUP, TRAIN, SEED])}
    rng = np.random.RandomState(4)
    a_arrays = _rand_arrays(rng, {"actor/embedder/dense0": (D, 20, 1), "actor/middleware/dense0": (20, 12, 1),
                                  "actor/ddpg_actor_head/fc_mean": (12, A, 1)})
    c_arrays = _rand_arrays(rng, {"critic/embedder/dense0": (D, 18, 1), "critic/middleware/dense0": (18 + A, 12, 1),
                                  "critic/v_head/output": (12, 1, 1)})
    for k, v in list(a_arrays.items()) + list(c_arrays.items()):
        for t, arr in enumerate(v):
            out["init|%s|%d" % (k, t)] = arr
    ap = DDPGAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()
    for n in ap.network_wrappers.values():
        n.batch_size = B
    holder = {}

    class Agent(DDPGAgent):
        def create_networks(self):
            actor = O.ActorOracle(copy.deepcopy(a_arrays), 1.0, lr=1e-4)
            critic = O.CriticOracle(copy.deepcopy(c_arrays), streams=1, lr=1e-3)
            holder["actor"], holder["critic"] = actor, critic
            wa, wc = OB.ActorWrapper(actor), OB.CriticWrapper(critic, A)
            for w, net in ((wa, actor), (wc, critic)):
                w.has_target = True
                w.set_is_training = lambda s: None
                w.update_target_network = (lambda rate=1.0, net=net: net.mix_target(rate))
                for nn in (w.online_network, w.target_network):
                    nn.reset_internal_memory = lambda: None
            return {'actor': wa, 'critic': wc}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
        action=BoxActionSpace(A, -1.0, 1.0), reward=RewardSpace(1)))
    env = SynthVecEnv(1, 1, D, L, 56)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, iters, keys = [], [], []
    sample = agent.memory.sample

    def logged_sample(size):
        b = sample(size)
        keys.append([float(t.state['observation'][0]) for t in b])
        return b
    agent.memory.sample = logged_sample
    resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
    agent.reset_internal_state()
    reset_required, first = False, None
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        agent.exploration_policy.change_phase(agent.phase)
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(np.array(agent.act().action, dtype=np.float64))
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    out["actions"], out["iters"] = np.array(actions), np.array(iters)
    out["keys"] = np.array(keys)
    _flat("final|actor", holder["actor"].weights(), out)
    _flat("final|critic", holder["critic"].weights(), out)
    _save("ddpg_loop", **out)


def gen_sac_loop():
    """The REAL reference `SoftActorCriticAgent(SoftActorCriticAgentParameters())` (own __init__: non-episodic
    ExperienceReplay, heat-up with random actions, then the squashed policy SAMPLE as the action, one update per
    env-step as soon as the memory holds a batch, V target mixed after every update) with oracle-backed policy / twin-Q
    / V stand-ins, stepped like gen_td3_loop.  TF's sampling op is the stand-in's np.random.standard_normal draw (one
    per sess.run of the policy graph: 1 at acting, 3 in learn_from_batch).  tests/test_update_pins.py replays
    oracle.agents.SACAgentOracle under the same seeds."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle import ac_nets as O
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.soft_actor_critic_agent import SoftActorCriticAgent, SoftActorCriticAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.spaces import BoxActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, HEATUP, TRAIN, SEED = 6, 3, 7, 8, 6, 30, 23
    out = {"hp": np.array([D, A, L, B, HEATUP, TRAIN, SEED])}
    rng = np.random.RandomState(5)
    p_arr = _rand_arrays(rng, {"policy/embedder/dense0": (D, 20, 1), "policy/middleware/dense0": (20, 12, 1),
                               "policy/sac_policy_head/policy_mu_logsig": (12, 2 * A, 1)})
    q_arr = _rand_arrays(rng, {"q/q_head/obs_fc": (D, 14, 2), "q/q_head/act_fc": (A, 14, 2),
                               "q/q_head/fc1": (14, 14, 2), "q/q_head/q_output": (14, 1, 2)})
    v_arr = _rand_arrays(rng, {"v/embedder/dense0": (D, 20, 1), "v/middleware/dense0": (20, 12, 1),
                               "v/v_values_head/output": (12, 1, 1)})
    for arrs in (p_arr, q_arr, v_arr):
        for k, v in arrs.items():
            for t, arr in enumerate(v):
                out["init|%s|%d" % (k, t)] = arr
    ap = SoftActorCriticAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()
    for n in ap.network_wrappers.values():
        n.batch_size = B
    holder = {}

    class Agent(SoftActorCriticAgent):
        def create_networks(self):
            pol, qn, vn = O.SACPolicyOracle(copy.deepcopy(p_arr)), O.SACQOracle(copy.deepcopy(q_arr)), \
                O.SACValueOracle(copy.deepcopy(v_arr))
            holder["policy"], holder["q"], holder["v"] = pol, qn, vn
            ws = {'policy': OB.SACPolicyWrapper(pol, A), 'q': OB.SACQWrapper(qn), 'v': OB.SACValueWrapper(vn)}
            for name, w in ws.items():
                w.has_target = name == 'v'
                w.set_is_training = lambda s: None
                w.update_target_network = lambda rate=1.0: None          # no target network: nothing to mix
                nets = [w.online_network] + ([w.target_network] if name == 'v' else [])
                for nn in nets:
                    nn.reset_internal_memory = lambda: None
            ws['v'].update_target_network = lambda rate=1.0: vn.mix_target(rate)
            return ws
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
        action=BoxActionSpace(A, -1.0, 1.0), reward=RewardSpace(1)))
    env = SynthVecEnv(1, 1, D, L, 57)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, iters, keys, visible = [], [], [], []
    sample = agent.memory.sample

    def logged_sample(size):
        b = sample(size)
        keys.append([float(t.state['observation'][0]) for t in b])
        visible.append(agent.memory.num_transitions())
        return b
    agent.memory.sample = logged_sample
    resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
    agent.reset_internal_state()
    reset_required, first = False, None
    for step in range(HEATUP + TRAIN):
        agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(np.array(agent.act().action, dtype=np.float64))
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        if step >= HEATUP:
            agent.train()
        iters.append(agent.training_iteration)
    out["actions"], out["iters"] = np.array(actions), np.array(iters)
    out["keys"], out["visible"] = np.array(keys), np.array(visible)
    for nm in ("policy", "q", "v"):
        _flat("final|" + nm, holder[nm].weights(), out)
    _save("sac_loop", **out)


def gen_ppo_loop():
    """The REAL reference `ClippedPPOAgent(ClippedPPOAgentParameters())` (own __init__: episodic memory,
    Categorical exploration, episode buffers, _should_train with act_for_full_episodes, train ->
    fill_advantages -> train_network), network = oracle stand-in, stepped through the reference cycle for
    three rollouts; tests/test_update_pins.py replays oracle.agents.ClippedPPOAgentOracle (act + train)."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import ClippedPPOOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.spaces import DiscreteActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, PLAY, EPOCHS, STEPS, SEED = 5, 4, 8, 8, 16, 2, 48, 9
    out = {"hp": np.array([D, A, L, B, PLAY, EPOCHS, STEPS, SEED])}
    rng = np.random.RandomState(1)
    arrays = _rand_arrays(rng, {"main/embedder/dense0": (D, 20, 2), "main/middleware/dense0": (20, 12, 2),
                                "main/v_head/dense": (12, 1, 1), "main/ppo_head/policy_fc": (12, A, 1)})
    for k, v in arrays.items():
        for t, arr in enumerate(v):
            out["init|%s|%d" % (k, t)] = arr
    ap = ClippedPPOAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()
    ap.network_wrappers['main'].batch_size = B
    ap.network_wrappers['main'].learning_rate = 1e-3
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAY)
    ap.algorithm.optimization_epochs = EPOCHS
    holder = {}

    class Agent(ClippedPPOAgent):
        def create_networks(self):
            net = ClippedPPOOracle(copy.deepcopy(arrays), (D,), A, lr=1e-3,
                                   clip_eps=self.ap.algorithm.clip_likelihood_ratio_using_epsilon,
                                   beta_entropy=self.ap.algorithm.beta_entropy)
            holder["net"] = net
            w = OB.PPOWrapper(net)
            w.has_target = True
            return {'main': w}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
        action=DiscreteActionSpace(A), reward=RewardSpace(1)))
    agent.update_log = lambda: None
    env = SynthVecEnv(1, 1, D, L, 77)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, trained_at = [], []
    resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
    agent.reset_internal_state()
    agent.phase = RunPhase.TRAIN
    reset_required, first = False, None
    for step in range(STEPS):
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(int(agent.act().action))
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        before = agent.training_iteration
        agent.train()
        if agent.training_iteration != before:
            trained_at.append(step)
    out["actions"], out["trained_at"] = np.array(actions), np.array(trained_at)
    _flat("final", holder["net"].weights(), out)
    _save("ppo_loop", **out)


def gen_ppoc_loop():
    """The REAL reference `ClippedPPOAgent` on a BoxActionSpace with the Mujoco_ClippedPPO pre-network filter
    (ObservationNormalizationFilter, numpy running statistics): acting normalises with the statistics as they are
    (update_pre_network_filters_state_on_inference = False) and samples np.random.normal(mean, std) through
    AdditiveNoise; train() pushes the whole dataset into the statistics first, then normalises it with the updated
    statistics (clipped_ppo_agent.py:318-322) before fill_advantages / train_network.  Three rollouts on the synthetic env;
    tests/test_update_pins.py replays oracle.agents.ClippedPPOAgentOracle(continuous=True, normalize=True)."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import ClippedPPOOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import InputFilter, NoInputFilter, NoOutputFilter
    from rl_coach.filters.observation.observation_normalization_filter import ObservationNormalizationFilter
    from rl_coach.spaces import BoxActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    D, A, L, B, PLAY, EPOCHS, STEPS, SEED = 5, 2, 8, 8, 16, 2, 48, 21
    out = {"hp": np.array([D, A, L, B, PLAY, EPOCHS, STEPS, SEED])}
    rng = np.random.RandomState(2)
    arrays = _rand_arrays(rng, {"main/embedder/dense0": (D, 20, 2), "main/middleware/dense0": (20, 12, 2),
                                "main/v_head/dense": (12, 1, 1), "main/ppo_head/policy_mean": (12, A, 1)})
    arrays["main/ppo_head/policy_log_std"] = [rng.uniform(-0.5, 0.1, (A,)).astype(np.float32)]
    for k, v in arrays.items():
        for t, arr in enumerate(v):
            out["init|%s|%d" % (k, t)] = arr
    ap = ClippedPPOAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter = NoInputFilter(), NoOutputFilter()
    ap.pre_network_filter = InputFilter(is_a_reference_filter=False)
    ap.pre_network_filter.add_observation_filter('observation', 'normalize_observation',
                                                 ObservationNormalizationFilter(name='normalize_observation'))
    ap.network_wrappers['main'].batch_size = B
    ap.network_wrappers['main'].learning_rate = 1e-3
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAY)
    ap.algorithm.optimization_epochs = EPOCHS
    holder = {}

    class Agent(ClippedPPOAgent):
        def create_networks(self):
            net = ClippedPPOOracle(copy.deepcopy(arrays), (D,), A, lr=1e-3,
                                   clip_eps=self.ap.algorithm.clip_likelihood_ratio_using_epsilon,
                                   beta_entropy=self.ap.algorithm.beta_entropy, continuous=True)
            holder["net"] = net
            w = OB.PPOContinuousWrapper(net)
            w.has_target = True
            return {'main': w}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
        action=BoxActionSpace(A, -1.0, 1.0), reward=RewardSpace(1)))
    agent.update_log = lambda: None
    env = SynthVecEnv(1, 1, D, L, 78)
    random.seed(SEED)
    np.random.seed(SEED)
    actions, trained_at, stats = [], [], []
    resp = EnvResponse(next_state={'observation': env.reset()[0].copy()}, reward=0, game_over=False)
    agent.reset_internal_state()
    agent.phase = RunPhase.TRAIN
    agent.exploration_policy.change_phase(RunPhase.TRAIN)
    reset_required, first = False, None
    for step in range(STEPS):
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': first.copy()}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(np.array(agent.act().action, dtype=np.float64))
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': nxt[0].copy()}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        before = agent.training_iteration
        agent.train()
        if agent.training_iteration != before:
            trained_at.append(step)
            rs = agent.pre_network_filter._observation_filters['observation']['normalize_observation'] \
                .running_observation_stats
            stats.append(np.concatenate([[rs.n], np.asarray(rs.mean, dtype=np.float64).ravel(),
                                         np.asarray(rs.std, dtype=np.float64).ravel()]))
    out["actions"], out["trained_at"], out["stats"] = np.array(actions), np.array(trained_at), np.array(stats)
    _flat("final", holder["net"].weights(), out)
    _save("ppoc_loop", **out)


def gen_ppo_image_loop():
    """The headline path at loop level: the REAL reference `ClippedPPOAgent` (DiscreteActionSpace) behind the Atari
    input filter's stateful part — ObservationStackingFilter(4) on uint8 frames and RewardClippingFilter(-1, 1), as
    Atari_ClippedPPO-style presets configure them (the rescale / grayscale stages are stateless and pinned on their own) —
    stepped through three rollouts on the synthetic image env: the stack restarts at every episode start (first frame
    four times), the transitions hold LazyStack states, rewards are clipped on the way into the memory.
    tests/test_update_pins.py replays oracle.agents.ClippedPPOAgentOracle in image mode."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import ClippedPPOOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import InputFilter, NoInputFilter, NoOutputFilter
    from rl_coach.filters.observation.observation_stacking_filter import ObservationStackingFilter
    from rl_coach.filters.reward.reward_clipping_filter import RewardClippingFilter
    from rl_coach.spaces import DiscreteActionSpace, ObservationSpace, RewardSpace, SpacesDefinition, StateSpace
    H, A, L, B, PLAY, EPOCHS, STEPS, SEED, STACK = 36, 3, 6, 6, 12, 2, 36, 33, 4
    out = {"hp": np.array([H, A, L, B, PLAY, EPOCHS, STEPS, SEED, STACK])}
    rng = np.random.RandomState(6)
    arrays = _rand_arrays(rng, {"main/embedder/conv0": (8 * 8 * STACK, 32, 2), "main/embedder/conv1": (4 * 4 * 32, 64, 2),
                                "main/embedder/conv2": (3 * 3 * 64, 64, 2), "main/middleware/dense0": (64, 16, 2),
                                "main/v_head/dense": (16, 1, 1), "main/ppo_head/policy_fc": (16, A, 1)})
    for k, v in arrays.items():
        for t, arr in enumerate(v):
            out["init|%s|%d" % (k, t)] = arr
    ap = ClippedPPOAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter = InputFilter(is_a_reference_filter=False)
    ap.input_filter.add_observation_filter('observation', 'stacking', ObservationStackingFilter(STACK))
    ap.input_filter.add_reward_filter('clipping', RewardClippingFilter(-1.0, 1.0))
    ap.output_filter, ap.pre_network_filter = NoOutputFilter(), NoInputFilter()
    ap.network_wrappers['main'].batch_size = B
    ap.network_wrappers['main'].learning_rate = 1e-3
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(PLAY)
    ap.algorithm.optimization_epochs = EPOCHS
    holder = {}

    class Agent(ClippedPPOAgent):
        def create_networks(self):
            net = ClippedPPOOracle(copy.deepcopy(arrays), (H, H, STACK), A, lr=1e-3,
                                   clip_eps=self.ap.algorithm.clip_likelihood_ratio_using_epsilon,
                                   beta_entropy=self.ap.algorithm.beta_entropy)
            holder["net"] = net
            w = OB.PPOWrapper(net)
            w.has_target = True
            return {'main': w}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': ObservationSpace(np.array([H, H]), low=0, high=255)}), goal=None,
        action=DiscreteActionSpace(A), reward=RewardSpace(1)))
    agent.update_log = lambda: None
    env = SynthVecEnv(0, 1, H * H, L, 79)
    frame = lambda flat: np.asarray(flat, dtype=np.uint8).reshape(H, H).copy()
    random.seed(SEED)
    np.random.seed(SEED)
    actions, trained_at, rewards = [], [], []
    resp = EnvResponse(next_state={'observation': frame(env.reset()[0])}, reward=0, game_over=False)
    agent.reset_internal_state()
    agent.phase = RunPhase.TRAIN
    reset_required, first = False, None
    for step in range(STEPS):
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': frame(first)}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        actions.append(int(agent.act().action))
        nxt, rst, rew, done = env.step()
        rewards.append(float(rew[0]))
        resp = EnvResponse(next_state={'observation': frame(nxt[0])}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        before = agent.training_iteration
        agent.train()
        if agent.training_iteration != before:
            trained_at.append(step)
    out["actions"], out["trained_at"], out["env_rewards"] = np.array(actions), np.array(trained_at), np.array(rewards)
    _flat("final", holder["net"].weights(), out)
    _save("ppo_image_loop", **out)


def gen_dqn_image_loop():
    """C3 at loop level: the REAL reference `DQNAgent` with PrioritizedExperienceReplay behind ObservationStackingFilter(4)
    + RewardClippingFilter(-1, 1) on uint8 frames (the stateful part of the Atari input filter), heat-up then training with
    a wrapping buffer, on the synthetic image env.  Stores every action, the transitions visible at every train(), a key
    of every sampled transition (sum of its stacked state) and the final weights; tests/test_update_pins.py replays
    oracle.agents.DQNAgentOracle in image mode (reference_order = True)."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import DQNOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import InputFilter, NoInputFilter, NoOutputFilter
    from rl_coach.filters.observation.observation_stacking_filter import ObservationStackingFilter
    from rl_coach.filters.reward.reward_clipping_filter import RewardClippingFilter
    from rl_coach.memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplayParameters
    from rl_coach.schedules import LinearSchedule
    from rl_coach.spaces import DiscreteActionSpace, ObservationSpace, RewardSpace, SpacesDefinition, StateSpace
    H, A, L, B, CAP, HEATUP, TRAIN, SEED, STACK = 36, 3, 7, 6, 32, 10, 60, 41, 4
    out = {"hp": np.array([H, A, L, B, CAP, HEATUP, TRAIN, SEED, STACK])}
    rng = np.random.RandomState(8)
    arrays = _rand_arrays(rng, {"main/embedder/conv0": (8 * 8 * STACK, 32, 1), "main/embedder/conv1": (4 * 4 * 32, 64, 1),
                                "main/embedder/conv2": (3 * 3 * 64, 64, 1), "main/middleware/dense0": (64, 16, 1),
                                "main/q_head/dense": (16, A, 1)})
    for k, v in arrays.items():
        out["init|%s|0" % k] = v[0]
    frame = lambda flat: np.asarray(flat, dtype=np.uint8).reshape(H, H).copy()
    for variant in ("uniform", "per"):
        ap = DQNAgentParameters()
        ap.task_parameters = TaskParameters()
        ap.name = "agent"
        ap.visualization.dump_csv = False
        ap.is_a_highest_level_agent = False
        ap.input_filter = InputFilter(is_a_reference_filter=False)
        ap.input_filter.add_observation_filter('observation', 'stacking', ObservationStackingFilter(STACK))
        ap.input_filter.add_reward_filter('clipping', RewardClippingFilter(-1.0, 1.0))
        ap.output_filter, ap.pre_network_filter = NoOutputFilter(), NoInputFilter()
        if variant == "per":
            ap.memory = PrioritizedExperienceReplayParameters()           # alpha .6, constant beta .4
        ap.network_wrappers['main'].batch_size = B
        ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(1)
        ap.algorithm.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(10)
        ap.memory.max_size = (MemoryGranularity.Transitions, CAP)
        ap.exploration.epsilon_schedule = LinearSchedule(1.0, 0.1, 40)
        holder = {}

        class Agent(DQNAgent):
            def create_networks(self):
                net = DQNOracle(copy.deepcopy(arrays), (H, H, STACK), A, lr=1e-3, huber=True)
                holder["net"] = net
                w = OB.DQNWrapper(net)
                w.has_target = True
                w.set_is_training = lambda s: None
                w.update_target_network = lambda rate=1.0: net.update_target(rate)
                for n in (w.online_network, w.target_network):
                    n.reset_internal_memory = lambda: None
                return {'main': w}
        agent = Agent(ap)
        agent.set_environment_parameters(SpacesDefinition(
            state=StateSpace({'observation': ObservationSpace(np.array([H, H]), low=0, high=255)}), goal=None,
            action=DiscreteActionSpace(A), reward=RewardSpace(1)))
        env = SynthVecEnv(0, 1, H * H, L, 80)
        random.seed(SEED)
        np.random.seed(SEED)
        agent.exploration_policy.current_random_value = np.random.rand()      # e_greedy.py:82, after seeding
        actions, visible, keys = [], [], []
        sample = agent.memory.sample

        def logged_sample(size):
            b = sample(size)
            keys.append([float(np.asarray(t.state['observation'], dtype=np.float64).sum()) for t in b])
            visible.append(agent.memory.num_transitions())
            return b
        agent.memory.sample = logged_sample
        resp = EnvResponse(next_state={'observation': frame(env.reset()[0])}, reward=0, game_over=False)
        agent.reset_internal_state()
        reset_required, first = False, None
        for step in range(HEATUP + TRAIN):
            agent.phase = RunPhase.HEATUP if step < HEATUP else RunPhase.TRAIN
            if reset_required:
                agent.reset_internal_state()
                resp = EnvResponse(next_state={'observation': frame(first)}, reward=0, game_over=False)
                reset_required = False
            agent.observe(resp)
            actions.append(int(agent.act().action))
            nxt, rst, rew, done = env.step()
            resp = EnvResponse(next_state={'observation': frame(nxt[0])}, reward=float(rew[0]), game_over=bool(done[0]))
            if resp.game_over:
                agent.observe(resp)
                agent.handle_episode_ended()
                reset_required, first = True, rst[0]
            if step >= HEATUP:
                agent.train()
        out[variant + "|actions"], out[variant + "|visible"] = np.array(actions), np.array(visible)
        out[variant + "|keys"] = np.array(keys)
        _flat(variant + "|final", holder["net"].weights(), out)
    _save("dqn_image_loop", **out)


def gen_csv_columns():
    """What Agent.update_log (agent.py:509-556) writes: the CSV column list of the REAL reference DQNAgent (own
    __init__, registered signals) after one update_log call, the values of the statistics columns for known
    samples, and `Signal` (utils.py:162-212) statistics of random sample streams (scalars and arrays mixed)."""
    import copy
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _oracle_backend as OB
    from oracle.agents import DQNOracle
    from rl_coach.agents.dqn_agent import DQNAgent, DQNAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import RunPhase
    from rl_coach.filters.filter import NoInputFilter, NoOutputFilter
    from rl_coach.spaces import DiscreteActionSpace, RewardSpace, SpacesDefinition, StateSpace, VectorObservationSpace
    from rl_coach.utils import Signal
    D, A = 4, 3
    rng = np.random.RandomState(0)
    arrays = {}
    for name, (i, o) in {"main/embedder/dense0": (D, 16), "main/middleware/dense0": (16, 12), "main/q_head/dense": (12, A)}.items():
        arrays[name + "/kernel"] = [rng.uniform(-.5, .5, (i, o)).astype(np.float32)]
        arrays[name + "/bias"] = [np.zeros(o, np.float32)]
    ap = DQNAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    ap.input_filter, ap.output_filter, ap.pre_network_filter = NoInputFilter(), NoOutputFilter(), NoInputFilter()

    class Agent(DQNAgent):
        def create_networks(self):
            net = DQNOracle(copy.deepcopy(arrays), (D,), A, lr=1e-3, huber=False)
            w = OB.DQNWrapper(net)
            w.has_target = True
            for n in (w.online_network, w.target_network):
                n.reset_internal_memory = lambda: None
            return {'main': w}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(state=StateSpace({'observation': VectorObservationSpace(D)}), goal=None,
                                                      action=DiscreteActionSpace(A), reward=RewardSpace(1)))
    agent.get_current_time = lambda: 1
    agent.agent_logger.dump_output_csv = lambda *a, **k: None
    agent.reset_internal_state()
    agent.phase = RunPhase.TRAIN
    for v in (0.5, 0.25, 2.0):
        agent.loss.add_sample(v)
    agent.q_values.add_sample(np.array([1.0, -1.0, 3.0]))
    agent.update_log()
    data = agent.agent_logger.data
    out = {"dqn_index": agent.agent_logger.index_name if hasattr(agent.agent_logger, "index_name") else "Episode #",
           "dqn_columns": list(data.columns),
           "dqn_row": {k: (v if isinstance(v, str) else float(v)) for k, v in data.iloc[0].to_dict().items()},
           "signal_cases": []}
    for case in range(4):
        s = Signal("x")
        samples = []
        for _ in range(rng.randint(1, 9)):
            if rng.rand() < 0.5:
                v = rng.randn(rng.randint(1, 40)).astype(np.float32 if case % 2 else np.float64)
                s.add_sample(v) if case >= 2 else [s.add_sample(float(x)) for x in v]
            else:
                v = np.array([rng.randn()])
                s.add_sample(v) if case >= 2 else s.add_sample(float(v[0]))
            samples.append([float(x) for x in v])
        out["signal_cases"].append({"samples": samples, "f32": bool(case % 2), "mean": float(s.get_mean()),
                                    "stdev": float(s.get_stdev()), "max": float(s.get_max()), "min": float(s.get_min())})
    with open(os.path.join(HERE, "csv_columns.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote csv_columns.json")


GROUPS = {"csv_columns": gen_csv_columns, "per": gen_per, "er": gen_er, "episodic": gen_episodic, "stack": gen_stack, "filters": gen_filters, "gae": gen_gae,
          "targets": gen_targets, "explore": gen_explore, "updates": gen_updates, "ppo_update": gen_ppo_update, "cadence": gen_cadence, "defaults": gen_defaults, "presets": gen_presets, "loop": gen_loop, "ppo_loop": gen_ppo_loop, "td3_loop": gen_td3_loop, "ddpg_loop": gen_ddpg_loop, "sac_loop": gen_sac_loop, "ppoc_loop": gen_ppoc_loop, "ppo_image_loop": gen_ppo_image_loop, "dqn_image_loop": gen_dqn_image_loop}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GROUPS)
    for n in names:
        GROUPS[n]()
