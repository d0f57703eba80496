"""Whole-update pins: the oracle's update functions (`ddpg_update`, `td3_update`, `sac_update`,
`DQNOracle.learn_from_batch`) against fixtures produced by the REFERENCE's own
`learn_from_batch` code (agents/{ddpg,td3,soft_actor_critic,dqn}_agent.py) executing in the build
container on oracle-backed network stand-ins (tests/golden/make_golden.py `updates`,
tests/golden/_oracle_backend.py).  Both sides share the oracle's layer arithmetic, so what is pinned
is everything the agent code decides: which network sees which input, the order of the passes, the TD /
value targets, the sign and weighting of the policy gradients, when the actor is updated.
Tolerance: 1e-6 absolute on weights after three updates (fp32; the reference mixes in float64 scalars)."""
import os

import numpy as np
import pytest

from oracle import ac_nets as O
from oracle.agents import DQNOracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "updates.npz"))


def _arrays(fx, prefix):
    out = {}
    for k in fx.files:
        if k.startswith(prefix + "|init|"):
            _, _, name, t = k.split("|")
            out.setdefault(name, {})[int(t)] = fx[k]
    return {name: [towers[t] for t in sorted(towers)] for name, towers in out.items()}


def _select(arrays, root):
    return {k: v for k, v in arrays.items() if k.startswith(root)}


def _batch(fx, prefix, it):
    return tuple(fx["%s|batch%d|%s" % (prefix, it, k)] for k in ("s", "a", "r", "done", "ns"))


def _check_final(fx, prefix, net, atol=1e-6):
    n = 0
    for name, towers in net.weights().items():
        for t, arr in towers.items():
            np.testing.assert_allclose(arr, fx["%s|%s|%d" % (prefix, name, t)], rtol=0, atol=atol,
                                       err_msg="%s %s[%d]" % (prefix, name, t))
            n += 1
    assert n > 0


@pytest.mark.parametrize("name,streams", [("ddpg", 1), ("td3", 2)])
def test_ddpg_td3_update_equals_reference_learn_from_batch(fx, name, streams):
    arrays = _arrays(fx, name)
    actor = O.ActorOracle(_select(arrays, "actor/"), 1.0, lr=1e-3)
    critic = O.CriticOracle(_select(arrays, "critic/"), streams=streams, lr=1e-3)
    A = fx["%s|batch1|a" % name].shape[1]
    for it in range(1, 4):
        batch = _batch(fx, name, it)
        if name == "ddpg":
            r = O.ddpg_update(actor, critic, batch, 0.99)
        else:
            r = O.td3_update(actor, critic, batch, fx["td3|noise%d" % it], it, np.full(A, -1.0, np.float32),
                             np.full(A, 1.0, np.float32), 0.99, noise_clipping=0.5, policy_every=2)
        np.testing.assert_allclose(r["loss"], float(fx["%s|loss%d" % (name, it)]), rtol=1e-5)
        if name == "ddpg" or it % 2 == 0:
            actor.mix_target(0.01); critic.mix_target(0.01)
    _check_final(fx, name + "|final|actor", actor)
    _check_final(fx, name + "|final|critic", critic)


def test_sac_update_equals_reference_learn_from_batch(fx):
    arrays = _arrays(fx, "sac")
    pol = O.SACPolicyOracle(_select(arrays, "policy/"))
    qn = O.SACQOracle(_select(arrays, "q/"))
    vn = O.SACValueOracle(_select(arrays, "v/"))
    for it in range(1, 4):
        r = O.sac_update(pol, qn, vn, _batch(fx, "sac", it), fx["sac|normals%d" % it], 0.99, resample=True)
        np.testing.assert_allclose(r["loss"], float(fx["sac|loss%d" % it]), rtol=1e-5)
        vn.mix_target(0.005)
    _check_final(fx, "sac|final|policy", pol)
    _check_final(fx, "sac|final|q", qn)
    _check_final(fx, "sac|final|v", vn)


def test_dqn_update_equals_reference_learn_from_batch(fx):
    arrays = _arrays(fx, "dqn")
    D, A = fx["dqn|batch1|s"].shape[1], int(arrays["main/q_head/dense/kernel"][0].shape[1])
    o = DQNOracle(arrays, (D,), A, lr=1e-3)
    for it in range(1, 4):
        s, a, r, done, ns = _batch(fx, "dqn", it)
        res = o.learn_from_batch(s, ns, a, r, done, 0.99, None, False)
        np.testing.assert_allclose(res["loss"], float(fx["dqn|loss%d" % it]), rtol=1e-5)
        if it == 2:
            o.update_target(1.0)
    _check_final(fx, "dqn|final|main", o)


@pytest.mark.parametrize("prefix", ["ppo", "ppoc"])
def test_clipped_ppo_train_equals_reference_train(prefix):
    """ClippedPPOAgentOracle.train vs the reference's ClippedPPOAgent.train (clipped_ppo_agent.py:314-344)
    run on the oracle-backed stand-in: same transitions, weights and `random` seed -> same dataset
    shuffle, same per-epoch Batch.shuffle, same standardised advantages, same weights after
    epochs x minibatches updates.  "ppo": DiscreteActionSpace; "ppoc": BoxActionSpace (old policy = mean and std of
    the frozen copy, vector actions, the clip rescaler on input output_1_3)."""
    import random
    from oracle.agents import ClippedPPOAgentOracle
    fx = np.load(os.path.join(HERE, "golden", "ppo_update.npz"))
    D, A, B, n_env, L, epochs, seed = (int(x) for x in fx[prefix + "|hp"])
    arrays = _arrays(fx, prefix)
    continuous = prefix == "ppoc"

    class Env(object):
        kind, n_env = 1, 0
    Env.n_env = n_env
    o = ClippedPPOAgentOracle(arrays, Env(), A, discount=0.99, gae_lambda=0.95, batch_size=B,
                              playing_steps=n_env * L, epochs=epochs, clip_eps=0.2, beta_entropy=0.01, lr=1e-3,
                              reward_clip=None, continuous=continuous)
    o._ensure_net((D,))
    s, a, r, go = fx[prefix + "|states"], fx[prefix + "|actions"], fx[prefix + "|rewards"], fx[prefix + "|go"]
    for e in range(n_env):
        o.transitions[e] = [(s[i], a[i] if continuous else int(a[i]), float(r[i]), bool(go[i]))
                            for i in range(e * L, (e + 1) * L)]
    random.seed(seed)
    res = o.train()
    assert len(res) == epochs
    # standardised advantages, matched through the first state component (the reference shuffled its
    # dataset in place)
    by_key = {float(k): v for k, v in zip(fx[prefix + "|adv_state0"], fx[prefix + "|adv"])}
    ref_adv = np.array([by_key[float(x)] for x in s[:, 0]])
    np.testing.assert_allclose(o.dbg["adv"], ref_adv, rtol=1e-6, atol=1e-7)
    _check_final(fx, prefix + "|final", o.net, atol=2e-6)


def _cadence_cases():
    fx = np.load(os.path.join(HERE, "golden", "cadence.npz"))
    names = sorted({k.split("|")[0] for k in fx.files if not k.startswith(("full_", "episodic_"))})
    return fx, names


def test_product_training_cadence_equals_reference_agent_train():
    """The HOST scheduling of the device agents (coach_amd/agents/vector_agent.py: _training_phases_due,
    num_consecutive_training_steps, _should_update_online_weights_to_target) with one env, against the
    reference's Agent.train (agent.py:640-784) executed on the same 60-step sequence: identical
    number of updates and target copies at every env-step.  Runs on CPU: memory, networks and
    learn_from_batch are counters."""
    from coach_amd.agents.vector_agent import VectorOffPolicyAgent
    from coach_amd.core_types import EnvironmentSteps, RunPhase, TrainingSteps
    fx, names = _cadence_cases()
    assert len(names) == 4
    for name in names:
        playing, tkind, tn, consecutive = (int(x) for x in fx[name + "|cfg"])

        class Alg(object):
            act_for_full_episodes = False
            num_consecutive_playing_steps = EnvironmentSteps(playing)
            num_consecutive_training_steps = consecutive
            num_steps_between_copying_online_weights_to_target = TrainingSteps(tn) if tkind else EnvironmentSteps(tn)
            rate_for_copying_weights_to_target = 1.0

        class Ap(object):
            algorithm = Alg()

        class Memory(object):
            stored = 0
            def num_transitions(self): return self.stored
            def draw(self, B): return [0] * B
            def collate(self, d, B): return object()

        class Net(object):
            target, copies = object(), 0
            def update_target(self, rate): self.copies += 1

        class Agent(VectorOffPolicyAgent):
            def __init__(self):                       # no device state: scheduling logic only
                self.ap, self.memory, self.networks = Ap(), Memory(), {"main": Net()}
                self.n_env, self.batch_size, self.phase = 1, 4, RunPhase.TRAIN
                self.total_steps_counter = self.last_training_phase_step = self.training_iteration = 0
                self.last_target_network_update_step = 0
                self.debug_draws = self.debug_losses = None
                self.learned = 0

            def learn_from_batch(self, batch):
                self.learned += 1
                return 0.0
        ag = Agent()
        learned, copied = [], []
        for t in range(60):
            ag.total_steps_counter += 1
            ag.memory.stored += 1
            a, b = ag.learned, ag.networks["main"].copies
            ag.train()
            learned.append(ag.learned - a)
            copied.append(ag.networks["main"].copies - b)
        np.testing.assert_array_equal(learned, fx[name + "|learned"], err_msg=name)
        np.testing.assert_array_equal(copied, fx[name + "|copied"], err_msg=name)


def test_oracle_loop_cadence_equals_reference_agent_train():
    """oracle.agents.DQNAgentOracle.train (n_env = 1) on the same sequences (env-step target cadence)."""
    from oracle.agents import DQNAgentOracle
    fx, names = _cadence_cases()
    checked = 0
    for name in names:
        playing, tkind, tn, consecutive = (int(x) for x in fx[name + "|cfg"])
        if tkind or consecutive != 1:
            continue                                   # the oracle loop models EnvironmentSteps cadence
        o = DQNAgentOracle.__new__(DQNAgentOracle)
        o.n_env, o.per, o.B, o.discount, o.double = 1, None, 4, 0.99, False
        o.playing_steps, o.target_every = playing, tn
        o.total_steps = o.last_train = o.last_target = o.training_iteration = 0
        o.count, o.losses = 0, []
        counts = {"learn": 0, "copy": 0}

        class Net(object):
            def learn_from_batch(self, *a):
                counts["learn"] += 1
                return dict(loss=0.0, td_errors=np.zeros(4))

            def update_target(self, rate):
                counts["copy"] += 1
        o.net = Net()
        o._draw = lambda B: [0] * B
        o._collate = lambda d, B: ((None,) * 5, None, None)
        learned, copied = [], []
        for t in range(60):
            o.total_steps += 1
            o.count += 1
            a, b = counts["learn"], counts["copy"]
            o.train()
            learned.append(counts["learn"] - a)
            copied.append(counts["copy"] - b)
        np.testing.assert_array_equal(learned, fx[name + "|learned"], err_msg=name)
        np.testing.assert_array_equal(copied, fx[name + "|copied"], err_msg=name)
        checked += 1
    assert checked == 3


def _mine(obj):
    """Scalar / step / schedule fields of one of the device agents' parameter objects, in the fixture's form."""
    out = {}
    for k, v in vars(obj).items():
        if isinstance(v, (int, float, bool, str, type(None))):
            out[k] = v
        elif hasattr(v, "num_steps"):
            out[k] = [type(v).__name__, v.num_steps]
        elif isinstance(v, (tuple, list)) and all(isinstance(x, (int, float)) for x in v):
            out[k] = list(v)
        elif hasattr(v, "current_value") and hasattr(v, "initial_value"):
            out[k] = ["schedule", type(v).__name__, float(v.initial_value),
                      float(getattr(v, "final_value", v.initial_value)), int(getattr(v, "decay_steps", 0) or 0)]
    return out


def test_default_hyperparameters_equal_reference():
    """Every hyper-parameter the device agents' parameter classes carry under the reference's name has
    the reference's default (fixture: the reference's AgentParameters classes instantiated in the build
    container, tests/golden/defaults.json).  Fields this engine adds (reward_clipping, reward_rescale,
    resample_noise_per_pass, ...) have no counterpart and are not compared."""
    import json
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgentParameters
    from coach_amd.agents.ddpg_agent import DDPGAgentParameters
    from coach_amd.agents.dqn_agent import DDQNAgentParameters, DQNAgentParameters
    from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgentParameters
    from coach_amd.agents.td3_agent import TD3AgentParameters
    ref = json.load(open(os.path.join(HERE, "golden", "defaults.json")))
    # the network shapes are given as tuples of layer widths here and as scheme enums in the reference
    skip = {"embedder_scheme", "middleware_scheme", "observation_embedder_scheme", "action_embedder_scheme"}
    compared, bad = 0, []
    for P in (DQNAgentParameters, DDQNAgentParameters, ClippedPPOAgentParameters, DDPGAgentParameters,
              TD3AgentParameters, SoftActorCriticAgentParameters):
        ap, r = P(), ref[P.__name__]
        groups = [("algorithm", _mine(ap.algorithm), r["algorithm"])]
        for n, w in ap.network_wrappers.items():
            groups.append(("networks/" + n, _mine(w), r["networks"][n]))
        if getattr(ap, "exploration", None) is not None and r["exploration"]["class"] != "dict":
            groups.append(("exploration", _mine(ap.exploration), r["exploration"]))
        if getattr(ap, "memory", None) is not None:
            groups.append(("memory", _mine(ap.memory), r["memory"]))
        for gname, mine, theirs in groups:
            for k, v in mine.items():
                if k in skip or k not in theirs:
                    continue
                compared += 1
                same = v == theirs[k] or (isinstance(v, float) and isinstance(theirs[k], (int, float)) and
                                          abs(v - theirs[k]) <= 1e-12 * max(1.0, abs(v)))
                if not same:
                    bad.append("%s.%s.%s: here %r, reference %r" % (P.__name__, gname, k, v, theirs[k]))
    assert not bad, "\n".join(bad)
    assert compared >= 120, compared


def _layers(v):
    """A network shape as the fixture writes it: 'Medium' / 'Empty' scheme names or a list of widths."""
    if isinstance(v, str):
        return v.split(".")[-1]
    return [int(x) for x in v]


def _compare_agent_params(ap, ref, label, bad, extra_skip=()):
    skip = {"embedder_scheme", "middleware_scheme", "observation_embedder_scheme", "action_embedder_scheme"} | set(extra_skip)
    n = 0
    groups = [("algorithm", _mine(ap.algorithm), ref["algorithm"])]
    for name, w in ap.network_wrappers.items():
        groups.append(("networks/" + name, _mine(w), ref["networks"][name]))
    if getattr(ap, "exploration", None) is not None and ref["exploration"]["class"] != "dict":
        groups.append(("exploration", _mine(ap.exploration), ref["exploration"]))
    if getattr(ap, "memory", None) is not None:
        groups.append(("memory", _mine(ap.memory), ref["memory"]))
    for gname, mine, theirs in groups:
        for k, v in mine.items():
            if k in skip or k not in theirs:
                continue
            n += 1
            same = v == theirs[k] or (isinstance(v, float) and isinstance(theirs[k], (int, float)) and
                                      abs(v - theirs[k]) <= 1e-12 * max(1.0, abs(v)))
            if not same:
                bad.append("%s.%s.%s: here %r, reference %r" % (label, gname, k, v, theirs[k]))
    return n


def test_presets_equal_reference_presets():
    """The agent hyper-parameters of this package's presets against the reference preset modules
    (imported in the build container -> tests/golden/presets.json): CartPole_DQN, Atari_Dueling_DDQN,
    Mujoco_ClippedPPO preset for preset; Atari_DQN_with_PER / Mujoco_TD3 / Mujoco_SAC (the BASELINE
    configs C3 / C4 / C5, which bench.py builds from the parameter classes' defaults) against those
    defaults.  Environment and schedule sizes are this engine's (synthetic envs, smoke-sized runs)."""
    import importlib
    import json
    from coach_amd.agents.dqn_agent import DQNAgentParameters
    from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgentParameters
    from coach_amd.agents.td3_agent import TD3AgentParameters
    from coach_amd.memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplayParameters
    ref = json.load(open(os.path.join(HERE, "golden", "presets.json")))
    bad, n = [], 0
    for name in ("CartPole_DQN", "Atari_Dueling_DDQN", "Mujoco_ClippedPPO"):
        ap = importlib.import_module("coach_amd.presets." + name).graph_manager.agent_params
        r = ref[name]
        assert type(ap).__name__ == r["agent_class"], name
        n += _compare_agent_params(ap, r, name, bad)
        for wname, w in ap.network_wrappers.items():
            rn = r["networks"][wname]
            # network shape: scheme names or explicit widths, under this package's attribute names
            if hasattr(w, "middleware_scheme"):
                n += 1
                if _layers(w.middleware_scheme) != _layers(rn["middleware_layers"]):
                    bad.append("%s.%s middleware: here %r, reference %r" % (name, wname, w.middleware_scheme, rn["middleware_layers"]))
            if hasattr(w, "embedder_scheme"):
                n += 1
                if _layers(w.embedder_scheme) != _layers(rn["embedder_layers"]["observation"]):
                    bad.append("%s.%s embedder: here %r, reference %r" % (name, wname, w.embedder_scheme, rn["embedder_layers"]["observation"]))
            n += 1
            if w.activation_function != rn["middleware_activation"]:
                bad.append("%s.%s activation: here %r, reference %r" % (name, wname, w.activation_function, rn["middleware_activation"]))
            if hasattr(w, "heads_parameters"):
                n += 1
                mine = [[type(h).__name__, float(h.rescale_gradient_from_head_by_factor)] for h in w.heads_parameters]
                if mine != rn["heads"]:
                    bad.append("%s.%s heads: here %r, reference %r" % (name, wname, mine, rn["heads"]))
    from coach_amd.schedules import LinearSchedule
    ap = DQNAgentParameters()
    ap.memory = PrioritizedExperienceReplayParameters()
    assert _mine(ap.memory)["beta"] == ["schedule", "ConstantSchedule", 0.4, 0.4, 0]      # the class default ...
    ap.memory.beta = LinearSchedule(0.4, 1.0, 12500000)         # ... annealed by the preset (as bench.py c3 sets it)
    for label, params, rname in (("C3", ap, "Atari_DQN_with_PER"), ("C4", TD3AgentParameters(), "Mujoco_TD3"),
                                 ("C5", SoftActorCriticAgentParameters(), "Mujoco_SAC")):
        n += _compare_agent_params(params, ref[rname], label + "/" + rname, bad)
    assert not bad, "\n".join(bad)
    assert n >= 150, n


def test_ppo_should_train_equals_reference_with_full_episodes():
    """ClippedPPOAgent._should_train of the device agent (enough env-steps AND the lockstep episodes
    complete) against the reference's Agent._should_train with act_for_full_episodes = True
    (agent.py:662-699) on 80-step sequences with episodes of fixed length L, one env."""
    from coach_amd.agents.clipped_ppo_agent import ClippedPPOAgent
    from coach_amd.core_types import EnvironmentSteps
    fx = np.load(os.path.join(HERE, "golden", "cadence.npz"))
    names = sorted({k.split("|")[0] for k in fx.files if k.startswith("full_")})
    assert len(names) == 3
    for name in names:
        playing, L = (int(x) for x in fx[name + "|cfg"])

        class Memory(object):
            steps, T = 0, 10 ** 9
            def num_transitions(self): return self.steps
            def num_transitions_in_complete_episodes(self): return L * (self.steps // L)

        class Alg(object):
            num_consecutive_playing_steps = EnvironmentSteps(playing)

        class Ap(object):
            algorithm = Alg()
        # lockstep rule, and the rule for envs that end on different steps (complete episodes hold enough
        # transitions): for one env both are the reference's
        for ragged in (False, True):
            ag = ClippedPPOAgent.__new__(ClippedPPOAgent)
            ag.ap, ag.memory, ag.L, ag.ragged = Ap(), Memory(), L, ragged
            ag.total_steps_counter = ag.last_training_phase_step = 0
            opened = []
            for t in range(80):
                ag.total_steps_counter += 1
                ag.memory.steps += 1
                opened.append(int(ag._should_train()))
                if opened[-1]:
                    ag.memory.steps = 0                      # post_training_commands: memory.clean()
            np.testing.assert_array_equal(opened, fx[name + "|opened"], err_msg="%s ragged=%s" % (name, ragged))


@pytest.mark.parametrize("variant", ["uniform", "per"])
def test_oracle_agent_loop_equals_real_reference_agent_loop(variant):
    """oracle.agents.DQNAgentOracle (n_env = 1, reference_order = True) against the REAL reference
    DQNAgent object — built by its own __init__, with its own ExperienceReplay /
    PrioritizedExperienceReplay, EGreedy and Agent.observe / act / train, only the network replaced by
    the oracle stand-in — stepped through the reference's cycle for 12 heat-up + 120 training steps with
    a 32-transition buffer (so the FIFO wraps): every action, the number of transitions visible at every
    train(), every sampled transition and the final weights agree."""
    import random
    from coach_amd.schedules import LinearSchedule
    from oracle.agents import DQNAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "loop.npz"))
    D, A, L, B, CAP, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    arrays = {k[len("init|"):]: [fx[k]] for k in fx.files if k.startswith("init|")}
    random.seed(SEED)
    np.random.seed(SEED)
    o = DQNAgentOracle(arrays, SynthVecEnv(1, 1, D, L, 99), A, (D,), capacity=CAP, per={} if variant == "per" else None,
                       batch_size=B, playing_steps=1, target_every=10, huber=False, lr=1e-3,
                       epsilon_schedule=LinearSchedule(1.0, 0.1, 50))
    o.reference_order = True
    o.reset()
    actions, visible, keys = [], [], []
    collate = o._collate

    def logged(d, B_):
        out = collate(d, B_)
        keys.append([float(s[0]) for s in out[0][0]])
        visible.append(o._num_transitions())
        return out
    o._collate = logged
    for step in range(HEATUP + TRAIN):
        a = o.heatup_step() if step < HEATUP else o.act()
        actions.append(int(a[0]))
        if step >= HEATUP:
            o.train()
    np.testing.assert_array_equal(actions, fx[variant + "|actions"])
    np.testing.assert_array_equal(visible, fx[variant + "|visible"])
    np.testing.assert_array_equal(np.array(keys), fx[variant + "|keys"])
    _check_final(fx, variant + "|final", o.net, atol=1e-6)
    assert max(visible) == CAP                                   # the buffer did wrap


@pytest.mark.parametrize("variant", ["uniform", "per"])
def test_oracle_image_dqn_loop_with_stacking_equals_real_reference_agent_loop(variant):
    """C3 at loop level: oracle.agents.DQNAgentOracle in image mode (StackingOracle, reward clipping, Huber loss, the
    reference's store order) against the REAL reference DQNAgent behind ObservationStackingFilter(4) +
    RewardClippingFilter(-1, 1) on uint8 frames, uniform and prioritized replay with a wrapping 32-transition buffer
    (tests/golden/dqn_image_loop.npz): every action, the transitions visible at every train(), every sampled transition
    (keyed by the sum of its stacked state) and the final conv / dense weights."""
    import random
    from coach_amd.schedules import LinearSchedule
    from oracle.agents import DQNAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "dqn_image_loop.npz"))
    H, A, L, B, CAP, HEATUP, TRAIN, SEED, STACK = (int(x) for x in fx["hp"])
    arrays = {k.split("|")[1]: [fx[k]] for k in fx.files if k.startswith("init|")}
    random.seed(SEED)
    np.random.seed(SEED)
    o = DQNAgentOracle(arrays, SynthVecEnv(0, 1, H * H, L, 80), A, (H, H, STACK), stack=STACK, capacity=CAP,
                       per={} if variant == "per" else None, batch_size=B, playing_steps=1, target_every=10, huber=True,
                       lr=1e-3, epsilon_schedule=LinearSchedule(1.0, 0.1, 40), reward_clip=(-1.0, 1.0))
    o.reference_order = True
    o.reset(frame_hw=(H, H))
    actions, visible, keys = [], [], []
    collate = o._collate

    def logged(d, B_):
        out = collate(d, B_)
        keys.append([float(np.asarray(s_, dtype=np.float64).sum()) for s_ in out[0][0]])
        visible.append(o._num_transitions())
        return out
    o._collate = logged
    for step in range(HEATUP + TRAIN):
        a = o.heatup_step() if step < HEATUP else o.act()
        actions.append(int(a[0]))
        if step >= HEATUP:
            o.train()
    np.testing.assert_array_equal(visible, fx[variant + "|visible"])
    assert max(visible) == CAP                                   # the buffer did wrap
    np.testing.assert_array_equal(actions, fx[variant + "|actions"])
    np.testing.assert_array_equal(np.array(keys), fx[variant + "|keys"])
    _check_final(fx, variant + "|final", o.net, atol=2e-6)


@pytest.mark.parametrize("ragged", [False, True])
def test_oracle_ppo_agent_loop_equals_real_reference_agent_loop(ragged):
    """oracle.agents.ClippedPPOAgentOracle (act + train, n_env = 1) against the REAL reference
    ClippedPPOAgent object (own __init__, EpisodicExperienceReplay, Categorical exploration,
    _should_train with act_for_full_episodes, train -> fill_advantages -> train_network; network = oracle
    stand-in) over three rollouts: every sampled action, the steps at which training ran and the final
    weights agree."""
    import random
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "ppo_loop.npz"))
    D, A, L, B, PLAY, EPOCHS, STEPS, SEED = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    random.seed(SEED)
    np.random.seed(SEED)
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(1, 1, D, L, 77), A, batch_size=B, playing_steps=PLAY, epochs=EPOCHS,
                              lr=1e-3, reward_clip=None, ragged=ragged)
    o.reset()
    actions, trained_at, since = [], [], 0
    for step in range(STEPS):
        a, _ = o.act()
        actions.append(int(a[0]))
        since += 1
        # lockstep: enough steps and the episode is complete; ragged: the oracle's own rule (complete episodes hold
        # enough transitions) — for ONE env the two are the reference's rule
        if o.should_train() if ragged else (since >= PLAY and (step + 1) % L == 0):
            o.train()
            trained_at.append(step)
            since = 0
    np.testing.assert_array_equal(actions, fx["actions"])
    np.testing.assert_array_equal(trained_at, fx["trained_at"])
    _check_final(fx, "final", o.net, atol=2e-6)


def test_oracle_continuous_normalised_ppo_loop_equals_real_reference_agent_loop():
    """oracle.agents.ClippedPPOAgentOracle(continuous=True, normalize=True) against the REAL reference ClippedPPOAgent on
    a BoxActionSpace with the Mujoco_ClippedPPO pre-network filter (ObservationNormalizationFilter on numpy running
    statistics; tests/golden/ppoc_loop.npz): every sampled action (np.random.normal(mean, std) on observations
    normalised with the statistics as they were), the steps at which training ran, the running statistics after every
    training phase — count grows by 2 x the dataset: InputFilter.filter pushes the states and then the next states —
    and the final weights (policy_log_std included)."""
    import random
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "ppoc_loop.npz"))
    D, A, L, B, PLAY, EPOCHS, STEPS, SEED = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    random.seed(SEED)
    np.random.seed(SEED)
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(1, 1, D, L, 78), A, batch_size=B, playing_steps=PLAY, epochs=EPOCHS,
                              lr=1e-3, reward_clip=None, continuous=True, normalize=True)
    o.reset()
    actions, trained_at, stats, since = [], [], [], 0
    for step in range(STEPS):
        a, _ = o.act()
        actions.append(np.asarray(a[0], dtype=np.float64))
        since += 1
        if since >= PLAY and (step + 1) % L == 0:
            o.train()
            trained_at.append(step)
            stats.append(np.concatenate([[o.stats._count], o.stats._mean, o.stats._std]))
            since = 0
    np.testing.assert_array_equal(trained_at, fx["trained_at"])
    np.testing.assert_allclose(np.array(stats), fx["stats"], rtol=1e-12, atol=1e-12)
    assert abs(fx["stats"][0][0] - (2 * PLAY + 1e-2)) < 1e-9          # states AND next states were pushed
    np.testing.assert_allclose(np.array(actions), fx["actions"], rtol=0, atol=1e-6)
    _check_final(fx, "final", o.net, atol=2e-6)


def test_oracle_image_ppo_loop_with_stacking_and_reward_clipping_equals_real_reference_agent_loop():
    """The headline path at loop level: oracle.agents.ClippedPPOAgentOracle in image mode (StackingOracle per env, reward
    clipping) against the REAL reference ClippedPPOAgent behind ObservationStackingFilter(4) + RewardClippingFilter(-1, 1)
    on uint8 frames (tests/golden/ppo_image_loop.npz): the stack restarts at every episode start, transitions hold the
    stacked states, rewards are clipped before they are stored — every sampled action, the training steps and the final
    weights of the conv torso and both heads.  The device agent is compared with this oracle mode on the GPU
    (tests/test_ppo_agent.py), which chains the C2 path to the reference agent itself."""
    import random
    from oracle.agents import ClippedPPOAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "ppo_image_loop.npz"))
    H, A, L, B, PLAY, EPOCHS, STEPS, SEED, STACK = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    random.seed(SEED)
    np.random.seed(SEED)
    o = ClippedPPOAgentOracle(arrays, SynthVecEnv(0, 1, H * H, L, 79), A, stack=STACK, batch_size=B, playing_steps=PLAY,
                              epochs=EPOCHS, lr=1e-3, reward_clip=(-1.0, 1.0))
    o.reset(frame_hw=(H, H))
    actions, trained_at, since = [], [], 0
    for step in range(STEPS):
        a, _ = o.act()
        actions.append(int(a[0]))
        since += 1
        if since >= PLAY and (step + 1) % L == 0:
            o.train()
            trained_at.append(step)
            since = 0
    assert np.abs(fx["env_rewards"]).max() >= 1.0 and len(set(fx["env_rewards"].tolist())) > 1
    np.testing.assert_array_equal(actions, fx["actions"])
    np.testing.assert_array_equal(trained_at, fx["trained_at"])
    _check_final(fx, "final", o.net, atol=2e-6)


def test_product_training_waits_for_a_complete_episode_with_episodic_memory():
    """DDPG-style scheduling: an episodic memory holds only complete episodes (agent.py:576-584), so the
    reference opens no training phase before the first episode has ended; the device agents' host
    scheduling (_training_phases_due) against the reference's _should_train on the same sequences."""
    from coach_amd.agents.vector_agent import VectorOffPolicyAgent
    from coach_amd.core_types import EnvironmentSteps
    fx = np.load(os.path.join(HERE, "golden", "cadence.npz"))
    names = sorted({k.split("|")[0] for k in fx.files if k.startswith("episodic_")})
    assert len(names) == 2
    for name in names:
        playing, L = (int(x) for x in fx[name + "|cfg"])

        class Memory(object):
            count = 0
            def num_transitions(self): return self.count
            def num_transitions_in_complete_episodes(self): return L * (self.count // L)

        class Alg(object):
            act_for_full_episodes = False
            num_consecutive_playing_steps = EnvironmentSteps(playing)

        class Ap(object):
            algorithm = Alg()
        ag = VectorOffPolicyAgent.__new__(VectorOffPolicyAgent)
        ag.ap, ag.memory, ag.n_env = Ap(), Memory(), 1
        ag.total_steps_counter = ag.last_training_phase_step = 0
        opened = []
        for t in range(40):
            ag.total_steps_counter += 1
            ag.memory.count += 1
            opened.append(int(ag._training_phases_due() > 0))
        np.testing.assert_array_equal(opened, fx[name + "|opened"], err_msg=name)


def test_oracle_td3_agent_loop_equals_real_reference_agent_loop():
    """oracle.agents.TD3AgentOracle (n_env = 1) against the REAL reference TD3Agent object — own __init__,
    EpisodicExperienceReplay, AdditiveNoise exploration, Agent.observe / act / train, `episode length` updates at every
    episode end; only the networks replaced by the oracle stand-ins — over 12 heat-up and 30 training steps: every
    RECORDED action (the unclipped noisy one), the training iteration after every step, every sampled transition, the
    stored game_over flags (cleared on the time limit) and the final actor / critic weights."""
    import random
    from oracle.agents import TD3AgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "td3_loop.npz"))
    D, A, L, B, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    a_arr = {k: v for k, v in arrays.items() if k.startswith("actor/")}
    c_arr = {k: v for k, v in arrays.items() if k.startswith("critic/")}
    random.seed(SEED)
    np.random.seed(SEED)
    o = TD3AgentOracle(a_arr, c_arr, SynthVecEnv(1, 1, D, L, 55), A, batch_size=B)
    o.reset()
    iters, stored = [], []
    for step in range(HEATUP + TRAIN):
        before = len(o.memory.rows)
        if step < HEATUP:
            o.heatup_step()
        else:
            o.act()
        if len(o.memory.rows) != before or (o.memory.episodes and before == len(o.memory.rows) and False):
            stored.append([bool(r[3]) for r in o.memory.rows[-L:]])
        iters.append(o.training_iteration)
    np.testing.assert_array_equal(np.array(o.recorded_actions)[:, 0], fx["actions"])
    np.testing.assert_array_equal(iters, fx["iters"])
    np.testing.assert_array_equal(np.array(stored), fx["stored_game_over"])
    keys = np.array([[float(o.memory.rows[i][0][0]) for i in idx] for idx in o.sampled])
    np.testing.assert_array_equal(keys, fx["keys"])
    _check_final(fx, "final|actor", o.actor, atol=2e-6)
    _check_final(fx, "final|critic", o.critic, atol=2e-6)


def test_oracle_ddpg_agent_loop_equals_real_reference_agent_loop():
    """oracle.agents.DDPGAgentOracle (n_env = 1) against the REAL reference DDPGAgent object (own __init__,
    EpisodicExperienceReplay, OUProcess exploration restarted per episode, one update per env-step, targets mixed after
    every update): recorded actions, training iterations, sampled transitions, final weights."""
    import random
    from oracle.agents import DDPGAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "ddpg_loop.npz"))
    D, A, L, B, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    a_arr = {k: v for k, v in arrays.items() if k.startswith("actor/")}
    c_arr = {k: v for k, v in arrays.items() if k.startswith("critic/")}
    random.seed(SEED)
    np.random.seed(SEED)
    o = DDPGAgentOracle(a_arr, c_arr, SynthVecEnv(1, 1, D, L, 56), A, batch_size=B)
    o.reset()
    iters = []
    for step in range(HEATUP + TRAIN):
        o.heatup_step() if step < HEATUP else o.act()
        iters.append(o.training_iteration)
    np.testing.assert_array_equal(np.array(o.recorded_actions)[:, 0], fx["actions"])
    np.testing.assert_array_equal(iters, fx["iters"])
    keys = np.array([[float(o.memory.rows[i][0][0]) for i in idx] for idx in o.sampled])
    np.testing.assert_array_equal(keys, fx["keys"])
    _check_final(fx, "final|actor", o.actor, atol=2e-6)
    _check_final(fx, "final|critic", o.critic, atol=2e-6)


def test_oracle_sac_agent_loop_equals_real_reference_agent_loop():
    """oracle.agents.SACAgentOracle (n_env = 1, reference_order = True) against the REAL reference SoftActorCriticAgent
    object (own __init__, ExperienceReplay, heat-up, sampled actions, one update per env-step with the transitions
    visible at that train(), V target mixed after every update): recorded actions, training iterations, visible
    transitions, sampled transitions, final weights."""
    import random
    from oracle.agents import SACAgentOracle
    from oracle.synth_env import SynthVecEnv
    fx = np.load(os.path.join(HERE, "golden", "sac_loop.npz"))
    D, A, L, B, HEATUP, TRAIN, SEED = (int(x) for x in fx["hp"])
    arrays = {}
    for k in fx.files:
        if k.startswith("init|"):
            _, name, t = k.split("|")
            arrays.setdefault(name, {})[int(t)] = fx[k]
    arrays = {n: [tw[t] for t in sorted(tw)] for n, tw in arrays.items()}
    sub = lambda pre: {k: v for k, v in arrays.items() if k.startswith(pre)}
    random.seed(SEED)
    np.random.seed(SEED)
    o = SACAgentOracle(sub("policy/"), sub("q/"), sub("v/"), SynthVecEnv(1, 1, D, L, 57), A, batch_size=B)
    o.reference_order = True
    o.reset()
    iters = []
    for step in range(HEATUP + TRAIN):
        o.heatup_step() if step < HEATUP else o.act()
        iters.append(o.training_iteration)
    np.testing.assert_array_equal(iters, fx["iters"])
    np.testing.assert_array_equal(o.visible, fx["visible"])
    np.testing.assert_array_equal(np.array(o.sampled_keys), fx["keys"])
    np.testing.assert_allclose(np.array(o.recorded_actions)[:, 0], fx["actions"], rtol=0, atol=1e-7)
    _check_final(fx, "final|policy", o.policy, atol=2e-6)
    _check_final(fx, "final|q", o.q, atol=2e-6)
    _check_final(fx, "final|v", o.v, atol=2e-6)
