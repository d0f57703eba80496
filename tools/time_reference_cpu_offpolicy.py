"""CPU timing of the REFERENCE's own per-update code for the off-policy BASELINE workloads (build container only: needs
/root/reference), the `kind: "reference"` companion of bench.py's `cpu_baseline` for C3 / C4 / C5.

What runs is rl_coach's code: the memory class of the preset (`ExperienceReplay`, `PrioritizedExperienceReplay`,
`EpisodicExperienceReplay`) stores real `Transition` objects and samples, `Batch` collates, and the agent's own
`learn_from_batch` (agents/{dqn,td3,soft_actor_critic}_agent.py) drives the networks — which are the repo's numpy oracle
behind tests/golden/_oracle_backend.py, because TensorFlow cannot be installed.  Shapes are BASELINE.json's:
  c3  DQN, 84x84x4 uint8 (LazyStack-free ndarray states), 4 actions, B = 32, prioritized replay (capacity 2^14 here:
      a million 28 KB python transitions do not fit a container; the tree depth, not the payload, is what sampling costs)
  c4  TD3, obs 17, act 6, B = 100, twin critic 400-300, policy every 2nd update
  c5  SAC, obs 376, act 17, B = 256, 256-256 networks
One thread, as the reference pins it (OMP_NUM_THREADS=1, coach.py:666; TF intra/inter-op = 1, graph_manager.py:219-220).
One update per env-step, as the bench workloads are scheduled; acting is not included (the update dominates).

    python tools/time_reference_cpu_offpolicy.py --workload c4 > profiles/r02_cpu_reference_c4_container.json
"""
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
import argparse
import json
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _refstub  # noqa: E402

_refstub.install()
import _oracle_backend as OB  # noqa: E402
from rl_coach.core_types import Batch, Transition  # noqa: E402
from rl_coach.memories.memory import MemoryGranularity  # noqa: E402


class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _xavier(rng, spec):
    out = {}
    for name, (i, o, towers) in spec.items():
        lim = np.sqrt(6.0 / (i + o))
        out[name + "/kernel"] = [rng.uniform(-lim, lim, (i, o)).astype(np.float32) for _ in range(towers)]
        out[name + "/bias"] = [np.zeros(o, np.float32) for _ in range(towers)]
    return out


def _fake(cls):
    class Fake(cls):
        def __init__(self):
            pass
    return Fake()


_keys = lambda *names: _Obj(input_embedders_parameters={n: None for n in names})
_sink = _Obj(add_sample=lambda v: None)


def build_c4(rng):
    from oracle import ac_nets as O
    from rl_coach.agents.td3_agent import TD3Agent
    from rl_coach.memories.episodic.episodic_experience_replay import EpisodicExperienceReplay
    from rl_coach.spaces import BoxActionSpace
    D, A, B = 17, 6, 100
    actor = O.ActorOracle(_xavier(rng, {"actor/embedder/dense0": (D, 400, 1), "actor/middleware/dense0": (400, 300, 1),
                                        "actor/ddpg_actor_head/fc_mean": (300, A, 1)}), 1.0, lr=1e-3)
    critic = O.CriticOracle(_xavier(rng, {"critic/middleware/dense0": (D + A, 400, 2),
                                          "critic/middleware/dense1": (400, 300, 2),
                                          "critic/v_head/output": (300, 1, 2)}), streams=2, lr=1e-3)
    f = _fake(TD3Agent)
    f.ap = _Obj(network_wrappers={'actor': _keys('observation'), 'critic': _keys('observation', 'action')},
                algorithm=_Obj(discount=0.99, use_non_zero_discount_for_terminal_states=False, clip_critic_targets=None,
                               policy_noise=0.2, noise_clipping=0.5, update_policy_every_x_episode_steps=2))
    f.TD_targets_signal = _sink
    f.spaces = _Obj(action=BoxActionSpace(A, -1.0, 1.0))
    f.networks = {'actor': OB.ActorWrapper(actor), 'critic': OB.CriticWrapper(critic, A)}
    memory = EpisodicExperienceReplay(max_size=(MemoryGranularity.Transitions, 1000000))

    def after(it):
        f.training_iteration = it + 1
        if it % 2 == 1:
            actor.mix_target(0.005); critic.mix_target(0.005)
    sample = lambda: rng.uniform(-1, 1, A).astype(np.float32)
    return f, memory, D, B, sample, after, 100, "rl_coach EpisodicExperienceReplay.store_episode / sample + Batch " \
        "collation + TD3Agent.learn_from_batch (critic every update, actor + target mixing every 2nd)"


def build_c5(rng):
    from oracle import ac_nets as O
    from rl_coach.agents.soft_actor_critic_agent import SoftActorCriticAgent
    from rl_coach.memories.non_episodic.experience_replay import ExperienceReplay
    D, A, B = 376, 17, 256
    pol = O.SACPolicyOracle(_xavier(rng, {"policy/embedder/dense0": (D, 256, 1), "policy/middleware/dense0": (256, 256, 1),
                                          "policy/sac_policy_head/policy_mu_logsig": (256, 2 * A, 1)}))
    qn = O.SACQOracle(_xavier(rng, {"q/q_head/obs_fc": (D, 256, 2), "q/q_head/act_fc": (A, 256, 2),
                                    "q/q_head/fc1": (256, 256, 2), "q/q_head/q_output": (256, 1, 2)}))
    vn = O.SACValueOracle(_xavier(rng, {"v/embedder/dense0": (D, 256, 1), "v/middleware/dense0": (256, 256, 1),
                                        "v/v_values_head/output": (256, 1, 1)}))
    f = _fake(SoftActorCriticAgent)
    f.ap = _Obj(network_wrappers={n: _keys('observation') for n in ('policy', 'q', 'v')}, algorithm=_Obj(discount=0.99))
    for sig in ("policy_means", "policy_logsig", "policy_logprob_sampled", "q1_values", "q2_values", "policy_grads",
                "v_onl_ys", "v_tgt_ns", "TD_err1", "TD_err2"):
        setattr(f, sig, _sink)
    pw = OB.SACPolicyWrapper(pol, A)
    f.networks = {'policy': pw, 'q': OB.SACQWrapper(qn), 'v': OB.SACValueWrapper(vn)}
    memory = ExperienceReplay(max_size=(MemoryGranularity.Transitions, 1000000), allow_duplicates_in_batch_sampling=True)

    def after(it):
        pw.normals = []
        vn.mix_target(0.005)
    sample = lambda: rng.uniform(-1, 1, A).astype(np.float32)
    return f, memory, D, B, sample, after, None, "rl_coach ExperienceReplay.store / sample + Batch collation + " \
        "SoftActorCriticAgent.learn_from_batch (policy, V and twin-Q updates, three policy passes)"


def build_c3(rng):
    from oracle.agents import DQNOracle
    from rl_coach.agents.dqn_agent import DQNAgent
    from rl_coach.memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplay
    from rl_coach.schedules import LinearSchedule
    A, B = 4, 32
    shape = (84, 84, 4)
    spec = {"main/embedder/conv0": (8 * 8 * 4, 32, 1), "main/embedder/conv1": (4 * 4 * 32, 64, 1),
            "main/embedder/conv2": (3 * 3 * 64, 64, 1), "main/middleware/dense0": (7 * 7 * 64, 512, 1),
            "main/q_head/dense": (512, A, 1)}
    net = DQNOracle(_xavier(rng, spec), shape, A, lr=2.5e-4, huber=True)
    f = _fake(DQNAgent)
    f.ap = _Obj(network_wrappers={'main': _keys('observation')}, algorithm=_Obj(discount=0.99))
    f.q_values = _sink
    memory = PrioritizedExperienceReplay(max_size=(MemoryGranularity.Transitions, 1 << 14), alpha=0.6,
                                         beta=LinearSchedule(0.4, 1.0, 12500000), epsilon=1e-6,
                                         allow_duplicates_in_batch_sampling=True)
    f.memory = memory
    f.shared_memory = False
    f.networks = {'main': OB.DQNWrapper(net)}

    def after(it):
        if it % 2500 == 0:
            net.update_target(1.0)
    sample = lambda: int(rng.randint(0, A))
    return f, memory, shape, B, sample, after, None, "rl_coach PrioritizedExperienceReplay (sum / min / max trees) " \
        "store / sample / update_priorities + Batch collation + DQNAgent.learn_from_batch on 84x84x4 uint8 states"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["c3", "c4", "c5"])
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--prefill", type=int, default=4096)
    args = ap.parse_args()
    rng = np.random.RandomState(0)
    np.random.seed(0)
    f, memory, D, B, sample_action, after, ep_len, what = {"c3": build_c3, "c4": build_c4, "c5": build_c5}[args.workload](rng)
    image = isinstance(D, tuple)
    obs_of = (lambda: rng.randint(0, 256, size=D).astype(np.uint8)) if image else (lambda: rng.randn(D).astype(np.float32))

    def transition(done):
        return Transition(state={'observation': obs_of()}, action=sample_action(), reward=float(rng.randn()),
                          next_state={'observation': obs_of()}, game_over=done)
    if ep_len is not None:                                  # episodic memory: whole episodes
        from rl_coach.core_types import Episode
        for _ in range(args.prefill // ep_len):
            ep = Episode()
            for t in range(ep_len):
                ep.insert(transition(t == ep_len - 1))
            memory.store_episode(ep)
    else:
        for i in range(args.prefill):
            memory.store(transition(i % 200 == 199))
    for it in range(3):                                     # warm
        f.training_iteration = it + 1
        f.learn_from_batch(Batch(memory.sample(B)))
        after(it)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        if ep_len is None:
            memory.store(transition(False))                 # one env-step's store per update
        f.learn_from_batch(Batch(memory.sample(B)))
        after(n)
        n += 1
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "env-steps/sec (= grad-updates/sec at one update per env-step), %s, reference code on CPU" % args.workload.upper(),
        "value": round(n / dt, 2), "unit": "env-steps/s", "grad_updates_per_s": round(n / dt, 2),
        "kind": "reference", "cores": 1,
        "sample": "%d updates in %.1f s after a %d-transition prefill: %s; numpy oracle as the network backend"
                  % (n, dt, args.prefill, what),
        "host": platform.processor() or platform.machine(), "where": "build container (no GPU)"}))


if __name__ == "__main__":
    main()
