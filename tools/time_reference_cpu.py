"""CPU timing of the REFERENCE's own hot-path code (build container only: needs /root/reference).

Config C1 (CartPole_DQN sizes: obs 4, 2 actions, batch 32, 40 000-transition uniform replay, one update
per env-step) as the reference executes it per env-step — `ExperienceReplay.store` / `.sample`
(memories/non_episodic/experience_replay.py), `Batch` collation (core_types.py:405-649) and
`DQNAgent.learn_from_batch` (agents/dqn_agent.py:81-113) — with the repo's numpy oracle as the network
backend (tests/golden/_oracle_backend.py), because TensorFlow cannot be installed.  One thread, as the
reference pins it (OMP_NUM_THREADS=1, coach.py:666; TF intra/inter-op = 1, graph_manager.py:219-220).

    python tools/time_reference_cpu.py > profiles/r01_cpu_reference_c1_container.json

This is the `kind: "reference"` companion of bench.py's `cpu_baseline` (`kind: "port"`, timed on the GPU
box's host, where the reference tree does not exist)."""
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
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
from oracle.agents import DQNOracle  # noqa: E402
from rl_coach.agents.dqn_agent import DQNAgent  # noqa: E402
from rl_coach.core_types import Batch, Transition  # noqa: E402
from rl_coach.memories.memory import MemoryGranularity  # noqa: E402
from rl_coach.memories.non_episodic.experience_replay import ExperienceReplay  # noqa: E402


class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def main(seconds=20.0, heatup=1000):
    D, A, B = 4, 2, 32
    rng = np.random.RandomState(0)
    np.random.seed(0)

    def xavier(i, o):
        lim = np.sqrt(6.0 / (i + o))
        return [rng.uniform(-lim, lim, (i, o)).astype(np.float32)]
    arrays = {}
    for name, (i, o) in {"main/embedder/dense0": (D, 256), "main/middleware/dense0": (256, 512),
                         "main/q_head/dense": (512, A)}.items():
        arrays[name + "/kernel"], arrays[name + "/bias"] = xavier(i, o), [np.zeros(o, np.float32)]
    net = DQNOracle(arrays, (D,), A, lr=2.5e-4, huber=False)
    wrapper = OB.DQNWrapper(net)

    class Fake(DQNAgent):
        def __init__(self):
            pass
    f = Fake()
    f.ap = _Obj(network_wrappers={'main': _Obj(input_embedders_parameters={'observation': None})},
                algorithm=_Obj(discount=0.99))
    f.q_values = _Obj(add_sample=lambda v: None)
    f.memory = object()
    f.update_transition_priorities_and_get_weights = lambda errs, batch: None
    f.networks = {'main': wrapper}
    memory = ExperienceReplay(max_size=(MemoryGranularity.Transitions, 40000), allow_duplicates_in_batch_sampling=True)
    obs = rng.randn(D)
    t_in_ep, steps, t0 = 0, 0, None
    while True:
        if steps == heatup:
            t0 = time.perf_counter()
        if steps >= heatup:
            q = wrapper.online_network.predict({'observation': obs[None]})[0]      # act
            action = int(np.argmax(q)) if np.random.rand() > 0.1 else int(np.random.choice(A))
        else:
            action = int(np.random.choice(A))
        nxt = rng.randn(D)
        t_in_ep += 1
        done = t_in_ep == 200
        memory.store(Transition(state={'observation': obs}, action=action, reward=1.0,
                                next_state={'observation': nxt}, game_over=done))
        obs = rng.randn(D) if done else nxt
        t_in_ep = 0 if done else t_in_ep
        if steps >= heatup:
            f.learn_from_batch(Batch(memory.sample(B)))                            # one update per env-step
            if (steps - heatup) % 100 == 0:
                net.update_target(1.0)
        steps += 1
        if t0 is not None and time.perf_counter() - t0 > seconds:
            break
    n = steps - heatup
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "env-steps/sec (+ grad-updates/sec), C1, reference code on CPU",
        "value": round(n / dt, 1), "unit": "env-steps/s", "grad_updates_per_s": round(n / dt, 1),
        "kind": "reference", "cores": 1,
        "sample": "%d env-steps in %.1f s after %d heat-up steps: rl_coach ExperienceReplay.store/sample + Batch "
                  "collation + DQNAgent.learn_from_batch (reference code), numpy oracle as the network backend, "
                  "epsilon-greedy acting with one forward pass per step" % (n, dt, heatup),
        "host": platform.processor() or platform.machine(), "where": "build container (no GPU)"}))


if __name__ == "__main__":
    main()
