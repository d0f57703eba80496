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


def _transition(i):
    return Transition(state={'observation': np.array([i])}, action=0, reward=float(i),
                      next_state={'observation': np.array([i + 1])}, game_over=False)


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


GROUPS = {"per": gen_per}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GROUPS)
    for n in names:
        GROUPS[n]()
