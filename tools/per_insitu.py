"""The prioritized-replay kernels INSIDE the C3 workload (DQN + PER, 2^20 transitions pre-filled, 64 envs): in-update
durations from the in-process kernel timer over eager steps, for three placements of rlx_per_touch:
   none            : the round-3 flow
   before-update   : touch the update's tree nodes right before rlx_per_update (diagnostic: how much of the in-situ
                     update time is cold tree lines)
   after-sample    : touch them right behind the sample — ~one DQN update (200 us of traffic) before they are needed
    python tools/per_insitu.py [steps=48]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from coach_amd import _rlx
from coach_amd.core_types import RunPhase


class _Dist(object):
    rank, world_size, enabled = 0, 1, False


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
agent = bench.build_off_policy("c3", dev, _Dist())
bench.prefill_c3(agent)
agent.phase = RunPhase.TRAIN
agent.use_graphs = False
mem, lib = agent.memory, _rlx.lib()
sink = torch.zeros(256, dtype=torch.float64, device=dev)
orig_update, orig_collate = mem.update_priorities, mem.collate
mode = ["none"]


def touch(idx):
    lib.per_touch(mem.sum_tree, mem.min_tree, mem.max_tree, mem.power_of_2_size, idx, int(idx.numel()), sink,
                  _rlx.current_stream())


def update(indices, errors):
    if mode[0] == "before-update":
        touch(indices)
    return orig_update(indices, errors)


def collate(drawn, size):
    b = orig_collate(drawn, size)
    if mode[0] == "after-sample":
        touch(b.info("idx"))
    return b


mem.update_priorities, mem.collate = update, collate
for _ in range(16):
    agent.act(); agent.train()
for m in ("none", "before-update", "after-sample", "none"):
    mode[0] = m
    for _ in range(8):
        agent.act(); agent.train()
    torch.cuda.synchronize()
    with _rlx.KernelTimer(1 << 16) as t:
        it0 = agent.training_iteration
        for _ in range(steps):
            agent.act(); agent.train()
        torch.cuda.synchronize()
    updates = agent.training_iteration - it0
    agg = {}
    for name, us in t.records:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1; a[1] += us
    total = sum(v[1] for v in agg.values())
    print("== touch %-14s %d updates in %d steps; all kernels %.1f us per update" % (m, updates, steps, total / max(updates, 1)))
    for name, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if name.startswith("per_") or name.startswith("kernel"):
            print("   %-44s %6d x %7.2f us" % (name[:44], c, us / c))
agent.check_status()
