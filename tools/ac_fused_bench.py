"""Fused vs layer-by-layer TD3 / SAC update on the BASELINE shapes: device time per update (events around replays of the
captured update graphs), launches per update (librlx's in-process kernel timer over eager updates), per-kernel durations.
    python tools/ac_fused_bench.py [td3|sac] [--iters 400]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx


def make_agent(kind, dev, fused):
    from coach_amd.environments.synthetic_vector_environment import (
        SyntheticVectorEnvironment, SyntheticVectorEnvironmentParameters)
    from coach_amd.memories.memory import MemoryGranularity
    if kind == "td3":
        from coach_amd.agents.td3_agent import TD3Agent as Cls, TD3AgentParameters as P
        D, A, B = 17, 6, 100
    else:
        from coach_amd.agents.soft_actor_critic_agent import SoftActorCriticAgent as Cls, SoftActorCriticAgentParameters as P
        D, A, B = 376, 17, 256
    Cls.FUSED_UPDATE = fused
    ep = SyntheticVectorEnvironmentParameters("vector", 4, (D,), None, action_dim=A, episode_length=8, seed=5)
    env = SyntheticVectorEnvironment(ep, dev)
    params = P()
    params.memory.max_size = (MemoryGranularity.Transitions, 1024)
    return Cls(params, env, dev, use_graphs=False), D, A, B


class Batch:
    def __init__(self, dev, rng, B, D, A, scale=1.0):
        s = (rng.randn(B, D) * scale).astype(np.float32)
        ns = (rng.randn(B, D) * scale).astype(np.float32)
        both = torch.as_tensor(np.stack([s, ns]), device=dev)
        self._info = {"states_pair": both}
        self._states, self._next_states = {"observation": both[0]}, {"observation": both[1]}
        self._a = torch.as_tensor(rng.uniform(-1, 1, (B, A)).astype(np.float32), device=dev)
        self._r = torch.as_tensor(rng.randn(B).astype(np.float32), device=dev)
        self._d = torch.as_tensor((rng.rand(B) < 0.1).astype(np.uint8), device=dev)

    def actions(self): return self._a
    def rewards(self): return self._r
    def game_overs(self): return self._d


def update_fn(kind, ag, b, it):
    if kind == "td3":
        def f():
            ag._critic_device(b)
            if it[0] % 2 == 0:
                ag._actor_device(b)
            it[0] += 1
        return f
    return lambda: ag._learn_device(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", nargs="?", default="td3")
    ap.add_argument("--iters", type=int, default=400)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _rlx.lib()
    for fused in (False, True):
        ag, D, A, B = make_agent(args.kind, dev, fused)
        rng = np.random.RandomState(0)
        b = Batch(dev, rng, B, D, A, 0.05 if D > 100 else 1.0)
        if args.kind == "td3":
            ag.noise.copy_(torch.as_tensor(rng.normal(0, 0.2, (B, A)), device=dev))
        else:
            ag.normals.copy_(torch.as_tensor(rng.standard_normal((3, B, A)), device=dev))
        it = [1]
        f = update_fn(args.kind, ag, b, it)
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        # launches / kernel durations over 2 eager updates (one with, one without the actor step for TD3)
        lib.profile_begin(4096)
        f(); f()
        torch.cuda.synchronize()
        n = ctypes.c_int()
        lib.profile_end(ctypes.byref(n))
        recs = []
        for i in range(n.value):
            name, ms = ctypes.c_char_p(), ctypes.c_float()
            lib.profile_read(i, ctypes.byref(name), ctypes.byref(ms))
            recs.append((name.value.decode(), 1e3 * ms.value))
        # graph: two consecutive updates captured (so that TD3's alternation is inside), replayed
        g = torch.cuda.CUDAGraph()
        it[0] = 1
        with torch.cuda.graph(g):
            f(); f()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters // 2):
            g.replay()
        e1.record(); e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (args.iters // 2 * 2)
        print("%s %-8s: %6.1f us per update in a graph; %4.1f library kernels per update, sum of kernel durations %6.1f us"
              % (args.kind, "fused" if fused else "unfused", us, len(recs) / 2.0, sum(t for _, t in recs) / 2.0))
        if fused:
            for name, t in recs:
                print("      %-60s %6.1f us" % (name[:60], t))
        ag.check_status()


if __name__ == "__main__":
    main()
