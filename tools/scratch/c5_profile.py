import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from coach_amd.core_types import RunPhase
class _D: rank, world_size, enabled = 0, 1, False
agent = bench.build_off_policy("c5", torch.device("cuda:0"), _D())
agent.phase = RunPhase.HEATUP
for _ in range(4): agent.act()
agent.phase = RunPhase.TRAIN
agent.act(); agent.train(); torch.cuda.synchronize()
t0 = time.perf_counter(); agent.act(); agent.train(); torch.cuda.synchronize(); print("one step s", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable()
agent.act(); agent.train(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
