"""Is an off-policy update loop HOST-bound or GPU-bound?  (C3 / C4 / C5 of bench.py.)

  (a) wall time per update of the agent's own train() path (draw, collate, learn_from_batch, target update);
  (b) the same updates with every captured hipGraph replayed back to back from a tight loop (no draws, no staging,
      no Python bookkeeping) = the GPU's own time;
  (c) the host half alone: train() with graph replays and library calls stubbed out is not possible without changing
      results, so it is reported as (a) with a device sync after every update minus (b) (serialised host + GPU).

    python tools/update_host_gpu_split.py --workload c4
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


class _Dist(object):
    rank, world_size, enabled = 0, 1, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4", choices=["c3", "c4", "c5"])
    ap.add_argument("--updates", type=int, default=2000)
    args = ap.parse_args()
    import bench
    from coach_amd.core_types import RunPhase
    dev = torch.device("cuda:0")
    agent = bench.build_off_policy(args.workload, dev, _Dist())
    desc, n_env, vsteps, heat = bench.OFF_POLICY[args.workload]
    agent.phase = RunPhase.HEATUP
    for _ in range(heat):
        agent.act()
    agent.phase = RunPhase.TRAIN
    while agent.training_iteration < 300:                 # warm: every graph variant captured
        agent.act(); agent.train()
    B = agent.batch_size

    def updates(n, sync_each=False):
        t0 = time.perf_counter()
        for _ in range(n):
            d = agent.memory.draw(B)
            agent.training_iteration += 1
            batch = agent.memory.collate(d, B)
            mix = any(nn.target is not None for nn in agent.networks.values()) and \
                agent._should_update_online_weights_to_target()
            agent._mix_rate = agent.ap.algorithm.rate_for_copying_weights_to_target if mix else None
            agent._mixed = set()
            agent.learn_from_batch(batch)
            if mix:
                for name, net in agent.networks.items():
                    if net.target is not None and name not in agent._mixed:
                        net.update_target(agent._mix_rate)
            agent._mix_rate = None
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    updates(200)
    a = updates(args.updates)
    a_sync = updates(args.updates, sync_each=True)
    graphs = [(k, g) for k, g in agent._graphs.items() if hasattr(g, "replay")]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = args.updates // 2
    for _ in range(reps):
        for _, g in graphs:
            g.replay()
    torch.cuda.synchronize()
    per_cycle = (time.perf_counter() - t0) / reps * 1e6
    out = {"workload": args.workload, "updates": args.updates, "us_per_update_train_path": round(a, 1),
           "us_per_update_with_a_sync_after_each": round(a_sync, 1),
           "graphs": [str(k) for k, _ in graphs],
           "us_per_cycle_all_graphs_replayed_back_to_back": round(per_cycle, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
