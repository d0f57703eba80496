"""Where a C2 iteration spends its time: 32 vector steps of acting, fill_advantages, 10 epochs x 32 minibatch updates
(device events around the phases of ClippedPPOAgent, hipGraph replays as in bench.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


class _Dist(object):
    rank, world_size, enabled = 0, 1, False

    def barrier(self):
        pass


def main():
    import bench
    dev = torch.device("cuda:0")
    agent = bench.build_agent(dev, _Dist())
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def iteration(timed):
        marks = [ev() for _ in range(4)]
        marks[0].record()
        for _ in range(agent.steps_per_phase):
            agent.act()
        marks[1].record()
        assert agent._should_train()
        agent.networks["main"].update_target(1.0)
        agent.fill_advantages()
        marks[2].record()
        import random
        n = agent.memory.num_transitions()
        order = list(range(n))
        random.shuffle(order)
        agent.train_network(order, agent.ap.algorithm.optimization_epochs)
        agent.post_training_commands()
        marks[3].record()
        marks[3].synchronize()
        return [marks[i].elapsed_time(marks[i + 1]) for i in range(3)]
    for _ in range(4):
        iteration(False)
    runs = [iteration(True) for _ in range(5)]
    act, fill, train = (sum(r[i] for r in runs) / len(runs) for i in range(3))
    print(json.dumps({"ms_act_32_vector_steps": round(act, 3), "ms_fill_advantages": round(fill, 3),
                      "ms_train_320_updates": round(train, 3), "ms_total": round(act + fill + train, 3),
                      "us_per_update": round(1e3 * train / 320, 1), "us_per_act_step": round(1e3 * act / 32, 1)}))


if __name__ == "__main__":
    main()
