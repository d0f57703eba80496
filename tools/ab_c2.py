"""Same-box A/B of the C2 minibatch update (bench.py's agent): us per update with the round-3 changes switched off one
at a time — XCD-chunked tile order, direct convolution input gradients, tile-size thresholds.  Variants are measured alternately in ONE
process so that box-to-box differences (25 % between boxes of the pool) cancel.  python tools/ab_c2.py [rounds]"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


class _Dist(object):
    rank, world_size, enabled = 0, 1, False

    def barrier(self):
        pass


def build(direct, defer, tuning=(192, 192, -1), heads_one_launch=True):
    import bench
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    G.DIRECT_CONV_INPUT_GRAD = direct
    ClippedPPONet.HEADS_LOSS_BACKWARD_ONE_LAUNCH = heads_one_launch
    agent = bench.build_agent(torch.device("cuda:0"), _Dist())
    if not defer:
        agent.networks["main"].ctx.begin_deferring = lambda: False
    agent._ab = (direct, defer, tuning, heads_one_launch)
    return agent


def train_ms(agent):
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    from coach_amd.nn.networks import ClippedPPONet
    G.DIRECT_CONV_INPUT_GRAD = agent._ab[0]
    ClippedPPONet.HEADS_LOSS_BACKWARD_ONE_LAUNCH = agent._ab[3]
    _rlx.lib().gemm_tuning(*agent._ab[2])            # read when the update's graphs are captured
    for _ in range(agent.steps_per_phase):
        agent.act()
    agent.networks["main"].update_target(1.0)
    agent.fill_advantages()
    order = list(range(agent.memory.num_transitions()))
    random.shuffle(order)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    agent.train_network(order, agent.ap.algorithm.optimization_epochs)
    e1.record()
    agent.post_training_commands()
    e1.synchronize()
    return e0.elapsed_time(e1)


def main(rounds=4):
    variants = {"default (head losses + heads' backward as one launch)": (False, True, (192, 192, -1), True),
                "head losses and heads' backward as two launches": (False, True, (192, 192, -1), False)}
    agents = {k: build(*v) for k, v in variants.items()}
    for a in agents.values():
        for _ in range(3):
            train_ms(a)
    from coach_amd import _rlx
    res = {k: [] for k in agents}
    for _ in range(rounds):
        for k, a in agents.items():
            res[k].append(train_ms(a))
    out = {k: {"us_per_update": round(1e3 * min(v) / 320, 1), "all": [round(1e3 * x / 320, 1) for x in v]} for k, v in res.items()}
    _rlx.lib().gemm_tuning(192, 192, -1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
