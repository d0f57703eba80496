"""Same-process A/B of the C2 minibatch update with the dense layers' Adam step overlapped with the next minibatch's
convolutions (ClippedPPOAgent.overlap_adam) or not: us per update from the captured epoch graphs, alternately.
    python tools/ab_c2_adam.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_c2


def main(rounds=4):
    agents = {}
    for name, ov in (("overlapped", True), ("in line", False)):
        a = ab_c2.build(False, True)
        a.overlap_adam = ov
        agents[name] = a
    for a in agents.values():
        for _ in range(3):
            ab_c2.train_ms(a)
    res = {k: [] for k in agents}
    for _ in range(rounds):
        for k, a in agents.items():
            res[k].append(ab_c2.train_ms(a))
    print(json.dumps({k: {"us_per_update": round(1e3 * min(v) / 320, 1), "all": [round(1e3 * x / 320, 1) for x in v]}
                      for k, v in res.items()}))
    w = [a.networks["main"].params.weights for a in agents.values()]
    import torch
    print("weights of the two agents equal:", bool(torch.equal(w[0], w[1])))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
