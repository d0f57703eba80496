"""One eager Clipped-PPO minibatch update (C2 shapes: 64 x 84x84x4 uint8, two towers) — a short
target for `rocprofv3 --pmc` passes (FETCH_SIZE / WRITE_SIZE per GEMM launch); the full bench replays
hipGraphs with tens of thousands of dispatches, far too many for serialized counter collection.
Usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/ppo_update_once.py
       [--xcd-mode M]   (rlx_gemm_tuning's third argument; default: the library's default)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

if "--lib" in sys.argv:                      # another build of the library (tools/ab_lib.sh)
    import coach_amd._rlx as _rlx_mod
    _rlx_mod.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from coach_amd.nn.networks import ClippedPPONet

if "--xcd-mode" in sys.argv:
    from coach_amd import _rlx
    _rlx.lib().gemm_tuning(192, 192, int(sys.argv[sys.argv.index("--xcd-mode") + 1]))
dev = torch.device("cuda:0")
B, A, shape = 64, 6, (84, 84, 4)
np.random.seed(0)
net = ClippedPPONet(dev, shape, A, seed=0)
rng = np.random.RandomState(0)
obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
actions = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
net.update_target(1.0)
old = net.policy_probs(obs, B, use_target=True, tag="old")
for _ in range(int(os.environ.get("REPS", "2"))):
    net.train_minibatch(obs, B, actions, adv, vt, old)
torch.cuda.synchronize()
print("ok", net.scalars[:6].cpu().numpy())
