"""Where a workgroup of the fused input-gradient launch of conv3 / conv2 (rlx_conv32_input_grad) spends its time INSIDE the
Clipped-PPO minibatch update (C2 shapes): rlx_conv32_debug_stamps makes every workgroup record 10 ns ticks at entry, once the
first product's operands are staged, after its K loop, once the second product's operands are staged (= the first gather is
done), after its K loop, and at exit.  Eager updates; the stamps of the last one are read.
Usage: python tools/conv32_timeline.py [--tail16]      (--tail16: rlx_conv32_tail_tiles(1))"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from coach_amd.nn.networks import ClippedPPONet

from coach_amd.nn import graph as _G

_G.FUSE_CONV_INPUT_GRADS = True          # (off by default: profiles/r05_ab_conv32.txt)
lib = _rlx.lib()
if "--tail16" in sys.argv:
    lib.conv32_tail_tiles(1)
if "--ahead2" in sys.argv:
    lib.conv32_prefetch(2)
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
for _ in range(3):
    net.train_minibatch(obs, B, actions, adv, vt, old)
stamps = torch.zeros(2 * B * 2 * 8, dtype=torch.int64, device=dev)
lib.conv32_debug_stamps(stamps)
for _ in range(5):
    net.train_minibatch(obs, B, actions, adv, vt, old)
torch.cuda.synchronize()
lib.conv32_debug_stamps(None)
st = stamps.cpu().numpy().reshape(-1, 8)[:, :6].astype(np.float64) * 0.01          # us
t0 = st[:, 0].min()
names = ["stage dz3", "dcol3 = dz3 W3^T (36 jobs)", "gather -> dz2", "dcol2 = dz2 W2^T (32 jobs)",
         "gather -> dz1, stores"]
print("workgroups %d; launch span (first entry -> last exit) %.2f us; entries spread over %.2f us"
      % (len(st), st[:, 5].max() - t0, st[:, 0].max() - t0))
for half in (0, 1):
    d = np.diff(st[half::2], axis=1)
    print("half %d: workgroup life median %.2f us" % (half, np.median(st[half::2, 5] - st[half::2, 0])))
    for j, nm in enumerate(names):
        print("   %-32s median %6.2f   p10 %6.2f   p90 %6.2f" % (nm, np.median(d[:, j]), np.percentile(d[:, j], 10),
                                                                   np.percentile(d[:, j], 90)))
