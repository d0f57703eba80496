"""Where a workgroup of the fused conv2 -> conv3 forward launch (rlx_conv23_forward) spends its time INSIDE the Clipped-PPO
minibatch update (C2 shapes): rlx_conv23_debug_stamps makes every workgroup record 10 ns ticks at entry, once conv2's
first weight slab and the conv1 rows are staged, after conv2's slab loop, once conv3's first slab is staged (= conv2's
epilogue done), after conv3's slab loop, and at exit.  Eager updates; the stamps of the last one are read.
With conv1 in front (rlx_conv123_forward, the default) two more stamps: the frame rows staged, conv1's K loop done.
Usage: python tools/conv23_timeline.py [--depth D] [--pair]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from coach_amd.nn.networks import ClippedPPONet

lib = _rlx.lib()
if "--pair" in sys.argv:
    from coach_amd.nn import graph as _G
    _G.FUSE_CONV_FIRST = False
if "--depth" in sys.argv:
    lib.conv23_depth(int(sys.argv[sys.argv.index("--depth") + 1]),
                     int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 1)
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
lib.conv23_debug_stamps(stamps)
for _ in range(5):
    net.train_minibatch(obs, B, actions, adv, vt, old)
torch.cuda.synchronize()
lib.conv23_debug_stamps(None)
raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64) * 0.01               # us
if raw[:, 6].any():        # conv1 in front (rlx_conv123_forward): entry, frame rows staged, conv1's K loop done, then as below
    st = raw[:, [0, 6, 7, 1, 2, 3, 4, 5]]
    names = ["stage frame rows + slab 0", "conv1 K loop", "conv1 epilogue (+ slab 4)", "conv2 slab loop",
             "conv2 epilogue (+ slab 20)", "conv3 slab loop", "conv3 epilogue"]
else:
    st = raw[:, :6]
    names = ["stage in1 + slab 0", "conv2 slab loop", "conv2 epilogue (+ slab 16)", "conv3 slab loop", "conv3 epilogue"]
t0 = st[:, 0].min()
print("workgroups %d; launch span (first entry -> last exit) %.2f us; entries spread over %.2f us"
      % (len(st), st[:, -1].max() - t0, st[:, 0].max() - t0))
for half in (0, 1):
    d = np.diff(st[half::2], axis=1)
    print("half %d (%s): workgroup life median %.2f us" % (half, "conv3 rows 0-2" if half == 0 else "conv3 rows 3-6",
                                                           np.median(st[half::2, -1] - st[half::2, 0])))
    for j, nm in enumerate(names):
        print("   %-28s median %6.2f   p10 %6.2f   p90 %6.2f" % (nm, np.median(d[:, j]), np.percentile(d[:, j], 10),
                                                               np.percentile(d[:, j], 90)))
