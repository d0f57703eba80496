"""Per-dispatch durations of the kernels of the C2 minibatch update, dispatch by dispatch (librlx's in-process timer =
the dispatch begin / end timestamps rocprofv3 reports): is a kernel's spread on a slow-class box bimodal, periodic, or a
uniform shift?  
    python tools/dispatch_histogram.py [--updates 200] [--kernel conv23_forward_kernel]
Prints, per kernel of the update: n, min / p10 / median / p90 / max in us, and for --kernel the whole series, its
histogram in 2 us bins and the autocorrelation at lags 1 .. 8 (periodicity)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from coach_amd.nn.networks import ClippedPPONet

ap = argparse.ArgumentParser()
ap.add_argument("--updates", type=int, default=200)
ap.add_argument("--kernel", default="conv23_forward_kernel")
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, A, shape = 64, 6, (84, 84, 4)
np.random.seed(0)
net = ClippedPPONet(dev, shape, A, seed=0)
rng = np.random.RandomState(0)
# 32 different minibatches, as the epoch loop sees them (a fresh set of frames per update)
obs = [torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev) for _ in range(32)]
actions = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
net.update_target(1.0)
old = net.policy_probs(obs[0], B, use_target=True, tag="old")
lib = _rlx.lib()
for i in range(8):
    net.train_minibatch(obs[i % 32], B, actions, adv, vt, old)
torch.cuda.synchronize()
lib.profile_begin(1 << 16)
for i in range(args.updates):
    net.train_minibatch(obs[i % 32], B, actions, adv, vt, old)
torch.cuda.synchronize()
n = ctypes.c_int()
lib.profile_end(ctypes.byref(n))
series = {}
order = []
for i in range(n.value):
    name, ms = ctypes.c_char_p(), ctypes.c_float()
    lib.profile_read(i, ctypes.byref(name), ctypes.byref(ms))
    k = name.value.decode()
    if k not in series:
        series[k] = []
        order.append(k)
    series[k].append(1e3 * ms.value)
out = {}
print("%-64s %6s %7s %7s %7s %7s %7s" % ("kernel (as launched)", "n", "min", "p10", "median", "p90", "max"))
for k in order:
    v = np.asarray(series[k])
    q = np.percentile(v, [0, 10, 50, 90, 100])
    out[k] = {"n": int(v.size), "min": q[0], "p10": q[1], "median": q[2], "p90": q[3], "max": q[4], "mean": float(v.mean())}
    print("%-64s %6d %7.1f %7.1f %7.1f %7.1f %7.1f" % (k[:64], v.size, *q))
for k in order:
    if args.kernel in k:
        v = np.asarray(series[k])
        print("\n%s: %d dispatches, mean %.1f us, std %.1f us" % (k, v.size, v.mean(), v.std()))
        lo = 2.0 * np.floor(v.min() / 2.0)
        bins = np.arange(lo, min(v.max(), lo + 80.0) + 2.0, 2.0)
        h, e = np.histogram(np.clip(v, None, bins[-1] - 1e-3), bins=bins)
        for c, b in zip(h, e):
            print("  %5.0f-%-5.0f us %5d %s" % (b, b + 2, c, "#" * int(60.0 * c / max(h.max(), 1))))
        d = v - v.mean()
        ac = [float((d[:-l] * d[l:]).sum() / (d * d).sum()) for l in range(1, 9)]
        print("  autocorrelation, lags 1..8:", " ".join("%.2f" % a for a in ac))
        print("  first 64 dispatches (us):", " ".join("%.0f" % x for x in v[:64]))
        out[k]["series_us"] = [round(float(x), 2) for x in v]
        out[k]["autocorr_1_8"] = ac
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f)
