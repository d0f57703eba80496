"""Per-workgroup timeline of the GEMM launches of one Clipped-PPO minibatch update (C2 shapes), replayed from a
hipGraph: rlx_gemm_debug_stamps makes every workgroup of the tiled kernel record wall-clock ticks (10 ns) at entry,
after its first slab is staged, after its main loop and at exit.  Prints, per launch: grid, splits, when the first /
last workgroup started and ended relative to the previous launch's last exit, and the median phase lengths.
Usage: python tools/gemm_timeline.py  [REPLAYS=3]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from coach_amd.nn.networks import ClippedPPONet

if "--lib" in sys.argv:               # another build of the library (tools/ab_lib.sh)
    _rlx.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
if "--pipeline" in sys.argv:          # rlx_gemm_pipeline: 1 = LDS-DMA ring (default), 0 = register-staged
    _rlx.lib().gemm_pipeline(int(sys.argv[sys.argv.index("--pipeline") + 1]))
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
torch.cuda.synchronize()

lib = _rlx.lib()
cap = 1 << 20
stamps = torch.zeros(cap, dtype=torch.int64, device=dev)
lib.gemm_debug_stamps(stamps, cap)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        net.train_minibatch(obs, B, actions, adv, vt, old)
    calls = np.zeros((512, 9), dtype=np.int64)
    n = ctypes.c_int(0)
    lib.gemm_debug_calls(calls.ctypes.data, 512, ctypes.addressof(n))
    lib.gemm_debug_stamps(None, 0)
    for _ in range(int(os.environ.get("REPLAYS", "3"))):
        graph.replay()
torch.cuda.synchronize()
st = stamps.cpu().numpy()
calls = calls[:n.value]
t_origin = None
prev_end = None
print("ticks are 10 ns; all times in us.  gap = first entry - previous GEMM launch's last exit (other kernels "
      "may run in between)")
print("%-34s %-14s %6s %6s | %7s %7s %7s | %6s %6s %6s | %7s" %
      ("M x N x K x batch", "grid", "splits", "WGs", "gap", "spread", "span", "prolog", "main", "epilog", "WG life"))
total_span = 0.0
for M, N, K, batch, splits, gx, gy, gz, off in calls:
    nwg = gx * gy * gz
    s = st[off:off + 4 * nwg].reshape(nwg, 4).astype(np.float64) * 0.01
    ok = (s > 0).all(axis=1)
    s = s[ok]
    if not len(s):
        continue
    first, last_start, last_end = s[:, 0].min(), s[:, 0].max(), s[:, 3].max()
    gap = first - prev_end if prev_end is not None else float("nan")
    med = np.median(np.diff(s, axis=1), axis=0)
    life = np.median(s[:, 3] - s[:, 0])
    print("%-34s %-14s %6d %6d | %7.2f %7.2f %7.2f | %6.2f %6.2f %6.2f | %7.2f" %
          ("%d x %d x %d x %d" % (M, N, K, batch), "%dx%dx%d" % (gx, gy, gz), splits, nwg, gap, last_start - first,
           last_end - first, med[0], med[1], med[2], life))
    total_span += last_end - first
    prev_end = last_end
    if t_origin is None:
        t_origin = first
print("sum of GEMM spans %.1f us; first GEMM entry to last GEMM exit %.1f us" % (total_span, prev_end - t_origin))
