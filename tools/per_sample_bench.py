"""rlx_per_sample on a 2^20-leaf tree, B = 32: kernel duration (the library's in-process timer) by how many levels are
descended from LDS and by where the uniform draws are read from (a device buffer / the pinned host slot)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from coach_amd import _rlx                                                                          # noqa: E402

dev = torch.device("cuda:0")
lib = _rlx.lib()
cap, B = 1 << 20, 32
n = 2 * cap - 1
rng = np.random.RandomState(0)
sum_t = torch.zeros(n, dtype=torch.float64, device=dev)
min_t = torch.zeros(n, dtype=torch.float64, device=dev)
max_t = torch.zeros(n, dtype=torch.float64, device=dev)
maxp = torch.zeros(1, dtype=torch.float64, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
s = _rlx.current_stream()
lib.per_init(sum_t, min_t, max_t, cap, maxp, s)
lib.per_store(sum_t, min_t, max_t, cap, 0, 64, 0.6, maxp, status, s)
idx_all = torch.arange(cap, dtype=torch.int32, device=dev)
err = torch.from_numpy(np.abs(rng.randn(cap)) + 1e-6).to(dev)
for i in range(0, cap, 256):        # fill every leaf
    lib.per_update(sum_t, min_t, max_t, cap, idx_all[i:i + 256], err[i:i + 256], 256, 0.6, 1e-6, maxp, status, s)
torch.cuda.synchronize()
oi = torch.zeros(B, dtype=torch.int32, device=dev)
ow = torch.zeros(B, dtype=torch.float64, device=dev)
rows = torch.zeros(B, dtype=torch.int32, device=dev)
u_host = torch.from_numpy(rng.rand(B)).pin_memory()
u_dev = u_host.to(dev)
junk = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
ref = None
for top in (11, 8, 5, 2, 0):
    lib.per_sample_top_steps(top)
    for name, u in (("device", u_dev), ("pinned", u_host)):
        ts = []
        for rep in range(40):
            junk.add_(1.0)                       # evict the tree from L2 as an update's kernels would
            with _rlx.KernelTimer(8) as t:
                lib.per_sample(sum_t, min_t, cap, u, B, float(cap), 0.4, oi, ow, None, cap, cap + 64, rows, s)
            ts += [us for nm, us in t.records if "per_sample" in nm]
        got = (oi.cpu().numpy().copy(), ow.cpu().numpy().copy())
        if ref is None:
            ref = got
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
        ts = np.array(ts[5:])
        print("top_steps %2d  u %-7s  median %.2f us  p10 %.2f  p90 %.2f" % (top, name, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
lib.per_sample_top_steps(11)
