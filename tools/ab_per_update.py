"""Same-process A/B of the two prioritized-replay priority-update kernels (rlx_per_tuning): a 2^20-leaf tree with
every leaf populated, `reps` updates of n random leaves each, timed with HIP events around the whole train of launches
on one stream (launch gaps included — that is how the DQN loop issues them: one update per training step).

    python tools/ab_per_update.py [--reps 2000] > profiles/r03_ab_per_update.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coach_amd import _rlx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2000)
    ap.add_argument("--capacity", type=int, default=1 << 20)
    a = ap.parse_args()
    lib, dev = _rlx.lib(), torch.device("cuda:0")
    cap = a.capacity
    trees = [torch.empty(2 * cap - 1, dtype=torch.float64, device=dev) for _ in range(3)]
    maxp = torch.zeros(1, dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    lib.per_init(*trees, cap, maxp, 0)
    for start in range(0, cap, 1 << 16):
        lib.per_store(*trees, cap, start, 1 << 16, 0.6, maxp, status, 0)
    rng = np.random.RandomState(0)
    print("capacity %d, %d updates per line, device %s" % (cap, a.reps, torch.cuda.get_device_name(0)))
    print("%6s  %22s  %22s" % ("n", "level-synchronous us", "LDS path walk us"))
    for n in (32, 64, 128, 256):
        idx = torch.from_numpy(rng.randint(0, cap, (64, n)).astype(np.int32)).to(dev)
        err = torch.from_numpy(rng.rand(64, n) * 3.0).to(dev)
        line = []
        for path_max in (0, 256, 0, 256):
            lib.per_tuning(path_max)
            s = torch.cuda.current_stream().cuda_stream
            for r in range(50):
                lib.per_update(*trees, cap, idx[r % 64], err[r % 64], n, 0.6, 1e-6, maxp, status, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(a.reps):
                lib.per_update(*trees, cap, idx[r % 64], err[r % 64], n, 0.6, 1e-6, maxp, status, s)
            e1.record()
            torch.cuda.synchronize()
            line.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        lib.per_tuning(256)
        print("%6d  %10.2f %10.2f   %10.2f %10.2f" % (n, line[0], line[2], line[1], line[3]))
    assert int(status.item()) == 0


if __name__ == "__main__":
    main()
