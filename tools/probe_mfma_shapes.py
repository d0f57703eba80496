"""Issue rate vs dependent latency of the two fp32 MFMA shapes (rlx_probe_mfma_shape): cycles per MFMA with 1, 2 and 4
independent accumulator chains per wave, one wave per SIMD (256-thread workgroups, one per CU) and two per SIMD.
    python tools/probe_mfma_shapes.py            -> table + one JSON line (bench.py's box.mfma_cycles uses the same call)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx


def measure(dev, wgs=256, iters=16384):
    lib = _rlx.lib()
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    res = {}
    for small in (0, 1):
        for chains in (1, 2, 4):
            out = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
            for _ in range(3):
                lib.probe_mfma_shape(wgs, iters, small, chains, out, sink, _rlx.current_stream())
            torch.cuda.synchronize()
            o = out.cpu().numpy().reshape(wgs, 2).astype(np.float64)
            res[("16x16x4" if small else "32x32x2", chains)] = (float(np.median(o[:, 0])) / (iters * chains),
                                                                float(np.median(o[:, 0] / o[:, 1])) * 100.0)
    return res


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    line = {}
    for wgs in (256, 512):
        r = measure(dev, wgs)
        for (shape, chains), (cyc, mhz) in sorted(r.items()):
            flops = 4096 if shape == "32x32x2" else 2048
            print("%4d workgroups (%d wave(s) per SIMD)  v_mfma_f32_%s  %d chain(s): %6.1f cycles per MFMA = %5.1f flop/clk/SIMD"
                  "%s   (shader %4.0f MHz)" % (wgs, wgs // 256, shape, chains, cyc, flops / cyc,
                                                "" if wgs == 256 else " per wave", mhz))
            line["%s_c%d_w%d" % (shape, chains, wgs // 256)] = round(cyc, 1)
    print(json.dumps({"mfma_cycles": line}))
