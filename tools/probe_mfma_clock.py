"""Shader clock and cycles per dependent fp32 MFMA under different loads (rlx_probe_mfma).
    python tools/probe_mfma_clock.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx

dev = torch.device("cuda:0")
lib = _rlx.lib()
sink = torch.zeros(1, dtype=torch.float32, device=dev)
for wgs, iters, reps in ((1, 4096, 3), (256, 4096, 3), (768, 4096, 3), (768, 65536, 3), (256, 262144, 2)):
    out = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
    for _ in range(reps):
        lib.probe_mfma(wgs, iters, out, sink, _rlx.current_stream())
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(wgs, 2).astype(np.float64)
    cyc, ticks = o[:, 0], o[:, 1]
    print("%4d workgroups x %6d MFMAs: %.1f shader cycles per MFMA (median), chain %.1f us, shader clock %.0f MHz "
          "(min %.0f, max %.0f)" % (wgs, iters, np.median(cyc) / iters, np.median(ticks) / 100.0,
                                    np.median(cyc / ticks * 100.0), np.min(cyc / ticks * 100.0), np.max(cyc / ticks * 100.0)))

# ---- do a wave's back-to-back 16-byte-per-lane requests overlap?  (global -> LDS requests vs loads into registers)
src = torch.ones((12 << 20) // 4 + 256 * 1024, dtype=torch.float32, device=dev)
for wgs, waves in ((1, 0), (1, 1), (256, 0), (256, 1)):
    for dma in (1, 0):
        row = []
        for k in (1, 2, 4, 8):
            out = torch.zeros(wgs, dtype=torch.int64, device=dev)
            best = None
            for r in range(6):
                lib.probe_requests(wgs, k, dma, r | (waves << 16), src, out, sink, _rlx.current_stream())
                torch.cuda.synchronize()
                c = float(np.median(out.cpu().numpy()))
                best = c if best is None or (r >= 1 and c < best) else best
            row.append("%d: %5.0f" % (k, best))
        print("%3d workgroups, %s issuing, %-22s cycles until K requests per wave have landed   K = %s" % (
            wgs, "1 wave " if waves else "4 waves", "global -> LDS requests" if dma else "loads into registers", "   ".join(row)))
