"""Where a forward-convolution-shaped product spends its workgroups' lives, by epilogue kind: the same M x N x K x batch
product with activation none / relu / tanh (bias on), event-timed over warm back-to-back launches, plus the median
prologue / main-loop / epilogue lengths from the per-workgroup phase stamps (rlx_gemm_debug_stamps).
    python tools/epilogue_probe.py [--pipeline 0|1]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx

lib = _rlx.lib()
if "--pipeline" in sys.argv:
    lib.gemm_pipeline(int(sys.argv[sys.argv.index("--pipeline") + 1]))
dev = torch.device("cuda:0")
SHAPES = [(25600, 64, 256, 1), (5184, 64, 512, 2), (3136, 64, 576, 2), (64, 512, 3136, 2)]
cap = 1 << 20
stamps = torch.zeros(cap, dtype=torch.int64, device=dev)
ws = torch.empty(1 << 24, dtype=torch.float32, device=dev)
print("%-26s %-5s | %8s | %6s %6s %6s | %7s" % ("M x N x K x batch", "act", "us/call", "prolog", "main", "epilog", "WG life"))
for M, N, K, T in SHAPES:
    A = torch.randn(T, M, K, device=dev) * 0.1
    B = torch.randn(T, K, N, device=dev) * 0.1
    bias = torch.randn(T, N, device=dev)
    C = torch.empty(T, M, N, device=dev)
    for act in (None, "relu", "tanh"):
        def run():
            _rlx.gemm(M, N, K, A, B, C, bias=bias, activation=act, batch=T, a_batch_stride=M * K, b_batch_stride=K * N,
                      c_batch_stride=M * N, bias_batch_stride=N, workspace=ws)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        stamps.zero_()
        lib.gemm_debug_stamps(stamps, cap)
        run()
        calls = np.zeros((8, 9), dtype=np.int64)
        n = ctypes.c_int(0)
        lib.gemm_debug_calls(calls.ctypes.data, 8, ctypes.addressof(n))
        lib.gemm_debug_stamps(None, 0)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()
        _, _, _, _, splits, gx, gy, gz, off = calls[0]
        nwg = gx * gy * gz
        s = st[off:off + 4 * nwg].reshape(nwg, 4).astype(np.float64) * 0.01
        s = s[(s > 0).all(axis=1)]
        med = np.median(np.diff(s, axis=1), axis=0)
        print("%-26s %-5s | %8.2f | %6.2f %6.2f %6.2f | %7.2f   (grid %dx%dx%d, splits %d)" % (
            "%d x %d x %d x %d" % (M, N, K, T), act or "none", us, med[0], med[1], med[2], np.median(s[:, 3] - s[:, 0]),
            gx, gy, gz, splits))
