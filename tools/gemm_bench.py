"""GEMM microbenchmark through the C ABI: python tools/gemm_bench.py M N K [batch] [reps]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coach_amd import _rlx

def bench(M, N, K, batch=1, reps=20, nt=False, tn=False):
    dev = torch.device("cuda:0")
    A = torch.randn(batch, M, K, device=dev) if not tn else torch.randn(batch, K, M, device=dev)
    B = torch.randn(batch, K, N, device=dev) if not nt else torch.randn(batch, N, K, device=dev)
    C = torch.empty(batch, M, N, device=dev)
    ws = torch.empty(1 << 26, device=dev)
    kw = dict(batch=batch, a_batch_stride=M * K, b_batch_stride=K * N, c_batch_stride=M * N, workspace=ws)
    if tn: kw["a_strides"] = (1, M)
    if nt: kw["b_strides"] = (1, K)
    for _ in range(3):
        _rlx.gemm(M, N, K, A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _rlx.gemm(M, N, K, A, B, C, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("M=%d N=%d K=%d b=%d %s%s: %.1f us  %.1f TFLOP/s" % (M, N, K, batch, "TN " if tn else "", "NT" if nt else "",
                                                             us, 2.0 * M * N * K * batch / us / 1e6))

if __name__ == "__main__":
    if len(sys.argv) > 3:
        a = [int(x) for x in sys.argv[1:]]
        bench(*a)
    else:
        for shp in [(4096, 4096, 4096), (8192, 8192, 512), (2048, 2048, 8192), (8192, 512, 64), (8192, 512, 256),
                    (8192, 512, 1024), (16384, 64, 512), (16384, 64, 2048)]:
            bench(*shp)
        bench(4096, 4096, 4096, nt=True)
        bench(4096, 4096, 4096, tn=True)
