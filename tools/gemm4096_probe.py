"""The 4096^3 fp32 calibration product of bench.py's `box` block on its own (5 launches), as a target for rocprofv3 --pmc
passes: which wave state the large-tile kernel (gemm_dma_big_kernel, 64 x 64 per wave) spends its cycles in.
Usage: rocprofv3 --kernel-trace --pmc <SQ counters> --output-format csv -d out -- python tools/gemm4096_probe.py [--big 0|1|2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from coach_amd import _rlx

lib = _rlx.lib()
if "--big" in sys.argv:
    lib.gemm_big_tiles(int(sys.argv[sys.argv.index("--big") + 1]))
dev = torch.device("cuda:0")
M = 4096
A = torch.randn(M, M, device=dev)
B = torch.randn(M, M, device=dev)
C = torch.empty(M, M, device=dev)
ws = torch.empty(1 << 22, device=dev)
for _ in range(2):
    _rlx.gemm(M, M, M, A, B, C, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    _rlx.gemm(M, M, M, A, B, C, workspace=ws)
e1.record()
e1.synchronize()
print("TFLOP/s", 2.0 * M ** 3 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
