"""Times the narrow-dense head kernels at the C2 head shapes (A/B helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coach_amd import _rlx
lib = _rlx.lib(); dev = torch.device("cuda:0")
def t(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): fn()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps // 50): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (M, K, N) in [(64, 512, 6), (64, 512, 1), (256, 256, 16), (32, 512, 4)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev)
    dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); dx = torch.empty(M, K, device=dev)
    s = lambda: _rlx.current_stream()
    f = t(lambda: lib.dense_small_forward(x, 0, w, 0, b, 0, y, 0, 1, M, K, N, 0, s()))
    bw = t(lambda: lib.dense_small_backward(x, 0, w, 0, dy, 0, None, 0, dw, 0, db, 0, dx, 0, 1, M, K, N, 0, 2, s()))
    print("M=%d K=%d N=%d  fwd %.2f us  bwd %.2f us" % (M, K, N, f, bw))
