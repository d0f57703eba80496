"""Phase stamps of conv_dw_f32_kernel (workgroup 0) at the C2 minibatch shapes.
    python tools/conv_dw_f32_phases.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx

dev = torch.device("cuda:0")
lib = _rlx.lib()
B, T, Co = 64, 2, 64
for name, (H, W, C, k, S) in (("conv3", (9, 9, 64, 3, 1)), ("conv2", (20, 20, 32, 4, 2))):
    OH = (H - k) // S + 1
    P, K = OH * OH, k * k * C
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(T, B, H, W, C).astype(np.float32)).to(dev)
    dz = torch.from_numpy(rng.randn(T, B * P, Co).astype(np.float32)).to(dev)
    need = ctypes.c_longlong()
    lib.conv_dw_f32_workspace_floats(B, H, W, C, k, k, S, Co, T, ctypes.byref(need))
    ws = torch.zeros(need.value, dtype=torch.float32, device=dev)
    dw = torch.zeros(T, K, Co, dtype=torch.float32, device=dev)
    db = torch.zeros(T, Co, dtype=torch.float32, device=dev)
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    job = _rlx.SplitkJob()
    s_ = _rlx.current_stream()
    lib.conv_dw_f32_stamps(st)
    for it in range(3):
        dz.mul_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.conv_dw_f32(x, B * H * W * C, dz, B * P * Co, B, H, W, C, k, k, S, Co, T, dw, K * Co, db, Co, ws, need.value,
                        ctypes.byref(job), s_)
        e1.record()
        torch.cuda.synchronize()
    v = st.cpu().numpy()
    print("%s: event %.1f us; ticks: operands in LDS %d, products %d, stores + column sums %d"
          % (name, 1e3 * e0.elapsed_time(e1), v[1] - v[0], v[2] - v[1], v[3] - v[2]))
    lib.conv_dw_f32_stamps(None)
