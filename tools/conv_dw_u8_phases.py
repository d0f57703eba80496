"""Phase stamps of conv_dw_u8_kernel (workgroup 0) at the C2 minibatch shape, cold (operands just written) and warm.
    python tools/conv_dw_u8_phases.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx

dev = torch.device("cuda:0")
lib = _rlx.lib()
B, H, W, C, KH, KW, S, Co, T = 64, 84, 84, 4, 8, 8, 4, 32, 2
P, K = 400, 256
rng = np.random.RandomState(0)
frames = torch.from_numpy(rng.randint(0, 256, size=(B, H, W, C)).astype(np.uint8)).to(dev)
dz = torch.from_numpy(rng.randn(T, B * P, Co).astype(np.float32)).to(dev)
need = ctypes.c_longlong()
lib.conv_dw_u8_workspace_floats(B, H, W, C, KH, KW, S, Co, T, ctypes.byref(need))
ws = torch.zeros(need.value, dtype=torch.float32, device=dev)
dw = torch.zeros(T, K, Co, dtype=torch.float32, device=dev)
db = torch.zeros(T, Co, dtype=torch.float32, device=dev)
st = torch.zeros(8, dtype=torch.int64, device=dev)
job = _rlx.SplitkJob()
s_ = _rlx.current_stream()
lib.conv_dw_u8_stamps(st)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for mode in ("cold (1 GB written in between)", "warm"):
    for it in range(3):
        if mode.startswith("cold"):
            big.fill_(1.0)
            dz.mul_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.conv_dw_u8(frames, 255.0, dz, B * P * Co, B, H, W, C, KH, KW, S, Co, T, dw, K * Co, db, Co, ws, need.value,
                       ctypes.byref(job), s_)
        e1.record()
        torch.cuda.synchronize()
    v = st.cpu().numpy()
    print("%-32s event %.1f us; ticks: requests issued + table %d, frame rows converted + dz landed %d, products %d, stores + column sums %d"
          % (mode, 1e3 * e0.elapsed_time(e1), v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3]))
lib.conv_dw_u8_stamps(None)
