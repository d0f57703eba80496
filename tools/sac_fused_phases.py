"""Phase stamps of the fused SAC backward kernel (workgroup 0 = policy branch, slice 0; workgroup 2 kSplit = Q tower 0, slice 0).
    python tools/sac_fused_phases.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from tools.ac_fused_bench import Batch, make_agent

dev = torch.device("cuda:0")
lib = _rlx.lib()
ag, D, A, B = make_agent("sac", dev, True)
rng = np.random.RandomState(0)
b = Batch(dev, rng, B, D, A, 0.05)
ag.normals.copy_(torch.as_tensor(rng.standard_normal((3, B, A)), device=dev))
lib.fused_phase_stamps(1)
for it in range(6):
    ag._learn_device(b)
torch.cuda.synchronize()
f = ag._fused()
st = f.ws[-192:].view(torch.int64).cpu().numpy()
mhz = 100.0          # s_memtime / readcyclecounter ticks: printed raw too
for name, lo, hi in (("layer1, policy slice 0 (1 rows loaded, 2 padded copy, 3 layer slice)", 0, 4),
                     ("layer2, policy slice 0 (9 rows + head rows staged, 10 layer slice, 11 head partial)", 8, 12),
                     ("q_pi, tower 0 slice 0 (17 head sums + rows, 18 sample, 19 act_fc, 20 fc1 slice, 21 partial Q)", 16, 22),
                     ("q_grad, tower 0 slice 0 (25 rows + min Q, 26 dy slice, 27 hidden gradient share, 28 dQ/da share)", 24, 29),
                     ("backward, policy branch (35 operands staged, 36 head gradient, 37 dz2 slice, 38 partial dz1)", 32, 39),
                     ("backward, Q tower 0 (49 loss + dQ, 50 dz slice, 51 partial d fc1-input)", 48, 52)):
    v = st[lo:hi]
    print(name, "raw ticks:", [int(x - v[0]) if x else None for x in v])
lib.fused_phase_stamps(0)
