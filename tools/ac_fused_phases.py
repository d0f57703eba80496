"""Phase stamps of the fused TD3 chain kernels (workgroup 0): where do their microseconds go?
    python tools/ac_fused_phases.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from coach_amd import _rlx
from tools.ac_fused_bench import Batch, make_agent

dev = torch.device("cuda:0")
lib = _rlx.lib()
ag, D, A, B = make_agent("td3", dev, True)
rng = np.random.RandomState(0)
b = Batch(dev, rng, B, D, A)
ag.noise.copy_(torch.as_tensor(rng.normal(0, 0.2, (B, A)), device=dev))
lib.fused_phase_stamps(1)
for it in range(6):
    ag._critic_device(b)
    if it % 2:
        ag._actor_device(b)
torch.cuda.synchronize()
f = ag._fused()
st = f.ws[-192:].view(torch.int64).cpu().numpy()
mhz = 2400.0
for name, lo, hi in (("forward1 (workgroup 0: mu_target, slice 0)", 0, 4), ("forward2", 16, 21), ("critic_backward", 32, 36),
                     ("actor_forward", 48, 50), ("actor_q", 56, 58), ("actor_backward", 64, 66),
                     ("dW + Adam of the critic (workgroup 0: 1 start, 2 products, 3 norm published, 4 ticket, 5 Adam stores issued, 6 barrier)", 80, 86)):
    v = st[lo:hi]
    if v[0] == 0:
        continue
    print(name)
    for i in range(1, len(v)):
        if v[i]:
            print("   phase %2d: %7.2f us" % (i, (v[i] - v[i - 1]) / mhz))
    print("   total   : %7.2f us" % ((v[v > 0][-1] - v[0]) / mhz))
    if lo == 80:
        w = st[80:92]
        print("   inside phase 1: start -> staging requests begin %.2f us, -> staged in LDS %.2f us, -> barrier passed %.2f us, -> products done %.2f us"
              % ((w[7] - w[0]) / mhz, (w[8] - w[7]) / mhz, (w[9] - w[8]) / mhz, (w[1] - w[9]) / mhz))
lib.fused_phase_stamps(0)
