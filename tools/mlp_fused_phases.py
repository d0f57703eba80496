"""Phase timeline of the fused DQN update kernel (s_memtime stamps of workgroup 0, csrc/mlp_fused.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coach_amd.nn.networks import DQNNet
dev = torch.device("cuda", 0)
net = DQNNet(dev, (4,), 2, replace_mse_with_huber_loss=False, seed=1)
B = 32
rng = np.random.RandomState(0)
t = lambda x, dt: torch.from_numpy(x).to(dev).to(dt)
args = (t(rng.randn(B, 4).astype(np.float32), torch.float32), t(rng.randn(B, 4).astype(np.float32), torch.float32), B,
        t(rng.randint(0, 2, B).astype(np.int32), torch.int32), t(rng.randn(B).astype(np.float32), torch.float32),
        t((rng.rand(B) < .1).astype(np.uint8), torch.uint8), 0.99)
td = torch.zeros(B, dtype=torch.float64, device=dev)
for _ in range(20):
    net.learn_from_batch(*args, td_errors=td)
torch.cuda.synchronize()
ws = net._fused["ws"]
G = 512 // 32
base = G * 3 * 32 * 2 + G * 32 * 256
off = base + ((G + 1) & ~1)
addr = ws.data_ptr() + off * 4
if addr & 4: off += 1
st = ws[off:off + 24].view(torch.int64).cpu().numpy()[:10]
names = ["inputs staged", "target tower", "online towers", "barrier 1", "loss/dQ", "wide-layer backward + Adam", "barrier 2", "first layer + Adam", "norm ticket / end"]
d = np.diff(st)
print("s_memtime ticks = shader cycles (about 2.1-2.4 GHz while the kernel runs)")
for n, x in zip(names, d):
    print("%-28s %7d cycles ~ %5.2f us at 2.3 GHz" % (n, x, x / 2300.0))
print("total %d cycles ~ %.1f us at 2.3 GHz" % (st[-1] - st[0], (st[-1] - st[0]) / 2300.0))
