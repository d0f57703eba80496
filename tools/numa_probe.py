"""Does the pool's "box class" (the same code: C2 update 195 us on some boxes, 258 us on others, identical calibration
numbers) come from WHERE THE PROCESS RUNS relative to the GPU?  Prints the GPU's NUMA node, the CPU lists, this process's
affinity, and times eager Clipped-PPO minibatch updates (sum of the library's kernel durations + wall time per update)
under the default affinity, pinned to the GPU's node, and pinned to another node.
Usage: python tools/numa_probe.py"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return "<%s>" % e.__class__.__name__


print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
cpulists = {}
for n in nodes:
    cpulists[int(n.rsplit("node", 1)[1])] = read(n + "/cpulist")
    print(n, "cpulist", cpulists[int(n.rsplit("node", 1)[1])], "meminfo", read(n + "/meminfo").split("\n")[0][:60])
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    if read(d + "/vendor") == "0x1002":
        print(d, "numa_node", read(d + "/numa_node"), "local_cpulist", read(d + "/local_cpulist"))
for k in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
    t = read(k)
    if "simd_count" in t and "simd_count 0" not in t:
        print(k, [l for l in t.split("\n") if l.split()[0] in ("simd_count", "numa_node", "domain", "location_id", "unique_id")][:5])
print("THP:", read("/sys/kernel/mm/transparent_hugepage/enabled"), "| numa_balancing:", read("/proc/sys/kernel/numa_balancing"))

import numpy as np
import torch

from coach_amd import _rlx
from coach_amd.nn.networks import ClippedPPONet


def parse(cl):
    out = set()
    for part in cl.split(","):
        if "-" in part:
            a, b = part.split("-")
            out |= set(range(int(a), int(b) + 1))
        elif part.strip().isdigit():
            out.add(int(part))
    return out


dev = torch.device("cuda:0")
B, A, shape = 64, 6, (84, 84, 4)
np.random.seed(0)
net = ClippedPPONet(dev, shape, A, seed=0)
rng = np.random.RandomState(0)
obs = torch.from_numpy(rng.randint(0, 256, size=(B,) + shape).astype(np.uint8)).to(dev)
actions = torch.from_numpy(rng.randint(0, A, size=B).astype(np.int32)).to(dev)
adv = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
vt = torch.from_numpy(rng.randn(B).astype(np.float32)).to(dev)
net.update_target(1.0)
old = net.policy_probs(obs, B, use_target=True, tag="old")


def measure(label):
    for _ in range(5):
        net.train_minibatch(obs, B, actions, adv, vt, old)
    torch.cuda.synchronize()
    with _rlx.KernelTimer(4096) as timer:
        for _ in range(20):
            net.train_minibatch(obs, B, actions, adv, vt, old)
    per = {}
    for n, us in timer.records:
        per.setdefault(n.split("<")[0], []).append(us)
    ksum = sum(sum(v) for v in per.values()) / 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        net.train_minibatch(obs, B, actions, adv, vt, old)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    fused = np.mean(per.get("conv23_forward_kernel", [0]))
    small = np.mean(per.get("splitk_reduce_rows_kernel", [0]))
    print("%-34s kernels %.1f us / update, eager wall %.1f us / update; fused forward %.1f us, reduce_rows %.1f us"
          % (label, ksum, wall, fused, small))


measure("default affinity")
gpu_node = None
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    if read(d + "/vendor") == "0x1002":
        v = read(d + "/numa_node")
        if v.lstrip("-").isdigit():
            gpu_node = int(v)
            break
all_cpus = os.sched_getaffinity(0)
for node, cl in cpulists.items():
    cpus = parse(cl) & all_cpus
    if not cpus:
        continue
    os.sched_setaffinity(0, cpus)
    time.sleep(0.05)
    measure("pinned to node %d%s" % (node, " (the GPU's)" if node == gpu_node else ""))
os.sched_setaffinity(0, all_cpus)
measure("default affinity again")
