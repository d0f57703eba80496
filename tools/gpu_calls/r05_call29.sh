#!/bin/bash
# round 5, call 29 (seeds 32-43): the DEVICE against itself as a free-running ensemble (the statistic of the K = 16 device-vs-oracle
# experiment): 16 seeds as they are (hip.npz) and from initial weights moved by one ulp (hip_ulp.npz), 10 epochs, 49 iterations
set -u
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" | tail -1
S=32,33,34,35,36,37,38,39,40,41,42,43
timeout 500 python tools/loss_curve_c2.py --side hip --hip-seeds $S --dir gpurun_out/lc_self --iterations 49 --epochs 10 --no-init 2>&1 | grep -v amdgpu.ids | tail -2
timeout 500 python tools/loss_curve_c2.py --side hip --hip-seeds $S --dir gpurun_out/lc_self --iterations 49 --epochs 10 --no-init --perturb-ulp 2>&1 | grep -v amdgpu.ids | tail -2
