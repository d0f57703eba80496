#!/bin/bash
# round 5, call 25 (seeds 16-31): the DEVICE against itself as a free-running ensemble (the statistic of the K = 16 device-vs-oracle
# experiment): 16 seeds as they are (hip.npz) and from initial weights moved by one ulp (hip_ulp.npz), 10 epochs, 49 iterations
set -u
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" | tail -1
S=16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31
timeout 500 python tools/loss_curve_c2.py --side hip --hip-seeds $S --dir gpurun_out/lc_self --iterations 49 --epochs 10 --no-init 2>&1 | grep -v amdgpu.ids | tail -2
timeout 500 python tools/loss_curve_c2.py --side hip --hip-seeds $S --dir gpurun_out/lc_self --iterations 49 --epochs 10 --no-init --perturb-ulp 2>&1 | grep -v amdgpu.ids | tail -2
