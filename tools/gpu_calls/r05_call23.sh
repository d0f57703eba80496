#!/bin/bash
# round 5, call 23: identical-history experiment, device side on the final tree: every seed trained on the action history of
# its call-21 hip.npz, once as is (hip_forced.npz) and once from initial weights moved by one ulp (hip_forced_ulp.npz)
set -u
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" | tail -1
timeout 400 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4,5,6,7 --dir profiles/r05_lc_forced --iterations 49 --epochs 10 --no-init --follow-hip-actions 2>&1 | grep -v amdgpu.ids | tail -9
timeout 400 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4,5,6,7 --dir profiles/r05_lc_forced --iterations 49 --epochs 10 --no-init --follow-hip-actions --perturb-ulp 2>&1 | grep -v amdgpu.ids | tail -17
mkdir -p gpurun_out/lc_forced_dev
for s in 0 1 2 3 4 5 6 7; do mkdir -p gpurun_out/lc_forced_dev/seed$s; cp profiles/r05_lc_forced/seed$s/hip_forced*.npz gpurun_out/lc_forced_dev/seed$s/; done
