#!/bin/bash
# round 5, call 30: is the clipped surrogate the amplifier?  identical histories, device vs device + 1 ulp with the clipping range
# widened to 10 (it then almost never binds), 5 seeds
set -u
export TMPDIR=/tmp
timeout 70 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4 --dir profiles/r05_lc_forced --iterations 49 --epochs 10 --no-init --follow-hip-actions --clip-eps 10 2>&1 | grep -v amdgpu.ids | tail -2
timeout 70 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4 --dir profiles/r05_lc_forced --iterations 49 --epochs 10 --no-init --follow-hip-actions --clip-eps 10 --perturb-ulp 2>&1 | grep -v amdgpu.ids | tail -2
mkdir -p gpurun_out/lc_forced_dev4
for s in 0 1 2 3 4; do mkdir -p gpurun_out/lc_forced_dev4/seed$s; cp profiles/r05_lc_forced/seed$s/hip_forced*eps10.npz gpurun_out/lc_forced_dev4/seed$s/; done
