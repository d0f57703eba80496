#!/bin/bash
# round 5, call 27: identical-history experiment, device against itself with EVERY initial weight moved by one ulp (8 seeds)
set -u
export TMPDIR=/tmp
timeout 400 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4,5,6,7 --dir profiles/r05_lc_forced --iterations 49 --epochs 10 --no-init --follow-hip-actions --perturb-all 2>&1 | grep -v amdgpu.ids | tail -4
mkdir -p gpurun_out/lc_forced_dev2
for s in 0 1 2 3 4 5 6 7; do mkdir -p gpurun_out/lc_forced_dev2/seed$s; cp profiles/r05_lc_forced/seed$s/hip_forced_ulpall.npz gpurun_out/lc_forced_dev2/seed$s/; done
