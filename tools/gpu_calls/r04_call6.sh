#!/bin/bash
# round 4, call 6: PER kernels in the C3 workload with / without touching the update's tree nodes early; baseline-vs-D2
# GEMM pipeline A/B once more (another box)
set -u
O=gpurun_out/r04_call6
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python tools/per_insitu.py 48 2>&1 | grep -v amdgpu.ids | tee $O/per_insitu.txt
timeout 600 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:0 coach_amd/ab/librlx_d2w3.so:1 2>&1 | grep -v amdgpu.ids | tee $O/ab_pipeline.txt
