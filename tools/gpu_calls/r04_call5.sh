#!/bin/bash
# round 4, call 5: the full suite after the no-GC-inside-capture fix; same-box A/B of GEMM pipeline builds
# (ring depth 2 at 3 / 4 workgroups per CU) and of the tile thresholds (conv3 forward / FC dX on 32 x 32 tiles).
set -u
O=gpurun_out/r04_call5
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 -p no:cacheprovider > $O/pytest_full.txt 2>&1
grep -n -i "fault\|fatal\|Aborted\|passed\|failed" $O/pytest_full.txt | head -10
timeout 900 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:0 coach_amd/librlx.so:0:192,200,-1 coach_amd/librlx.so:1 \
    coach_amd/ab/librlx_d2w3.so:1 coach_amd/ab/librlx_d2w4.so:1 coach_amd/ab/librlx_d2w4.so:1:192,200,-1 2>&1 | grep -v amdgpu.ids | tee $O/ab_pipeline.txt
