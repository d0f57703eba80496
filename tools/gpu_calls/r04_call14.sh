#!/bin/bash
# round 4, call 14: Adam + norm + finish in one launch (fence-free publication of the per-block sums of squares) —
# bit-identity against the two-launch form, the agents that use it, then the C2 update with the knob on / off
set -u
O=gpurun_out/r04_call14
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_fused_steps.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 400 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1:192,192,-1:adam_norm_in_kernel=0 coach_amd/librlx.so:1:192,192,-1:adam_norm_in_kernel=1 coach_amd/librlx.so:1:192,192,-1:adam_norm_in_kernel=2 > $O/ab.txt 2>&1
tail -30 $O/ab.txt
