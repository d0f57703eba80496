#!/bin/bash
# round 4, call 32: conv1 forward on 32 x 64 tiles with K split over the waves (800 workgroups: three per CU) — kw_below_tiles 401
set -u
O=gpurun_out/r04_call32
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 400 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1:192,192,-1 coach_amd/librlx.so:1:401,192,-1 > $O/ab.txt 2>&1
tail -3 $O/ab.txt
