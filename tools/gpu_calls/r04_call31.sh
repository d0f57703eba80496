#!/bin/bash
# round 4, call 31: conv1 forward on 128 x 64 tiles (two accumulators per wave, 200 workgroups) vs 64 x 64 (400)
set -u
O=gpurun_out/r04_call31
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 400 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1:192,192,-1:gemm_tall_tiles=0 coach_amd/librlx.so:1:192,192,-1:gemm_tall_tiles=1 > $O/ab.txt 2>&1
tail -3 $O/ab.txt
