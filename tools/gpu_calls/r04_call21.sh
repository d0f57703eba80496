#!/bin/bash
# round 4, call 21: direct convolution input gradient (windowed gather, register-staged kernel, its own launch) against
# dcol + col2im on the round-4 tree
set -u
O=gpurun_out/r04_call21
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 500 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1 coach_amd/librlx.so:1:192,192,-1:graph.DIRECT_CONV_INPUT_GRAD=1 coach_amd/librlx.so:1:192,192,-1:graph.DIRECT_CONV_INPUT_GRAD=always > $O/ab.txt 2>&1
tail -4 $O/ab.txt
