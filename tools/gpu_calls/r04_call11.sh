#!/bin/bash
# round 4, call 11: uint8 ring four slabs deep (rlx_gemm_pipeline 2) vs the default (uint8 register-staged), same build,
# same box; GEMM parity first
set -u
O=gpurun_out/r04_call11
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_gemm.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 400 python tools/ab_c2_pipeline.py 3 coach_amd/librlx.so:1 coach_amd/librlx.so:2 > $O/ab.txt 2>&1
tail -30 $O/ab.txt
