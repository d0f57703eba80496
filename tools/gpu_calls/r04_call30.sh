#!/bin/bash
# round 4, call 30: rlx_gemm_desc.row_heads across shapes / pass counts / activations against the separate launches
set -u
O=gpurun_out/r04_call30
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q -k row_heads > $O/tests.txt 2>&1
tail -15 $O/tests.txt
