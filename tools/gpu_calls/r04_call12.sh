#!/bin/bash
# round 4, call 12: per-workgroup phase stamps of the uint8 products (conv1 forward, conv1 dW) with the register-staged
# loop (mode 1) and the uint8 ring (mode 2) inside the captured update
set -u
O=gpurun_out/r04_call12
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for p in 1 2; do
timeout 200 python tools/gemm_timeline.py --pipeline $p > $O/timeline_$p.txt 2>&1
echo "## pipeline $p"; tail -9 $O/timeline_$p.txt
done
