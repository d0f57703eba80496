#!/bin/bash
# round 4, call 4: epilogue probe (activation kinds x pipelines) + the full suite with its whole output kept
set -u
O=gpurun_out/r04_call4
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for pl in 0 1; do echo "== pipeline $pl"; timeout 200 python tools/epilogue_probe.py --pipeline $pl 2>&1 | grep -v amdgpu.ids | tee $O/probe_pl$pl.txt; done
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 -p no:cacheprovider > $O/pytest_full.txt 2>&1
grep -n -i "fault\|fatal\|hsa_\|hip error\|Aborted\|passed\|failed" $O/pytest_full.txt | head -20
head -c 3000 $O/pytest_full.txt | grep -v "^  File" | head -40
