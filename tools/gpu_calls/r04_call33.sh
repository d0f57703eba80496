#!/bin/bash
# round 4, call 33: the full GPU suite once more on another box (flake check of the final tree)
set -u
O=gpurun_out/r04_call33
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -12 > $O/pytest.txt
tail -5 $O/pytest.txt
timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-260
