#!/bin/bash
# round 4, call 3: which pipeline the suite crash (test_reference_loop, DQN update capture) belongs to + phase stamps
# of the C2 GEMMs under both pipelines (tools/gemm_timeline.py).
set -u
O=gpurun_out/r04_call3
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for pl in 0 1; do
timeout 600 python - > $O/reference_loop_pl$pl.txt 2>&1 <<PY
import sys, pytest
sys.path.insert(0, '.')
from coach_amd import _rlx
_rlx.lib().gemm_pipeline($pl)
sys.exit(pytest.main(["tests/test_reference_loop.py", "-m", "gpu", "-q", "--tb=short", "-x", "-p", "no:cacheprovider"]))
PY
echo "== pipeline $pl"; grep -v "^  File\|^Extension\|pluggy\|_pytest" $O/reference_loop_pl$pl.txt | head -30
done
for pl in 1 0; do
timeout 300 python tools/gemm_timeline.py --pipeline $pl > $O/timeline_pl$pl.txt 2>&1; tail -16 $O/timeline_pl$pl.txt
done
