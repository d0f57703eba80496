#!/bin/bash
# round 4, call 28: the full GPU suite and smoke() on the final tree
set -u
O=gpurun_out/r04_call28
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 -p no:cacheprovider 2>&1 | tail -30 > $O/pytest.txt
tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r['update_us_in_epoch_graph'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'])
PY
