#!/bin/bash
# round 4, call 8: full suite (one record per update, SAC branches, DMA default); C5 with / without branches; C4
set -u
O=gpurun_out/r04_call8
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 -p no:cacheprovider > $O/pytest_full.txt 2>&1
grep -n -i "fault\|fatal\|Aborted\|passed\|failed\|error" $O/pytest_full.txt | head -20
for cfg in "c5 --sac-branches 1" "c5 --sac-branches 0" "c5 --sac-branches 1" "c5 --sac-branches 0" "c4"; do
  tag=$(echo $cfg | tr ' -' '__')
  timeout 400 python bench.py --workload $cfg --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('$cfg', d['value'], d['ms_per_step'], 'upd/s', d['grad_updates_per_s'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'))
except Exception as e:
    print('$cfg ERR', e); print(open('$O/bench_$tag.err').read()[-1500:])
PY
done
