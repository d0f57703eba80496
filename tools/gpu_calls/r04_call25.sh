#!/bin/bash
# round 4, call 25: one default bench line (box lottery: a slow-class box gives the final tree's slow-box line)
set -u
O=gpurun_out/r04_call25
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r['update_us_in_epoch_graph'], 'conv1', d['box'].get('conv1_forward_in_update_us'), 'MHz', d['box'].get('shader_MHz_under_mfma'))
PY
