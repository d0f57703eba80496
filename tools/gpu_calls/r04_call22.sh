#!/bin/bash
# round 4, call 22: run-to-run spread of the default bench line on one box (with and without the roofline / cpu legs)
set -u
O=gpurun_out/r04_call22
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for i in 1 2; do
timeout 300 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err
done
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-roofline > $O/bench_plain_30.json 2> $O/bench_plain_30.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('roofline',{}).get('update_us_in_epoch_graph'), d.get('phase_ms'))
PY
