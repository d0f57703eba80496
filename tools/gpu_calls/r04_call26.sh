#!/bin/bash
# round 4, call 26: rows per forward pass of Clipped PPO's whole-dataset passes (V(s), old policy): 256 / 512 / 1024 / 2048
set -u
O=gpurun_out/r04_call26
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for c in 256 512 1024 2048 256; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --ppo-chunk $c > $O/bench_$c.json 2> $O/bench_$c.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('chunk $c', d['value'], d['ms_per_step'])
except Exception as e:
    print('chunk $c ERR', e); print(open('$O/bench_$c.err').read()[-800:])
PY
done
