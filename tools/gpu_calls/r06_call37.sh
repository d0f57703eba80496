#!/bin/bash
# round 6, call 37: update_priorities as one more workgroup of the fused input-gradient launch (C3), fused backward at 64
# workgroups by default — full GPU suite, C3 bench + trace, C2 bench (unchanged path: check)
set -u
O=gpurun_out/r06_call37
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/pytest.txt
tail -15 $O/pytest.txt
run() { # name, workload, flags
  timeout 400 python bench.py --workload $2 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'update_us', r.get('update_us', r.get('update_us_in_epoch_graph')), 'calls', r.get('library_calls_per_update'), 'frac', r.get('frac'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c3 c3 ""
run c2 c2 ""
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-prefill > $R/$O/prof_c3.log 2>&1)
g=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c3_kernel_stats.csv
head -20 $O/c3_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
