#!/bin/bash
# final bench lines of the round for the workloads the 16x16 thin tiles and the epilogue changes touch (C4, C5, C2) + the C2 kernel trace
set -u
O=gpurun_out/r02_final2
mkdir -p $O
export TMPDIR=/tmp
timeout 80 python bench.py --workload c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 80 python bench.py --workload c5 --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 70 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
R=$PWD
(cd /tmp && timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
for w in c4 c5 c2; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('update_us'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
