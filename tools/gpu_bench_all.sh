#!/bin/bash
# the five bench lines + the C2 kernel trace of the same box
set -u
O=gpurun_out/${1:-r02_bench_all}
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do timeout 400 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in c2 c1 c3 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('traffic'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
f=$(find /tmp/prof_c2 -name '*domain_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_domain_stats.csv
grep "^{" $O/bench_c2.err | cut -c1-200
