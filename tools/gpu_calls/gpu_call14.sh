#!/bin/bash
# round-2 evidence: full GPU suite, the five bench lines, kernel traces of C2 / C4 / C5
set -u
O=gpurun_out/r02_call14
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x --durations=5 2>&1 | tail -40 > $O/pytest.txt
tail -12 $O/pytest.txt
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do
  timeout 400 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
for w in c2 c1 c3 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d.get('roofline'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
R=$PWD
for w in c2 c4 c5; do
  extra="--workload $w"; [ $w = c2 ] && extra=""
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $R/bench.py $extra --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_$w.log 2>&1)
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  f=$(find /tmp/prof_$w -name '*domain_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${w}_domain_stats.csv
done
head -12 $O/c4_kernel_stats.csv | cut -c1-160
