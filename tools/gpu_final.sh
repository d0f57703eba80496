#!/bin/bash
# round-2 final evidence: full GPU suite, smoke, the five bench lines, C2 kernel trace + PMC passes
set -u
O=gpurun_out/r02_final
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 2>&1 | tail -30 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do timeout 400 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in c2 c1 c3 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
f=$(find /tmp/prof_c2 -name '*domain_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_domain_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/ppo_update_once.py > $R/$O/pmc_$c.log 2>&1)
f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv
done
ls -la $O | head -30
head -8 $O/c2_kernel_stats.csv | cut -c1-160
