#!/bin/bash
# round 6, call 39: C3 with the FC layer's input gradient as 32 x 32 tiles with K over the waves (no K split over
# workgroups, no reduce launch): rlx_gemm_tuning kw_min_tiles 96 against the default 192; C2 the same (must not move)
set -u
O=gpurun_out/r06_call39
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
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
run c3_192 c3 ""
run c3_96 c3 "--kw-min-tiles 96"
run c3_48 c3 "--kw-min-tiles 48"
run c3_192b c3 ""
run c2_96 c2 "--kw-min-tiles 96"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-prefill --kw-min-tiles 96 > $R/$O/prof_c3.log 2>&1)
g=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c3_kw96_kernel_stats.csv
head -16 $O/c3_kw96_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
