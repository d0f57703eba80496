#!/bin/bash
# round 6, call 38: rlx_dqn_head_loss_backward (head loss + head backward as one launch), glibc pow tables in LDS inside
# rlx_per_sample (8 levels from LDS by default) — full GPU suite, per_sample microbench, C3 bench + trace
set -u
O=gpurun_out/r06_call38
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/pytest.txt
tail -15 $O/pytest.txt
timeout 300 python tools/per_sample_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/per_sample_bench.txt
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
run c1 c1 ""
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-prefill > $R/$O/prof_c3.log 2>&1)
g=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c3_kernel_stats.csv
head -18 $O/c3_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
