#!/bin/bash
# round 6, call 35: per_sample with the top levels in LDS + three levels per round trip, the uniform draws read from pinned
# host memory (no blit), the batch's small columns on the frame gather's launch — full GPU suite, C3 bench + trace
set -u
O=gpurun_out/r06_call35
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest.txt
tail -12 $O/pytest.txt
run() { # name, flags
  timeout 400 python bench.py --workload c3 --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run fused64 "--fuse-conv-bwd-min-wg 64"
run default ""
for v in fused64; do
  f="--fuse-conv-bwd-min-wg 64"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-prefill $f > $R/$O/prof_$v.log 2>&1)
  g=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c3_${v}_kernel_stats.csv
  head -24 $O/c3_${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
