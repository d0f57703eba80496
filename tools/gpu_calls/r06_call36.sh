#!/bin/bash
# round 6, call 36: update_priorities on the deferred-reduction launch; per_sample microbench (LDS levels x where the
# uniforms live); C3 bench + trace
set -u
O=gpurun_out/r06_call36
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_per_ride.py tests/test_per.py tests/test_dqn_full_size.py tests/test_dqn_agent.py tests/test_replay.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -30 | tee $O/pytest.txt
timeout 300 python tools/per_sample_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/per_sample_bench.txt
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
for v in fused64; do
  f="--fuse-conv-bwd-min-wg 64"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-prefill $f > $R/$O/prof_$v.log 2>&1)
  g=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c3_${v}_kernel_stats.csv
  head -20 $O/c3_${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
