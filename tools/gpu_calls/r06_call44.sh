#!/bin/bash
# round 6, call 44: the host-RNG side of Agent.train on a producer thread (TD3 / SAC): parity, full suite, C5 / C4 A/B
set -u
O=gpurun_out/r06_call44
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_draws_ahead.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -12 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest.txt
tail -8 $O/pytest.txt
run() { # name, workload, flags
  timeout 400 python bench.py --workload $2 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'host draws us', r.get('host_draws_us_per_update'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c5_on c5 "--host-draws-ahead 1"
run c5_off c5 "--host-draws-ahead 0"
run c4_on c4 "--host-draws-ahead 1"
run c4_off c4 "--host-draws-ahead 0"
run c5_on2 c5 ""
