#!/bin/bash
# round 6, call 45: K consecutive TD3 / SAC updates as one captured graph over one staged record: parity, suite, C4 / C5 A/B
set -u
O=gpurun_out/r06_call45
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_draws_ahead.py tests/test_agent_loops.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -25 | tee $O/pytest_new.txt
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
run c4_k8 c4 ""
run c4_k1 c4 "--update-chunk 1"
run c4_k16 c4 "--update-chunk 16"
run c5_k8 c5 ""
run c5_k1 c5 "--update-chunk 1"
run c5_k16 c5 "--update-chunk 16"
