#!/bin/bash
# round 6, call 40: rlx_gemm_desc.kw_min_tiles (the dense layers' input gradients at 96) — full GPU suite, C3, C2, C1
set -u
O=gpurun_out/r06_call40
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/pytest.txt
tail -12 $O/pytest.txt
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
run c3_kw0 c3 "--kw-min-tiles 0"
run c2 c2 ""
run c2_kw0 c2 "--kw-min-tiles 0"
run c1 c1 ""
run c4 c4 ""
run c5 c5 ""
