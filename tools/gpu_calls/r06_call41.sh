#!/bin/bash
# round 6, call 41: two image pairs per workgroup in the conv2 / conv3 weight gradients (half the splits): parity, C2 A/B
set -u
O=gpurun_out/r06_call41
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_dw_f32.py tests/test_conv_dw_u8.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -12 | tee $O/pytest.txt
run() { # name, workload, flags
  timeout 400 python bench.py --workload $2 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'update_us', r.get('update_us', r.get('update_us_in_epoch_graph')), 'frac', r.get('frac'))
    if 'update_kernels' in r: print('     ', '  '.join('%s %.1f' % (k['kernel'][:28], k['avg_us']) for k in r['update_kernels']))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c2_p1 c2 "--conv-dw-pairs 1"
run c2_p2 c2 "--conv-dw-pairs 2"
run c2_p1b c2 "--conv-dw-pairs 1"
run c2_p2b c2 "--conv-dw-pairs 2"
run c3_p2 c3 "--conv-dw-pairs 2"
run c3_p1 c3 "--conv-dw-pairs 1"
