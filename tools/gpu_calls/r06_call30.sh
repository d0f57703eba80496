#!/bin/bash
# round 6, call 30: four wave groups in every fused forward launch as the default (conv3 on all eight waves) + the round's other
# defaults (two-pass conv dW, 16-row tails, row-local heads): full GPU suite, A/B of the C2 line
set -u
O=gpurun_out/r06_call30
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -25 > $O/pytest.txt; tail -25 $O/pytest.txt
run() { # name, flags
  timeout 500 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r.get('update_us_in_epoch_graph'), 'conv', d['box'].get('fused_conv_forward_in_update_us'), 'launches', r.get('kernel_launches_per_update'))
    print('     ', '  '.join('%s %.1f' % (k['kernel'][:28], k['avg_us']) for k in r['update_kernels']))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run g4 "--steps 10 --warmup 3"
run rule "--steps 10 --warmup 3 --conv-fwd-groups 0"
run g4b "--steps 10 --warmup 3"
run ruleb "--steps 10 --warmup 3 --conv-fwd-groups 0"
