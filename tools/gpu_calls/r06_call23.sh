#!/bin/bash
# round 6, call 23: A/Bs of the C2 line — conv3 on all eight waves of the fused forward launch (four wave groups), FC forward
# with fewer K chunks (split cap 17 / 20)
set -u
O=gpurun_out/r06_call23
mkdir -p $O
export TMPDIR=/tmp
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
run base "--steps 10 --warmup 3"
run groups4 "--steps 10 --warmup 3 --conv-fwd-groups 4"
run cap17 "--steps 10 --warmup 3 --split-cap 17"
run cap20 "--steps 10 --warmup 3 --split-cap 20"
run base2 "--steps 10 --warmup 3"
timeout 300 python -m pytest tests/test_ppo_full_size.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
