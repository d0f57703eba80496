#!/bin/bash
# round 6, call 32: fused input-gradient chain — requests in need order (dz3 first), gathers with all taps in flight
set -u
O=gpurun_out/r06_call32
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_bwd_fused.py tests/test_ppo_full_size.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee $O/pytest.txt
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
run a "--steps 10 --warmup 3"
run b "--steps 10 --warmup 3"
run tail0 "--steps 10 --warmup 3 --conv32-tail16 0"
python tools/conv32_timeline.py --tail16 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/timeline.txt
