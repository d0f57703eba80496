#!/bin/bash
# round 6, call 19: rlx_conv_dw_multi (the three convolution weight gradients as one launch): bit identity, A/B of the C2 line
# A/B of the C2 line with the fused input-gradient chain
set -u
O=gpurun_out/r06_call19
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_dw_f32.py tests/test_conv_dw_u8.py tests/test_conv_bwd_fused.py tests/test_multi_dw.py tests/test_ppo_full_size.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest.txt; tail -30 $O/pytest.txt
run() { # name, flags
  timeout 500 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r.get('update_us_in_epoch_graph'), 'conv', d['box'].get('fused_conv_forward_in_update_us'), 'launches', r.get('kernel_launches_per_update'))
    for k in r['update_kernels']: print('      %-70s %5.1f x %4.1f' % (k['kernel'][:70], k['launches_per_update'], k['avg_us']))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run one "--steps 10 --warmup 3"
run three "--steps 10 --warmup 3 --conv-dw-one-launch 0"
run one2 "--steps 10 --warmup 3"
