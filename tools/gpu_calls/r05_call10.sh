#!/bin/bash
# round 5, call 10: the direct convolution input gradient (one windowed-gather product instead of dcol + col2im) re-measured on
# this round's tree; test_gemm's big-tile tests under both main loops
set -u
O=gpurun_out/r05_call10
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q --tb=short -p no:cacheprovider -k "big" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-10s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r['update_us_in_epoch_graph'], r['update_us_by_family'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
    if '$1' in ('always',):
        for k in r['update_kernels']: print('      ', k['kernel'][:70], k['launches_per_update'], k['us_per_update'])
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run off "--direct-conv-dx 0"
run on "--direct-conv-dx 1"
run always "--direct-conv-dx always"
run off2 "--direct-conv-dx 0"
