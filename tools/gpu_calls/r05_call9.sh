#!/bin/bash
# round 5, call 9: several accumulator tiles per wave for large products: bit-equality tests, the GEMM suite, the box figure,
# C2 at both episode lengths with / without
set -u
O=gpurun_out/r05_call9
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm.py tests/test_ppo_full_size.py tests/test_ppo_long_episodes.py -m gpu -q --tb=short -p no:cacheprovider -k "big or full_size or long_episodes or nn_bias or conv_forward" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -15
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r['update_us_in_epoch_graph'], 'box gemm', d['box']['gemm_4096_fp32_TFLOPs'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run small "--big-tiles 0 --steps 10 --warmup 3"
run big "--big-tiles 1 --steps 10 --warmup 3"
run small_L1024 "--big-tiles 0 --episode-length 1024 --steps 4 --warmup 2"
run big_L1024 "--big-tiles 1 --episode-length 1024 --steps 4 --warmup 2"
run small2 "--big-tiles 0 --steps 10 --warmup 3"
run big2 "--big-tiles 1 --steps 10 --warmup 3"
