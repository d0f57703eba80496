#!/bin/bash
# round 5, call 8: convolution weight gradients as one launch behind the dX chain: bit-equality, the suites that run through
# Sequential.backward, same-box A/B
set -u
O=gpurun_out/r05_call8
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_dw.py tests/test_conv_fused.py tests/test_ppo_full_size.py tests/test_nn.py tests/test_dqn_full_size.py tests/test_reference_image_loops.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25
run() { # name, flags
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-10s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r['update_us_in_epoch_graph'], 'sum', r['update_us_sum_of_kernels'], r['update_us_by_family'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
    if '$1' in ('multi', 'single'):
        for k in r['update_kernels']: print('      ', k['kernel'][:70], k['launches_per_update'], k['us_per_update'])
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run single "--multi-dw 0"
run multi "--multi-dw 1"
run single2 "--multi-dw 0"
run multi2 "--multi-dw 1"
