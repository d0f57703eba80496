#!/bin/bash
# round 5, call 12: the pair kernels without the rider (the FC layer's dW + dX pair had lost a workgroup per CU to its registers),
# and the convolution pairs compiled for 4 workgroups per CU
set -u
O=gpurun_out/r05_call12
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_full_size.py tests/test_conv_fused.py tests/test_multi_dw.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -6
run() { # name, flags
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    ks={k['kernel']:k['us_per_update'] for k in r['update_kernels']}
    print('%-10s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r['update_us_in_epoch_graph'], 'FCpair', ks.get('gemm_dma_pair_kernel<false, 2>'), 'pairs', ks.get('gemm_dma_pair_kernel<true, 1>'), ks.get('gemm_dma_pair_kernel<true, 1, 4>'), 'conv1', d['box'].get('conv1_forward_in_update_us'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run wgs3 "--pair-wgs 3"
run wgs4 "--pair-wgs 4"
run unfused3 "--pair-wgs 3 --fuse-conv 0"
run wgs3b "--pair-wgs 3"
run wgs4b "--pair-wgs 4"
