#!/bin/bash
# round 5, call 19: rlx_conv32_input_grad on 32 x 32 x 2 MFMAs with register-prefetched weights: bit-equality, timeline, A/B
set -u
O=gpurun_out/r05_call19
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_bwd_fused.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -30
python tools/conv32_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('%-14s' % '$1', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'fused', d.get('box',{}).get('fused_conv_forward_in_update_us'))
    for k in r.get('update_kernels', []): print('      %-70s %7.2f x %s' % (k['kernel'][:70], k['avg_us'], k.get('launches_per_update')))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c2_bwd_fused "--steps 10 --warmup 3 --fuse-conv-bwd 1"
run c2_bwd_pairs "--steps 10 --warmup 3 --fuse-conv-bwd 0"
