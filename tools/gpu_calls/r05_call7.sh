#!/bin/bash
# round 5, call 7: conv23 with S slabs per synchronisation step: tests at every (depth, step), timeline, A/B
set -u
O=gpurun_out/r05_call7
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_fused.py tests/test_ppo_full_size.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -12
for cfg in "4 1" "4 2" "8 2"; do set -- $cfg; echo "== depth $1 step $2"; python tools/conv23_timeline.py --depth $1 --step $2 2>&1 | grep -v amdgpu.ids | tee $O/timeline_d$1_s$2.txt | grep -A6 "half 1\|span"; done
run() { # name, flags
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    c23=[k['us_per_update'] for k in r['update_kernels'] if k['kernel'].startswith('conv23')]
    print('%-10s' % '$1', d['value'], d['ms_per_step'], 'update_us', r['update_us_in_epoch_graph'], 'sum', r['update_us_sum_of_kernels'], 'conv23', c23, r['update_us_by_family'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run unfused "--fuse-conv 0"
run d4s1 "--conv23-depth 4 --conv23-step 1"
run d4s2 "--conv23-depth 4 --conv23-step 2"
run d6s2 "--conv23-depth 6 --conv23-step 2"
run d8s2 "--conv23-depth 8 --conv23-step 2"
run d4s1b "--conv23-depth 4 --conv23-step 1"
