#!/bin/bash
# round 5, call 4: conv2 -> conv3 fused forward with a deeper weight ring — bit-equality at every depth, same-box A/B of the depth
set -u
O=gpurun_out/r05_call4
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_conv_fused.py tests/test_ppo_full_size.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -30 > $O/pytest.txt
tail -12 $O/pytest.txt
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
run d2 "--conv23-depth 2"
run d3 "--conv23-depth 3"
run d4 "--conv23-depth 4"
run d6 "--conv23-depth 6"
run d8 "--conv23-depth 8"
run d4b "--conv23-depth 4"
