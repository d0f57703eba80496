#!/bin/bash
# round 5, call 3: conv2 -> conv3 forward as one launch — bit-equality tests, the PPO parity tests through it, same-box A/B
set -u
O=gpurun_out/r05_call3
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_conv_fused.py tests/test_ppo_full_size.py tests/test_ppo_agent.py tests/test_nn.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -30 > $O/pytest.txt
tail -30 $O/pytest.txt
run() { # name, flags
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-14s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r['update_us_in_epoch_graph'], 'sum', r['update_us_sum_of_kernels'], r['update_us_by_family'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
    for k in r['update_kernels'][:6]: print('      ', k['kernel'][:70], k['launches_per_update'], k['us_per_update'])
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run unfused "--fuse-conv 0"
run fused "--fuse-conv 1"
run unfused2 "--fuse-conv 0"
run fused2 "--fuse-conv 1"
