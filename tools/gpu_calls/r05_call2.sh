#!/bin/bash
# round 5, call 2: the Adam rider — its tests, the PPO parity tests that run through it, the C3 full-size test again, and a
# same-box A/B of the C2 line: rider off / on (512 workgroups on the first pair launch) / other block counts / split over
# both pair launches
set -u
O=gpurun_out/r05_call2
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_adam_rider.py tests/test_dqn_full_size.py tests/test_ppo_full_size.py tests/test_ppo_long_episodes.py tests/test_ppo_agent.py tests/test_ppo_eval_reset.py tests/test_reference_image_loops.py tests/test_nn.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -45 > $O/pytest.txt
tail -45 $O/pytest.txt
run() { # name, flags
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-14s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r['update_us_in_epoch_graph'], 'sum', r['update_us_sum_of_kernels'], r['update_us_by_family'], 'conv1', d['box'].get('conv1_forward_in_update_us'))
    if '$1' in ('off', 'on512'):
        for k in r['update_kernels']: print('      ', k['kernel'][:70], k['launches_per_update'], k['us_per_update'])
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run off "--adam-rider 0"
run on512 "--adam-rider 1"
run on256 "--adam-rider 1 --rider-blocks 256"
run on1024 "--adam-rider 1 --rider-blocks 1024"
run on512x2 "--adam-rider 1 --rider-blocks 512 --rider-launches 2"
run on1024x2 "--adam-rider 1 --rider-blocks 1024 --rider-launches 2"
run off_again "--adam-rider 0"
