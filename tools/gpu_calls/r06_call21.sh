#!/bin/bash
# round 6, call 21: rlx_ppo_fc_rows (row-local heads inside the FC reduction, the all-rows part on the deferred-reduction launch):
# bit identity with the three-launch path, the PPO parity tests through it, A/B of the C2 line
set -u
O=gpurun_out/r06_call21
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ppo_fc_rows.py tests/test_ppo_fc_fused.py tests/test_gemm.py tests/test_ppo_full_size.py tests/test_ppo_long_episodes.py tests/test_ppo_agent.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest.txt; tail -40 $O/pytest.txt
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
run rows "--steps 10 --warmup 3"
run three "--steps 10 --warmup 3 --heads-row-local 0"
run rows2 "--steps 10 --warmup 3"
