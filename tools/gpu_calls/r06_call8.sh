#!/bin/bash
# round 6, call 8: rlx_ppo_fc_heads in the C2 update: per-dispatch table, same-box A/B of the bench line
set -u
O=gpurun_out/r06_call8
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2>/dev/null
timeout 300 python tools/dispatch_histogram.py --updates 100 --kernel ppo_fc_heads > $O/dispatch_hist.txt 2>&1; grep -v "amdgpu.ids" $O/dispatch_hist.txt | head -16
run() { # name, flags
  timeout 500 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'gemm_us', r['gemm_us_per_update'], 'update_us', r.get('update_us_in_epoch_graph'), 'conv', d['box'].get('fused_conv_forward_in_update_us'), 'launches', r.get('kernel_launches_per_update'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run fused "--steps 10 --warmup 3"
run unfused "--steps 10 --warmup 3 --fc-heads 0"
run fused2 "--steps 10 --warmup 3"
