#!/bin/bash
# round 6, call 14: full GPU suite with the fused SAC update + C4 / C5 lines (host or device bound?)
set -u
O=gpurun_out/r06_call14
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest.txt; tail -5 $O/pytest.txt
for w in c2 c5 c4; do
timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$w', (d.get('box',{}).get('settings') or {}).get('vbios'), d.get('box',{}).get('fused_conv_forward_in_update_us'), d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'), 'env steps/step', d['config'].get('env_steps_per_step_per_gpu'))
except Exception as e:
    print('ERR', e); print(open('$O/bench_$w.err').read()[-2000:])
PY
done
