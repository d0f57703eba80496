#!/bin/bash
# round 6, call 18: full GPU suite with rlx_conv_dw_u8 / rlx_conv_dw_f32 and the fused input-gradient chain as defaults + C2 / C3 lines
set -u
O=gpurun_out/r06_call18
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest.txt; tail -8 $O/pytest.txt
for w in c2; do
timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$w', (d.get('box',{}).get('settings') or {}).get('vbios'), d.get('box',{}).get('fused_conv_forward_in_update_us'), d['value'], d['ms_per_step'], 'update_us', r.get('update_us', r.get('update_us_in_epoch_graph')), 'frac', r['frac'])
except Exception as e:
    print('ERR', e); print(open('$O/bench_$w.err').read()[-2000:])
PY
done
