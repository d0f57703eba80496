#!/bin/bash
# round 6, call 2: the fused TD3 update — parity against the oracle (both paths), its time against the launch chain
set -u
O=gpurun_out/r06_call2
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/box_info.json'))['summary']; print('card', d['card'], d['gpu_unique_id'], 'numa', d['gpu_numa_node'], 'neighbours', d['neighbours_busy_W'])"
timeout 600 python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -x -k "td3" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -30
timeout 300 python tools/ac_fused_bench.py td3 2>&1 | grep -v "amdgpu.ids" | tee $O/ac_fused_bench_td3.txt
timeout 400 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.err; python - <<PY
import json
d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c4', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'), 'under_load', d['box'].get('under_load'))
PY
