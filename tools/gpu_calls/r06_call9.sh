#!/bin/bash
# round 6, call 9: the fused SAC update (six launches): parity tests of both paths, device time per update, C5 line + kernel trace
set -u
O=gpurun_out/r06_call9
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2>/dev/null
timeout 900 python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -k "sac or td3" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest_sac.txt; tail -40 $O/pytest_sac.txt
timeout 300 python tools/ac_fused_bench.py sac 2>&1 | grep -v "amdgpu.ids" > $O/ac_fused_bench_sac.txt; head -40 $O/ac_fused_bench_sac.txt
timeout 400 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv; rm -rf $O/prof_c5
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('c5', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'))
except Exception as e:
    print('ERR', e); print(open('$O/bench_c5.err').read()[-2000:])
PY
head -14 $O/c5_kernel_stats.csv | cut -c1-150
