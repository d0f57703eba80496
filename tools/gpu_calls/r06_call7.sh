#!/bin/bash
# round 6, call 7: the whole GPU suite on the tree with the fused TD3 update + C4 line and kernel trace
set -u
O=gpurun_out/r06_call7
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest.txt; tail -5 $O/pytest.txt
grep -n "uniform:\|PER:" $O/pytest.txt | head
timeout 300 python tools/ac_fused_bench.py td3 2>&1 | grep -v "amdgpu.ids" > $O/ac_fused_bench_td3.txt; head -3 $O/ac_fused_bench_td3.txt
timeout 400 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv; rm -rf $O/prof_c4
python - <<PY
import json
d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c4', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'))
PY
head -14 $O/c4_kernel_stats.csv | cut -c1-150
