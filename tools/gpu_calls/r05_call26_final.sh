#!/bin/bash
# round 5, final evidence of the final tree (the update's kernels are those of profiles/r05_final_*; what changed since is the
# acting step — both towers, V(s) / probabilities recorded — and fill_advantages): full GPU suite, smoke, the default bench line
# (roofline + box + cpu_baseline), C2 at L = 1024, rocprofv3 --kernel-trace --stats of the bench command itself
set -u
O=gpurun_out/r05_final2
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.txt
tail -4 $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --episode-length 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_c2_L1024.json 2> $O/bench_c2_L1024.err
for w in c2 c2_L1024; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    r=d.get('roofline', {})
    print('$w', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'update_us', r.get('update_us_in_epoch_graph'), 'cpu', d.get('cpu_baseline',{}).get('value'))
    if '$w' == 'c2': print('   box', d.get('box'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
tail -1 $O/prof_c2.log | cut -c1-300 > $O/prof_c2_bench_line.json
head -14 $O/c2_kernel_stats.csv | cut -c1-150
du -sh $O
