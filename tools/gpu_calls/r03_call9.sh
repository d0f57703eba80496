#!/bin/bash
# round 3, call 9: full GPU suite after the prune, smoke, the C2 loss-curve hip side at the FULL config (10 epochs)
set -u
O=gpurun_out/r03_call9
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q --tb=short --durations=6 2>&1 | tail -40 > $O/pytest.txt
tail -14 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python tools/loss_curve_c2.py --side hip --dir gpurun_out/lc_c2_full --iterations 49 --epochs 10 2>&1 | tail -1
for w in c4 c5 c1 c3; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json; python -c "import json;d=json.load(open('$O/bench_$w.json'));print('$w',d['value'],d['ms_per_step'],d.get('roofline',{}).get('update_us'))"; done
