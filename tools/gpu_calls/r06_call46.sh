#!/bin/bash
# round 6, call 46: the driver's own round-end commands on the final tree
set -u
O=gpurun_out/r06_call46
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee $O/pytest_x.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['config']); print({k:d['roofline'][k] for k in ('bound','achieved','peak','unit','frac','traffic')}); print({k:d['cpu_baseline'][k] for k in ('value','unit','cores','kind')})
PY
