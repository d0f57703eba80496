#!/bin/bash
# round 4, call 7: data-parallel tests with RCCL collectives inside the graphs; bench.py with / without --force-dist
# (structural cost of the collective path at world size 1); GEMM pipeline A/B (baseline vs ring depth 2, 3 WGs / CU)
set -u
O=gpurun_out/r04_call7
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/dp_tests.txt 2>&1
tail -25 $O/dp_tests.txt | cut -c1-220
for mode in plain force-dist; do
  flag=""; [ $mode = force-dist ] && flag="--force-dist"
  timeout 400 python bench.py --no-cpu-baseline --no-roofline $flag > $O/bench_$mode.json 2> $O/bench_$mode.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$mode.json').read().strip().splitlines()[-1])
    print('$mode', d['value'], d['ms_per_step'], d.get('rccl'))
except Exception as e:
    print('$mode ERR', e); print(open('$O/bench_$mode.err').read()[-1500:])
PY
done
timeout 600 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:0 coach_amd/ab/librlx_d2w3.so:1 2>&1 | grep -v amdgpu.ids | tee $O/ab_pipeline.txt
