#!/bin/bash
# round 5, call 21: device side of the IDENTICAL-HISTORY loss-curve experiment (8 seeds, 10 epochs per rollout, 49
# iterations; the oracle then records the device's actions: tools/loss_curve_c2.py --follow-hip-actions), the acting-step
# tests of the last commit, one default bench line
set -u
O=gpurun_out/r05_call21
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 300 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4,5,6,7 --dir gpurun_out/lc_forced --iterations 49 --epochs 10 --no-init 2>&1 | tail -8
timeout 200 python -m pytest tests/test_acting_fused.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5
timeout 200 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1])
    r=d.get('roofline', {})
    print('c2', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'gemm_us', r.get('gemm_us_per_update'), d.get('box'))
except Exception as e:
    print('c2 ERR', e); print(open('$O/bench_c2.err').read()[-800:])
PY
