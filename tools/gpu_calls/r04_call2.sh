#!/bin/bash
# round 4, call 2: LDS-DMA ring GEMM main loop — test_gemm with both pipelines, then the network / agent tests, then the C2
# bench line with --gemm-pipeline 1 and 0 (in-update kernel table of each).
set -u
O=gpurun_out/r04_call2
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_gemm.py -m gpu -q --tb=short -x 2>&1 | tail -40 > $O/test_gemm.txt
tail -15 $O/test_gemm.txt
if grep -q "failed\|error" $O/test_gemm.txt; then echo "GEMM TESTS FAILED: stopping"; exit 1; fi
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 --deselect tests/test_gemm.py 2>&1 | tail -40 > $O/pytest.txt
tail -12 $O/pytest.txt
for pl in 1 0; do
timeout 500 python bench.py --no-cpu-baseline --gemm-pipeline $pl > $O/bench_c2_pl$pl.json 2> $O/bench_c2_pl$pl.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_c2_pl$pl.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('pipeline $pl: c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'warm', r['frac_warm'], 'gemm_us', r['gemm_us_per_update'], r['gemm_us_per_update_warm'])
    print(r['update_us_by_family'], r['update_us_sum_of_kernels'], r['update_us_in_epoch_graph'])
    for k in r['update_kernels']: print('  %-70s %5.2f x %7.2f = %7.2f' % (k['kernel'][:70], k['launches_per_update'], k['avg_us'], k['us_per_update']))
    print(d['box'])
except Exception as e:
    print('ERR', e); print(open('$O/bench_c2_pl$pl.err').read()[-800:])
PY
done
