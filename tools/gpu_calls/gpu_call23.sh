#!/bin/bash
set -u
O=gpurun_out/r02_call23
mkdir -p $O
for v in 96 64; do
for w in c4 c1 c5; do
RLX_GEMM_THIN_MAX_TILES=$v timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_thin$v.json 2> $O/bench_${w}_thin$v.err
python -c "
import json; d=json.loads(open('$O/bench_${w}_thin$v.json').read().strip().splitlines()[-1]); print('$w thin_max_tiles=$v', d['ms_per_step'], d['value'])"
done; done
timeout 600 python -m pytest tests/test_gemm.py tests/test_ac_nets.py -m gpu -q --tb=short -x 2>&1 | tail -5
