#!/bin/bash
set -u
O=gpurun_out/r02_call32; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q --tb=short -x -k "two_dense_layers" 2>&1 | tail -4 | cut -c1-250
for v in 1 0; do
RLX_GEMM_CHAIN=$v timeout 300 python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_c4_chain$v.json 2> $O/bench_c4_chain$v.err
python -c "
import json; d=json.loads(open('$O/bench_c4_chain$v.json').read().strip().splitlines()[-1]); print('c4 chain=$v', d['ms_per_step'], d['value'])"
done
