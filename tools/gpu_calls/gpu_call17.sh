#!/bin/bash
set -u
O=gpurun_out/r02_call17
mkdir -p $O
timeout 900 python -m pytest tests/test_ac_nets.py tests/test_agent_loops.py tests/test_architecture.py tests/test_data_parallel_gpu.py tests/test_checkpoint.py -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest.txt
tail -25 $O/pytest.txt | cut -c1-220
for w in c4 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d['roofline'].get('library_calls_per_update'), d['roofline'].get('update_us'))"; done
