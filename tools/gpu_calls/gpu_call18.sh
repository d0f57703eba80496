#!/bin/bash
set -u
O=gpurun_out/r02_call18
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_steps.py tests/test_dqn_agent.py tests/test_agent_loops.py tests/test_reference_loop.py tests/test_mlp_fused.py tests/test_explore_env_losses.py -m gpu -q --tb=short 2>&1 | tail -60 > $O/pytest.txt
tail -40 $O/pytest.txt | cut -c1-220
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python $R/bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/prof_c1.log 2>&1)
f=$(find /tmp/prof_c1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c1_kernel_stats.csv
head -14 $O/c1_kernel_stats.csv | cut -c1-150
timeout 300 python bench.py --workload c1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err; python -c "
import json; d=json.loads(open('$O/bench_c1.json').read().strip().splitlines()[-1]); print('c1', d['ms_per_step'], d['value'])"
