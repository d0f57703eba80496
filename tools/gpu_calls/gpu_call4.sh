#!/bin/bash
set -u
O=gpurun_out/r02_call4
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp_fused.py tests/test_dqn_agent.py tests/test_agent_loops.py tests/test_reference_loop.py -m gpu -q --tb=short 2>&1 | tail -80 > $O/pytest.txt
timeout 300 python bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
RLX_NO_FUSED_MLP=1 timeout 300 python bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c1_nofused.json 2>> $O/bench_c1.err
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python $R/bench.py --workload c1 --steps 3 --warmup 2 --no-cpu-baseline > $R/$O/rocprof.log 2>&1
cd $R
find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c1_kernel_stats.csv
tail -5 $O/rocprof.log
tail -30 $O/pytest.txt; cut -c1-300 $O/bench_c1.json $O/bench_c1_nofused.json; head -40 $O/c1_kernel_stats.csv | cut -c1-170
