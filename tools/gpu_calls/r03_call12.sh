#!/bin/bash
# round 3, call 12: the PER path kernel that meets only in LDS — equality of the two kernels, A/B timing, C3 bench + trace
set -u
O=gpurun_out/r03_call12
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_per.py tests/test_replay.py tests/test_dqn_agent.py -m gpu -q --tb=short 2>&1 | tail -25 | tee $O/tests.txt
timeout 300 python tools/ab_per_update.py 2>&1 | tee $O/ab_per_update.txt
timeout 300 python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_c3.json | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/prof_c3.log 2>&1)
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
head -8 $O/c3_kernel_stats.csv | cut -c1-200
