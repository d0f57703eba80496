#!/bin/bash
# round 3, call 22: PER store of consecutive leaves without the rank loop and without the sibling loads a neighbour covers
set -u
O=gpurun_out/r03_call22
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_per.py tests/test_replay.py tests/test_dqn_agent.py tests/test_reference_loop.py -m gpu -q --tb=short 2>&1 | tail -8 | tee $O/tests.txt
timeout 300 python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_c3.json | cut -c1-250
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/prof_c3.log 2>&1)
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
head -4 $O/c3_kernel_stats.csv | cut -d'"' -f2,3 | cut -c1-50,190-300
