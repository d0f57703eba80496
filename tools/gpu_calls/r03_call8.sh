#!/bin/bash
# round 3, call 8: data-parallel update vs the oracle, the C2 loss-curve hip side, kernel traces (C2, C3) and PMC passes
set -u
O=gpurun_out/r03_call8
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q --tb=short 2>&1 | tail -25 | tee $O/dp_tests.txt
timeout 600 python tools/loss_curve_c2.py --side hip --dir gpurun_out/lc_c2 --iterations 49 --epochs 2 2>&1 | tail -2
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-600
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/prof_c3.log 2>&1)
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
rocprofv3 -L 2>/dev/null | grep -i "mfma\|GRBM_GUI_ACTIVE\|FETCH_SIZE\|WRITE_SIZE" | head -30 > $O/counters_available.txt
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/ppo_update_once.py > $R/$O/pmc_$c.log 2>&1)
f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_gemm_traffic.json 2>&1 | tail -3
python tools/pmc_summary.py --mfma $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv $O/pmc_GRBM_GUI_ACTIVE.csv $O/pmc_mfma_util.json 2>&1 | tail -30
ls -la $O gpurun_out/lc_c2 | head -40
