#!/bin/bash
# round 3, call 16: tiled GEMM kernels with 4 slabs in flight (was 2) — parity tests, same-box A/B of the two builds, C2 bench + trace
set -u
O=gpurun_out/r03_call16
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gemm.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_nn_graph.py -m gpu -q --tb=short 2>&1 | tail -8 | tee $O/tests.txt
timeout 900 python tools/ab_c2_libs.py coach_amd/librlx.so coach_amd/ab/librlx_depth2.so 2 2>&1 | tail -3 | tee $O/ab_depth.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_c2.json | cut -c1-200
grep -o '"roofline": {[^}]*}' $O/bench_c2.json | cut -c1-300
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/tools/ppo_update_once.py > $R/$O/kt.log 2>&1)
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_eager_update.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_call16/kernel_stats_eager_update.csv")))
for r in rows[:16]:
    print("%-100s %6s %8.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
