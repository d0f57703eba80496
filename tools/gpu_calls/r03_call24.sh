#!/bin/bash
# round 3, call 24: wave streams on the small GEMM tiles — parity tests, same-process A/B on the C2 update, per-kernel trace
set -u
O=gpurun_out/r03_call24
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_agent.py tests/test_dqn_agent.py -m gpu -q --tb=short 2>&1 | tail -12 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/tools/ppo_update_once.py > $R/$O/kt.log 2>&1)
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_eager_update.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_call24/kernel_stats_eager_update.csv")))
for r in rows[:14]:
    print("%-100s %6s %8.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
