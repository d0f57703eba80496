#!/bin/bash
# round 3, call 15: tile order handed to the XCDs (rlx_gemm_tuning xcd_mode 0 / 8 / -1): per-kernel durations over 50 eager
# updates, the traffic of the default order, the same-process A/B of the whole update, parity tests
set -u
O=gpurun_out/r03_call15
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in 0 8 -1; do
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python $R/tools/ppo_update_once.py --xcd-mode $v > $R/$O/kt_$v.log 2>&1)
f=$(find /tmp/kt_$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$v.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/ppo_update_once.py --xcd-mode 8 > $R/$O/pmc_$c.log 2>&1)
f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv
done
timeout 900 python -m pytest tests/test_gemm.py -m gpu -q --tb=short 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
python tools/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_gemm_traffic_groups_of_8.json 2>&1 | tail -3
python - <<'PY'
import csv
O="gpurun_out/r03_call15"
def load(v):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open("%s/kernel_stats_%s.csv" % (O, v)))}
a, b, c = load("0"), load("8"), load("-1")
print("%-100s %6s %10s %10s %10s" % ("kernel", "calls", "plain us", "groups8 us", "share us"))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:16]:
    print("%-100s %6d %10.2f %10.2f %10.2f" % (k[:100], a[k][0], a[k][1] / 1e3, b.get(k, (0, 0))[1] / 1e3, c.get(k, (0, 0))[1] / 1e3))
PY
