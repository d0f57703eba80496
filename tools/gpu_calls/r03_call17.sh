#!/bin/bash
# round 3, call 17: slabs in flight 4 vs 2, per kernel on ONE box (eager update, 50 repetitions each)
set -u
O=gpurun_out/r03_call17
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in librlx.so ab/librlx_depth2.so; do
n=$(basename $v .so)
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python $R/tools/ppo_update_once.py --lib $R/coach_amd/$v > $R/$O/kt_$n.log 2>&1)
f=$(find /tmp/kt_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv
done
python - <<'PY'
import csv
O="gpurun_out/r03_call17"
def load(v):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open("%s/kernel_stats_%s.csv" % (O, v)))}
a, b = load("librlx"), load("librlx_depth2")
print("%-100s %6s %10s %10s" % ("kernel", "calls", "depth 4", "depth 2"))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:16]:
    print("%-100s %6d %10.2f %10.2f" % (k[:100], a[k][0], a[k][1] / 1e3, b.get(k, (0, 0))[1] / 1e3))
PY
