#!/bin/bash
# round 3, call 20: 4 workgroups per CU for the tiled kernels without gather tables (launch bounds) vs the build in the tree
set -u
O=gpurun_out/r03_call20
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python tools/ab_c2_libs.py coach_amd/ab/librlx_lb4.so coach_amd/librlx.so 2 2>&1 | tail -3 | tee $O/ab_libs.json
for v in ab/librlx_lb4.so librlx.so; do
n=$(basename $v .so)
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python $R/tools/ppo_update_once.py --lib $R/coach_amd/$v > $R/$O/kt_$n.log 2>&1)
f=$(find /tmp/kt_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv
done
python - <<'PY'
import csv
O="gpurun_out/r03_call20"
def load(v):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open("%s/kernel_stats_%s.csv" % (O, v)))}
a, b = load("librlx_lb4"), load("librlx")
print("%-100s %6s %10s %10s" % ("kernel", "calls", "4 WG/CU", "in tree"))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:9]:
    print("%-100s %6d %10.2f %10.2f" % (k[:100], a[k][0], a[k][1] / 1e3, b.get(k, (0, 0))[1] / 1e3))
PY
