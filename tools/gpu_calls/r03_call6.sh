#!/bin/bash
# round 3, call 6: same-box A/B of the round's GEMM changes, then a kernel trace of the C2 bench on the same box
set -u
O=gpurun_out/r03_call6
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
tail -1 $O/prof_c2.log | cut -c1-300
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/c2_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6)
for r in rows[:26]:
    print('%-90s %6d %8.1f us %6.2f%%'%(r['Name'][:90],int(r['Calls']),float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY
