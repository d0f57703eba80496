#!/bin/bash
# round 6, call 48: kernel trace of C1 (1-env DQN step graph): device time per env-step against the loop's wall time
set -u
O=gpurun_out/r06_call48
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python $R/bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/prof_c1.log 2>&1)
g=$(find /tmp/prof_c1 -name '*kernel_stats.csv' | head -1); [ -n "$g" ] && cp $g $O/c1_kernel_stats.csv
tail -1 $O/prof_c1.log | cut -c1-300
python - <<PY
import csv
tot=0
for r in csv.DictReader(open('$O/c1_kernel_stats.csv')):
    c=int(r['Calls']); a=float(r['AverageNs'])/1e3; tot+=c*a
    if c>=1000: print(r['Name'][:70], c, round(a,2))
print('total kernel ms', tot/1e3)
PY
