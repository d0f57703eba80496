#!/bin/bash
set -u
O=gpurun_out/r02_call27; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for w in c4 c5; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_$w.log 2>&1)
f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
done
python - <<'PY'
import csv
for w in ('c4','c5'):
    rows=list(csv.DictReader(open('gpurun_out/r02_call27/%s_kernel_stats.csv'%w)))
    tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
    print(w,'kernel ms',round(tot/1e6,1),'launches',n)
    for r in rows[:16]:
        print('  %-62s %8d %7.2f %5.1f%%'%(r['Name'][:62],int(r['Calls']),float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY
