#!/bin/bash
# kernel traces of the C4 and C5 benches on the round's last code
set -u
O=gpurun_out/r02_final2
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for w in c4 c5; do
  (cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/$O/prof_$w.log 2>&1)
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  head -16 $O/${w}_kernel_stats.csv | cut -c1-150
done
