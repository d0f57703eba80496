#!/bin/bash
# round 3, call 13: FETCH_SIZE / WRITE_SIZE calibration on launches with known byte counts
set -u
O=gpurun_out/r03_call13
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 200 python tools/pmc_calibrate.py 2>&1 | tail -3
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -- python $R/tools/pmc_calibrate.py > $R/$O/cal_$c.log 2>&1)
f=$(find /tmp/cal_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/cal_$c.csv
done
python tools/pmc_calibrate.py --summarise $O/cal_FETCH_SIZE.csv $O/cal_WRITE_SIZE.csv $O/pmc_calibration.json 2>&1 | tail -80
