#!/bin/bash
# GPU call 1 (round 2): parity suite after the reference-order store / exact pow, all five workloads.
set -u
O=gpurun_out/r02_call1
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
echo "pytest rc=$?" >> $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err
done
tail -3 $O/pytest.txt
cat $O/bench_c*.json | cut -c1-400
