#!/bin/bash
set -u
O=gpurun_out/r02_call24
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -40 > $O/pytest.txt
tail -25 $O/pytest.txt | cut -c1-250
for w in c2 c3 c4 c5; do
for v in 0 1; do
extra="--workload $w"; [ $w = c2 ] && extra="--steps 8 --warmup 3"
RLX_ADAM_TWO_LAUNCHES=$v timeout 300 python bench.py $extra --no-cpu-baseline > $O/bench_${w}_two$v.json 2> $O/bench_${w}_two$v.err
python -c "
import json; d=json.loads(open('$O/bench_${w}_two$v.json').read().strip().splitlines()[-1]); print('$w two_launch_adam=$v', d['ms_per_step'], d['value'])"
done; done
