#!/bin/bash
set -u
O=gpurun_out/r02_call19
mkdir -p $O
for k in 0 600 1024; do
  RLX_GEMM_SPLIT_MIN_K=$k timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --shapes > $O/c2_mink_$k.json 2> $O/c2_mink_$k.err
  python -c "
import json; d=json.loads(open('$O/c2_mink_$k.json').read().strip().splitlines()[-1]); print('split min K $k:', d['ms_per_step'], d['value'], d['roofline']['frac'])"
  grep "^{" $O/c2_mink_$k.err | cut -c1-100 | head -4
done
timeout 900 python tools/loss_curve.py --steps 100000 --out $O/loss_curve_c1.json > $O/loss_curve.log 2>&1
tail -5 $O/loss_curve.log | cut -c1-300
