#!/bin/bash
set -u
O=gpurun_out/r02_call10
mkdir -p $O
for v in base "RLX_GEMM_SPLIT_MIN_K=600" "RLX_GEMM_SPLIT_MIN_K=1024" "RLX_GEMM_SPLIT_MIN_K=4000"; do
  if [ "$v" = base ]; then python bench.py --steps 6 --warmup 2 --no-cpu-baseline --shapes > $O/c2_base.json 2> $O/c2_base.err
  else env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --shapes > $O/c2_$v.json 2> $O/c2_$v.err; fi
done
for f in $O/c2_*.json; do echo $f; python -c "
import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_us_per_update'])"; done
for f in $O/c2_*.err; do echo $f; grep '^{' $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['M'],d['N'],d['K'],d['batch'],round(d['us'],1))"; done
