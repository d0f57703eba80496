#!/bin/bash
set -u
O=gpurun_out/r02_call15
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short --durations=5 2>&1 | tail -40 > $O/pytest.txt
tail -8 $O/pytest.txt
timeout 200 python tools/gemm_timeline.py > $O/gemm_timeline.txt 2>&1
cat $O/gemm_timeline.txt | cut -c1-170
for v in "2 256" "3 256" "4 256" "4 512" "6 1024" "1 256"; do
  set -- $v
  RLX_GEMM_SPLIT_WGS_PER_CU=$1 RLX_GEMM_SPLIT_MAX_TILES=$2 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --shapes > $O/c2_split_$1_$2.json 2> $O/c2_split_$1_$2.err
  python -c "
import json; d=json.loads(open('$O/c2_split_$1_$2.json').read().strip().splitlines()[-1]); print('split wgs/cu $1 max tiles $2:', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
