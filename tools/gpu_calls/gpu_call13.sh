#!/bin/bash
set -u
O=gpurun_out/r02_call13
mkdir -p $O
timeout 600 python -m pytest tests/test_ppo_heads_fused.py tests/test_ppo_agent.py tests/test_nn.py tests/test_architecture.py -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest.txt
tail -25 $O/pytest.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/c2_fused.json 2> $O/c2_fused.err
RLX_NO_FUSED_HEADS=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/c2_nofused.json 2> $O/c2_nofused.err
for f in $O/c2_fused.json $O/c2_nofused.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
