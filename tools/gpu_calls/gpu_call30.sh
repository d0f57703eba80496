#!/bin/bash
set -u
O=gpurun_out/r02_call30; mkdir -p $O
timeout 1200 python -m pytest tests/test_ppo_heads_fused.py tests/test_ppo_agent.py tests/test_nn.py tests/test_architecture.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -25 | cut -c1-250
for v in 0 1; do
RLX_PPO_HEADS_THREE_LAUNCHES=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c2_three$v.json 2> $O/bench_c2_three$v.err
python -c "
import json; d=json.loads(open('$O/bench_c2_three$v.json').read().strip().splitlines()[-1]); print('c2 three_launch_heads=$v', d['ms_per_step'], d['value'])"
done
