#!/bin/bash
# round 3, call 21: Clipped-PPO head losses + heads' backward as one launch — bit-equality tests, same-process A/B, C2 bench
set -u
O=gpurun_out/r03_call21
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn.py tests/test_ppo_agent.py tests/test_cartpole.py tests/test_data_parallel_gpu.py tests/test_graph_manager.py -m gpu -q --tb=short 2>&1 | tail -12 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_c2.json | cut -c1-200
