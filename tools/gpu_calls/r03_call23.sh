#!/bin/bash
# round 3, call 23: the heads' forward launch sums the FC layer's split-K partials — bit-equality tests, same-process A/B
set -u
O=gpurun_out/r03_call23
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn.py tests/test_ppo_agent.py tests/test_cartpole.py tests/test_data_parallel_gpu.py tests/test_architecture.py tests/test_reference_image_loops.py -m gpu -q --tb=short 2>&1 | tail -12 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
