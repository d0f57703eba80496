#!/bin/bash
# round 3, call 14: XCD-chunked tile order of the tiled GEMM kernels — parity tests, same-process A/B, C2 bench + trace
set -u
O=gpurun_out/r03_call14
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gemm.py tests/test_ppo_agent.py tests/test_dqn_agent.py -m gpu -q --tb=short 2>&1 | tail -15 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 4 2>/dev/null | tail -1 | tee $O/ab_c2.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_c2.json | cut -c1-200
grep -o '"roofline": {[^}]*}' $O/bench_c2.json | cut -c1-400
