#!/bin/bash
# round 3, call 5: direct convolution input gradient (windowed gather) — tests, then the C2 bench line + shapes + phases
set -u
O=gpurun_out/r03_call5
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm.py -m gpu -q --tb=short 2>&1 | tail -40 | tee $O/gemm_tests.txt
timeout 900 python -m pytest tests/test_nn.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_architecture.py tests/test_reference_image_loops.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $O/tests.txt
timeout 400 python bench.py --shapes --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-330; grep "products" $O/bench_c2.err | tail -14
timeout 300 python tools/c2_phase_split.py 2>/dev/null | tail -1 | tee $O/phase_split.txt
timeout 200 python bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
