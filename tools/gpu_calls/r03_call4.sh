#!/bin/bash
# round 3, call 4: deferred weight-gradient reductions (tests + C2/C3 bench lines), CartPole golden tests over 8 agent seeds
set -u
O=gpurun_out/r03_call4
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_architecture.py tests/test_ac_nets.py tests/test_agent_loops.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $O/tests.txt
timeout 400 python bench.py --shapes --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-330; grep "products" $O/bench_c2.err | tail -12
timeout 200 python bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
timeout 900 python tools/cartpole_golden_sweep.py 8 2>&1 | grep -v amdgpu.ids | tee $O/golden_sweep.txt
