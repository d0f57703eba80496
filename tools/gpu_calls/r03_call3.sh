#!/bin/bash
# round 3, call 3: PPO-on-CartPole loop parity (where does the device loop leave the oracle's?), the GEMM family with the
# in-workgroup K split (32 x 64 / 32 x 32 tiles), the network / agent suites on top of it, then the C2 bench line + shapes
set -u
O=gpurun_out/r03_call3
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_cartpole.py -m gpu -q -s --tb=short -k "oracle_loop or golden" 2>&1 | tail -40 | tee $O/cartpole.txt
timeout 900 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_architecture.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $O/gemm_tests.txt
timeout 400 python bench.py --shapes --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-400; grep -A14 "products" $O/bench_c2.err | tail -20
timeout 200 python bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
