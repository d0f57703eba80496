#!/bin/bash
set -u
O=gpurun_out/r02_call11
mkdir -p $O
timeout 600 python -m pytest tests/test_agent_loops.py tests/test_dqn_agent.py tests/test_reference_loop.py -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest.txt
tail -30 $O/pytest.txt
timeout 300 python bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
cut -c1-330 $O/bench_c1.json; tail -3 $O/bench_c1.err
