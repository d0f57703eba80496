#!/bin/bash
# round 3, call 7: epoch-level minibatch gathers + direct-dX policy (tests, A/B, bench), reference-API memory adapters
set -u
O=gpurun_out/r03_call7
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_replay.py tests/test_per.py tests/test_ppo_agent.py tests/test_gemm.py tests/test_reference_image_loops.py tests/test_data_parallel_gpu.py tests/test_checkpoint.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $O/tests.txt
timeout 600 python tools/ab_c2.py 3 2>/dev/null | tail -1 | tee $O/ab_c2.json
timeout 400 python bench.py --shapes --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-330
timeout 300 python tools/c2_phase_split.py 2>/dev/null | tail -1 | tee $O/phase_split.txt
