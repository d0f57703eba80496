#!/bin/bash
# First GPU call of the next round: candidates written after round 2's GPU budget was spent.
#   RLX_DENSE_SMALL_BWD_WIDE=1  head-backward kernel on 16 features x 16 row groups per workgroup (twice the workgroups,
#                               one load wave up to 128 rows); different grouping of the dW row sums -> tolerance tests
#   RLX_FOLD_ONLY_BEYOND_THIN=1 shared-input layers that fit the thin kernel run as batched thin launches (16 x 16 tiles) instead
#                               of one folded tiled GEMM + split-K reduce (SAC's Q towers)
#   tools/gpu_calls/next_round_tests/   device agents against the recorded image loops of the real reference agents (move to tests/ when green)
# correctness first (the suites that drive the narrow-dense kernels), then a same-box A/B on C4 / C5 / C1
set -u
O=gpurun_out/next_round_ab
mkdir -p $O
RLX_DENSE_SMALL_BWD_WIDE=1 timeout 300 python -m pytest tests/test_nn.py tests/test_ac_nets.py tests/test_dqn_agent.py tests/test_agent_loops.py tests/test_reference_loop.py -q -m gpu 2>&1 | tail -5 | tee $O/tests_wide.txt
for w in c4 c5 c1; do for v in 0 1 0 1; do
  RLX_DENSE_SMALL_BWD_WIDE=$v timeout 60 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'wide=$v', d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done; done
RLX_FOLD_ONLY_BEYOND_THIN=1 timeout 300 python -m pytest tests/test_nn.py tests/test_ac_nets.py tests/test_agent_loops.py tests/test_reference_loop.py -q -m gpu 2>&1 | tail -5 | tee $O/tests_nofold.txt
for w in c5 c4; do for v in 0 1 0 1; do
  RLX_FOLD_ONLY_BEYOND_THIN=$v timeout 60 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'fold_only_beyond_thin=$v', d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done; done
timeout 300 python -m pytest tools/gpu_calls/next_round_tests -q -m gpu 2>&1 | tail -15 | tee $O/tests_image_loops.txt
