#!/bin/bash
# round 3, call 1: the full root-level GPU suite (no -x), smoke, the C2 bench line, then the two round-2 candidates' A/B
set -u
O=gpurun_out/r03_call1
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q --tb=short --durations=8 2>&1 | tail -60 > $O/pytest.txt
tail -12 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json
RLX_DENSE_SMALL_BWD_WIDE=1 timeout 300 python -m pytest tests/test_nn.py tests/test_ac_nets.py tests/test_dqn_agent.py tests/test_agent_loops.py tests/test_reference_loop.py -q -m gpu 2>&1 | tail -5 | tee $O/tests_wide.txt
for w in c4 c5 c1; do for v in 0 1 0 1; do
  RLX_DENSE_SMALL_BWD_WIDE=$v timeout 60 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'wide=$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done; done
RLX_FOLD_ONLY_BEYOND_THIN=1 timeout 300 python -m pytest tests/test_nn.py tests/test_ac_nets.py tests/test_agent_loops.py tests/test_reference_loop.py -q -m gpu 2>&1 | tail -5 | tee $O/tests_nofold.txt
for w in c5 c4; do for v in 0 1 0 1; do
  RLX_FOLD_ONLY_BEYOND_THIN=$v timeout 60 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'fold_only_beyond_thin=$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done; done
