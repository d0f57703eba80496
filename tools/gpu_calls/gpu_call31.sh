#!/bin/bash
set -u
O=gpurun_out/r02_call31; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm.py tests/test_ac_nets.py tests/test_nn.py tests/test_architecture.py tests/test_agent_loops.py tests/test_dqn_agent.py tests/test_mlp_fused.py tests/test_fused_steps.py -m gpu -q --tb=short -x 2>&1 | tail -25 | cut -c1-250
for v in 0 1; do
for w in c4 c1; do
RLX_NO_GEMM_CHAIN=$v timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_nochain$v.json 2> $O/bench_${w}_nochain$v.err
python -c "
import json; d=json.loads(open('$O/bench_${w}_nochain$v.json').read().strip().splitlines()[-1]); print('$w nochain=$v', d['ms_per_step'], d['value'])"
done; done
