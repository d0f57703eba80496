#!/bin/bash
set -u
O=gpurun_out/r02_call21
mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_architecture.py tests/test_ac_nets.py -m gpu -q --tb=short -x 2>&1 | tail -40 > $O/pytest.txt
tail -30 $O/pytest.txt | cut -c1-250
for v in 0 1; do
RLX_NO_GEMM_PAIR=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c2_nopair$v.json 2> $O/bench_c2_nopair$v.err
python -c "
import json; d=json.loads(open('$O/bench_c2_nopair$v.json').read().strip().splitlines()[-1]); print('c2 nopair=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])"
RLX_NO_GEMM_PAIR=$v timeout 300 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3_nopair$v.json 2> $O/bench_c3_nopair$v.err
python -c "
import json; d=json.loads(open('$O/bench_c3_nopair$v.json').read().strip().splitlines()[-1]); print('c3 nopair=$v', d['ms_per_step'], d['value'])"
done
