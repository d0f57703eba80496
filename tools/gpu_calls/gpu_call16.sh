#!/bin/bash
set -u
O=gpurun_out/r02_call16
mkdir -p $O
timeout 600 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_agent.py tests/test_architecture.py tests/test_dqn_agent.py tests/test_ac_nets.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 200 python tools/gemm_timeline.py > $O/gemm_timeline.txt 2>&1
cat $O/gemm_timeline.txt | cut -c1-170
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --shapes > $O/bench_c2.json 2> $O/bench_c2.err
python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['value'], d['roofline']['frac'])"
grep "^{" $O/bench_c2.err | cut -c1-120
for w in c3 c4 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'])"; done
