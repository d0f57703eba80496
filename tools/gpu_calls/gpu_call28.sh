#!/bin/bash
set -u
O=gpurun_out/r02_call28; mkdir -p $O
timeout 900 python -m pytest tests/test_ppo_agent.py tests/test_graph_manager.py tests/test_checkpoint.py tests/test_signals_csv.py tests/test_preset_dropin.py -m gpu -q --tb=short -x 2>&1 | tail -8
for v in 0 1; do
RLX_PPO_MINIBATCH_GRAPHS=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c2_mbgraphs$v.json 2> $O/bench_c2_mbgraphs$v.err
python -c "
import json; d=json.loads(open('$O/bench_c2_mbgraphs$v.json').read().strip().splitlines()[-1]); print('c2 minibatch_graphs=$v', d['ms_per_step'], d['value'])"
done
