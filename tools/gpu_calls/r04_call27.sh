#!/bin/bash
# round 4, call 27: Clipped PPO with whole-dataset passes of 2048 rows: the PPO suites, then the default bench line
set -u
O=gpurun_out/r04_call27
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1200 python -m pytest tests/test_ppo_agent.py tests/test_ppo_full_size.py tests/test_ppo_eval_reset.py tests/test_reference_loop.py tests/test_cartpole.py tests/test_checkpoint.py tests/test_data_parallel_gpu.py tests/test_graph_manager.py tests/test_preset_dropin.py -m gpu -q > $O/tests.txt 2>&1
tail -4 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r['update_us_in_epoch_graph'])
PY
