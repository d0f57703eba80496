#!/bin/bash
# round 4, call 29: heads inside the dense reduction on the acting / value paths too — PPO suites, bench line
set -u
O=gpurun_out/r04_call29
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1200 python -m pytest tests/test_nn.py tests/test_ppo_agent.py tests/test_ppo_full_size.py tests/test_ppo_eval_reset.py tests/test_reference_loop.py tests/test_cartpole.py tests/test_checkpoint.py tests/test_architecture.py tests/test_evaluation.py -m gpu -q > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'])
PY
done
