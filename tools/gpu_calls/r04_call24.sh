#!/bin/bash
# round 4, call 24: the image DQN's Q head (online + target copy) inside the last dense layer's reduction — bit-identity
# test, the DQN / PER suites, C3 line
set -u
O=gpurun_out/r04_call24
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_nn.py tests/test_dqn_agent.py tests/test_reference_loop.py tests/test_abi.py tests/test_checkpoint.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 300 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print('c3', d['value'], d['ms_per_step'], d['roofline'].get('update_us'))
PY
