#!/bin/bash
# round 5, call 22: V(s) / action probabilities recorded by the acting steps (ClippedPPOAgent.RECORD_WHILE_ACTING): the new
# test, every Clipped-PPO test of the suite, same-box A/B of the C2 line and of C2 at L = 1024
set -u
O=gpurun_out/r05_call22
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_record_acting.py tests/test_acting_fused.py tests/test_ppo_agent.py tests/test_ppo_full_size.py tests/test_ppo_long_episodes.py tests/test_ppo_eval_reset.py tests/test_checkpoint.py tests/test_data_parallel_gpu.py tests/test_graph_manager.py tests/test_preset_dropin.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -40 | tee $O/pytest.txt
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline --no-roofline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    print('%-16s' % '$1', d['value'], d['ms_per_step'])
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c2_record "--steps 10 --warmup 3"
run c2_passes "--steps 10 --warmup 3 --record-acting 0"
run L1024_record "--episode-length 1024 --steps 4 --warmup 2"
run L1024_passes "--episode-length 1024 --steps 4 --warmup 2 --record-acting 0"
run c2_record2 "--steps 10 --warmup 3"
run c2_passes2 "--steps 10 --warmup 3 --record-acting 0"
