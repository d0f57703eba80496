#!/bin/bash
# round 5, call 20: an acting step's small launches merged (softmax + sample; reward filter + episode totals + columns)
set -u
O=gpurun_out/r05_call20
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_acting_fused.py tests/test_ppo_agent.py tests/test_ppo_full_size.py tests/test_ppo_long_episodes.py tests/test_ppo_eval_reset.py tests/test_cartpole.py tests/test_data_parallel_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -25
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
run L1024_merged "--episode-length 1024 --steps 4 --warmup 2"
run L1024_separate "--episode-length 1024 --steps 4 --warmup 2 --fuse-acting 0"
run c2_merged "--steps 10 --warmup 3"
run c2_separate "--steps 10 --warmup 3 --fuse-acting 0"
run L1024_merged2 "--episode-length 1024 --steps 4 --warmup 2"
run L1024_separate2 "--episode-length 1024 --steps 4 --warmup 2 --fuse-acting 0"
