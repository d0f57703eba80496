#!/bin/bash
# round 5, call 15: conv1 in front of the fused pair (rlx_conv123_forward): bit-equality with the tiled launches, the suites
# that run through it, C2 / C3 / C2-at-L=1024 with the triple (2), the pair (1)
set -u
O=gpurun_out/r05_call15
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_fused.py tests/test_abi.py tests/test_ppo_full_size.py tests/test_dqn_full_size.py tests/test_dqn_agent.py tests/test_ppo_agent.py tests/test_reference_image_loops.py tests/test_nn.py tests/test_architecture.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -40
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('%-14s' % '$1', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'conv1', d.get('box',{}).get('conv1_forward_in_update_us'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c2_triple "--steps 10 --warmup 3"
run c2_pair "--steps 10 --warmup 3 --fuse-conv 1"
run c3_triple "--workload c3"
run c3_pair "--workload c3 --fuse-conv 1"
run L1024_triple "--episode-length 1024 --steps 4 --warmup 2"
run L1024_pair "--episode-length 1024 --steps 4 --warmup 2 --fuse-conv 1"
run c2_triple2 "--steps 10 --warmup 3"
run c2_pair2 "--steps 10 --warmup 3 --fuse-conv 1"
