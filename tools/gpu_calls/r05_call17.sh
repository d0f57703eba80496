#!/bin/bash
# round 5, call 17: tile 8 of the in-launch conv1 as four 16 x 16 blocks (v_mfma_f32_16x16x4_f32): is the sum still the same chain?
set -u
O=gpurun_out/r05_call17
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_fused.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -40
python tools/conv23_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline_triple.txt
run() { # name, flags
  timeout 400 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('%-14s' % '$1', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'fused', d.get('box',{}).get('fused_conv_forward_in_update_us'))
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
