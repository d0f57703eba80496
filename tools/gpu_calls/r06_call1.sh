#!/bin/bash
# round 6, call 1: what kind of box is this (tools/box_info.py), the two fp32 MFMA shapes' issue rate vs dependent latency,
# per-dispatch histogram of the update's kernels, and this round's starting lines for C2 / C4 / C5 on the same box
set -u
O=gpurun_out/r06_call1
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/box_info.py > $O/box_info.json 2> $O/box_info.err
python - <<PY
import json
d=json.load(open('$O/box_info.json'))
print(json.dumps(d['summary'], indent=1)[:6000])
PY
timeout 200 python tools/probe_mfma_shapes.py > $O/mfma_shapes.txt 2>&1; cat $O/mfma_shapes.txt | grep -v "amdgpu.ids"
timeout 300 python tools/dispatch_histogram.py --updates 200 --json $O/dispatch_hist.json > $O/dispatch_hist.txt 2>&1; grep -v "amdgpu.ids" $O/dispatch_hist.txt | head -70
run() { # name, flags
  timeout 500 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s' % '$1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'calls', r.get('library_calls_per_update'), 'box gemm', d['box']['gemm_4096_fp32_TFLOPs'], 'conv', d['box'].get('fused_conv_forward_in_update_us'))
except Exception as e:
    print('$1', 'ERR', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
run c2 "--steps 10 --warmup 3"
run c4 "--workload c4 --steps 3 --warmup 1"
run c5 "--workload c5 --steps 3 --warmup 1"
run c3 "--workload c3 --steps 3 --warmup 1"
