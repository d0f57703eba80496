#!/bin/bash
# same-box A/B of two builds of librlx.so (build_ab/librlx_{old,new}.so): GEMM timeline of one PPO minibatch + a short C2 bench
set -u
mkdir -p gpurun_out/ab
for v in old new old new; do
  cp build_ab/librlx_$v.so coach_amd/librlx.so
  timeout 120 python tools/gemm_timeline.py > gpurun_out/ab/timeline_$v.txt 2>&1
  timeout 200 python bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> gpurun_out/ab/bench.txt 2>&1
done
cp build_ab/librlx_new.so coach_amd/librlx.so
cat gpurun_out/ab/bench.txt
tail -8 gpurun_out/ab/timeline_old.txt; tail -8 gpurun_out/ab/timeline_new.txt
