#!/bin/bash
# same-box A/B: build_ab/librlx_new.so (fast-kernel epilogue loads hoisted) vs librlx_new2.so (+ split-K reduce epilogue loads hoisted); tests on new2
set -u
mkdir -p gpurun_out/ab2
for v in new new2 new new2; do
  cp build_ab/librlx_$v.so coach_amd/librlx.so
  timeout 200 python bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> gpurun_out/ab2/bench.txt 2>&1
done
cp build_ab/librlx_new2.so coach_amd/librlx.so
cat gpurun_out/ab2/bench.txt
timeout 300 python -m pytest tests/test_gemm.py tests/test_ppo_agent.py tests/test_dqn_agent.py tests/test_networks.py -q -x 2>&1 | tail -4
