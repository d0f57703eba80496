#!/bin/bash
# round 4, call 16: the slab step software-pipelined through registers (operands of slab s + 1 read, and the wait /
# barrier / DMA issue done, in the middle of slab s's MFMA chain) — GEMM / network / PPO parity, phase stamps, C2 A/B
set -u
O=gpurun_out/r04_call16
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_full_size.py tests/test_ppo_agent.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for l in ab/librlx_prev.so librlx.so; do
n=$(basename $l .so)
timeout 200 python tools/gemm_timeline.py --lib coach_amd/$l > $O/timeline_$n.txt 2>&1
echo "## $l"; tail -7 $O/timeline_$n.txt
done
timeout 400 python tools/ab_c2_pipeline.py 2 coach_amd/ab/librlx_prev.so:1 coach_amd/librlx.so:1 > $O/ab.txt 2>&1
tail -4 $O/ab.txt
