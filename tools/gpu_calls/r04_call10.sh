#!/bin/bash
# round 4, call 10: uint8 operands through the LDS-DMA ring (conv1 forward, conv1 dW) — GEMM / network / full-size PPO
# parity, then the C2 update timed with the previous build and this one in one process each on the same box
set -u
O=gpurun_out/r04_call10
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_gemm.py tests/test_nn.py tests/test_ppo_full_size.py tests/test_ppo_agent.py -m gpu -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout 300 python tools/ab_c2_pipeline.py 3 coach_amd/ab/librlx_prev.so:1 coach_amd/librlx.so:1 > $O/ab.txt 2>&1
tail -30 $O/ab.txt
