#!/bin/bash
# round 4, call 20: the heads' forward inside the last dense layer's split-K reduction (rlx_gemm_desc.row_heads) —
# bit-identity tests, PPO parity, C2 update with the switch off / on
set -u
O=gpurun_out/r04_call20
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_nn.py tests/test_gemm.py tests/test_ppo_full_size.py tests/test_ppo_agent.py tests/test_ppo_eval_reset.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 400 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1:192,192,-1:net.HEADS_FORWARD_WITH_TORSO=0 coach_amd/librlx.so:1:192,192,-1:net.HEADS_FORWARD_WITH_TORSO=1 > $O/ab.txt 2>&1
tail -4 $O/ab.txt
