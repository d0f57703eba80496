#!/bin/bash
# round 4, call 13: heads kernel with the losses computed under the backward body's loads (+ one reduction tree for the
# three policy sums) vs the previous build; split cap 128 (conv1 dW: 4 x 125 workgroups instead of 4 x 62); 32 x 32 tiles
# for the conv2 / conv3 forward products (kw_min_tiles 390)
set -u
O=gpurun_out/r04_call13
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 600 python -m pytest tests/test_nn.py tests/test_explore_env_losses.py tests/test_ppo_agent.py tests/test_ppo_full_size.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 500 python tools/ab_c2_pipeline.py 2 coach_amd/ab/librlx_prev.so:1 coach_amd/librlx.so:1 coach_amd/librlx.so:1:192,192,-1,128 coach_amd/librlx.so:1:192,390,-1 > $O/ab.txt 2>&1
tail -30 $O/ab.txt
