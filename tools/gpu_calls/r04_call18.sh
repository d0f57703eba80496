#!/bin/bash
# round 4, call 18: the register-pipelined slab step with the ring 2 / 3 / 4 slabs deep (a CU's burst of 12-16 global ->
# LDS requests takes ~900 cycles to land, tools/probe_mfma_clock.py: one step of cover is not enough any more)
set -u
O=gpurun_out/r04_call18
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for l in librlx.so ab/librlx_d3.so ab/librlx_d4.so; do
n=$(basename $l .so)
timeout 200 python tools/gemm_timeline.py --lib coach_amd/$l > $O/timeline_$n.txt 2>&1
echo "## $l"; tail -7 $O/timeline_$n.txt
done
timeout 500 python tools/ab_c2_pipeline.py 2 coach_amd/ab/librlx_prev.so:1 coach_amd/librlx.so:1 coach_amd/ab/librlx_d3.so:1 coach_amd/ab/librlx_d4.so:1 > $O/ab.txt 2>&1
tail -5 $O/ab.txt
