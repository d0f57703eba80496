#!/bin/bash
# round 4, call 15: per-product phase stamps of the forward GEMMs with the ring 2 (default), 3 and 4 slabs deep — is a
# deeper ring worth selecting per launch for the products that run one workgroup per CU anyway?
set -u
O=gpurun_out/r04_call15
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for l in librlx.so ab/librlx_d3.so ab/librlx_d4.so; do
n=$(basename $l .so)
timeout 200 python tools/gemm_timeline.py --lib coach_amd/$l > $O/timeline_$n.txt 2>&1
echo "## $l"; tail -8 $O/timeline_$n.txt
done
