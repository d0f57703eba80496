#!/bin/bash
# round 4, call 19: where does a slab step go?  The forward products' phase stamps with the MFMAs removed, with the
# operand requests removed, with both removed (timing experiment builds: results are wrong, only the stamps are read)
set -u
O=gpurun_out/r04_call19
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for l in librlx.so ab/librlx_nomfma.so ab/librlx_nodma.so ab/librlx_neither.so; do
n=$(basename $l .so)
timeout 200 python tools/gemm_timeline.py --lib coach_amd/$l > $O/timeline_$n.txt 2>&1
echo "## $l"; tail -7 $O/timeline_$n.txt | head -6
done
