#!/bin/bash
# round 4, call 23: do the workgroups of a launch contend for the same weight lines?  Phase stamps with every workgroup
# starting at another slab of its K chunk (timing experiment build), with and without the MFMAs
set -u
O=gpurun_out/r04_call23
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
for l in librlx.so ab/librlx_rot.so ab/librlx_nomfma.so ab/librlx_rot_nomfma.so; do
n=$(basename $l .so)
timeout 200 python tools/gemm_timeline.py --lib coach_amd/$l > $O/timeline_$n.txt 2>&1
echo "## $l"; tail -7 $O/timeline_$n.txt | head -6
done
timeout 300 python tools/ab_c2_pipeline.py 2 coach_amd/librlx.so:1 coach_amd/ab/librlx_rot.so:1 > $O/ab.txt 2>&1
tail -3 $O/ab.txt
