#!/bin/bash
# round 6, call 43: per-workgroup timeline of the tiled GEMM launches of the C2 update (FC forward, FC dW + dX pair)
set -u
O=gpurun_out/r06_call43
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_timeline.txt | tail -60
