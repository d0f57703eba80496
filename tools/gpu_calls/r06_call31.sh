#!/bin/bash
# round 6, call 31: flake check of the RCCL-in-graph test with quiesce_before_capture (5 x), then the full GPU suite
set -u
O=gpurun_out/r06_call31
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -2; done | tee $O/flake.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -12 > $O/pytest.txt; tail -12 $O/pytest.txt
