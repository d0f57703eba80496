#!/bin/bash
# round 6, call 10: sac_backward_kernel with every operand of the policy branch requested up front: parity, phases, time
set -u
O=gpurun_out/r06_call10
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -k "sac" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/pytest_sac.txt; tail -30 $O/pytest_sac.txt
timeout 300 python tools/sac_fused_phases.py 2>&1 | grep -v "amdgpu.ids" > $O/sac_phases.txt; cat $O/sac_phases.txt
timeout 300 python tools/ac_fused_bench.py sac 2>&1 | grep -v "amdgpu.ids" > $O/ac_fused_bench_sac.txt; head -40 $O/ac_fused_bench_sac.txt
