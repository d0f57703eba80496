#!/bin/bash
# round 6, call 47: DDPG through the staged-rows record and eight updates per replay — parity (loops, reference-pinned loops), suite
set -u
O=gpurun_out/r06_call47
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_draws_ahead.py tests/test_agent_loops.py tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -20 > $O/pytest.txt
tail -6 $O/pytest.txt
