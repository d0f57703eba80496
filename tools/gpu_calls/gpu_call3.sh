#!/bin/bash
set -u
O=gpurun_out/r02_call3
mkdir -p $O
timeout 600 python -m pytest tests/test_reference_loop.py "tests/test_dqn_agent.py" -m gpu -q --tb=short 2>&1 | tail -150 > $O/pytest.txt
tail -120 $O/pytest.txt
