#!/bin/bash
set -u
O=gpurun_out/r02_call7
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short --durations=25 2>&1 | tail -150 > $O/pytest.txt
tail -100 $O/pytest.txt
