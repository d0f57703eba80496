#!/bin/bash
set -u
O=gpurun_out/r02_call9
mkdir -p $O
timeout 900 python -m pytest tests/test_emulator_frontend.py tests/test_signals_csv.py tests/test_filters.py -m gpu -q --tb=short --durations=8 2>&1 | tail -80 > $O/pytest.txt
tail -70 $O/pytest.txt
