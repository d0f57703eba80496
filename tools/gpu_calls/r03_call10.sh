#!/bin/bash
# round 3, call 10: tests touched since call 9, then the tile-threshold A/B on the C2 update
set -u
O=gpurun_out/r03_call10
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_signals_csv.py tests/test_emulator_frontend.py tests/test_episodic_replay.py tests/test_data_parallel_gpu.py tests/test_architecture.py tests/test_gemm.py -m gpu -q --tb=short 2>&1 | tail -25 | tee $O/tests.txt
timeout 900 python tools/ab_c2.py 3 2>/dev/null | tail -1 | tee $O/ab_c2.json
