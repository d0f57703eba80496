#!/bin/bash
set -u
O=gpurun_out/r02_call9
mkdir -p $O
timeout 900 python -m pytest tests/test_preset_dropin.py tests/test_signals_csv.py tests/test_episodic_replay.py tests/test_agent_loops.py tests/test_graph_manager.py tests/test_evaluation.py tests/test_checkpoint.py -m gpu -q --tb=short --durations=20 2>&1 | tail -80 > $O/pytest.txt
tail -70 $O/pytest.txt
