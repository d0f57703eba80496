#!/bin/bash
# round 3, call 2: CartPole (bit parity + the two golden tests with their printed outcome), then the full suite
set -u
O=gpurun_out/r03_call2
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_cartpole.py -m gpu -q -s --tb=short 2>&1 | tail -40 | tee $O/cartpole.txt
timeout 1500 python -m pytest -m gpu -q --tb=short --durations=8 -x 2>&1 | tail -40 > $O/pytest.txt
tail -14 $O/pytest.txt
