#!/bin/bash
set -u
O=gpurun_out/r02_call2
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
timeout 120 ./tools/microbench/launch_cost > $O/launch_cost.txt 2>&1
cat $O/launch_cost.txt; tail -5 $O/pytest.txt
