#!/bin/bash
# round 6, call 42: full GPU suite of the tree with two image pairs per workgroup by default (C2) + the alignment pre-check
set -u
O=gpurun_out/r06_call42
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest.txt
tail -10 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json | cut -c1-400
