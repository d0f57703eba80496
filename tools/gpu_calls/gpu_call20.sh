#!/bin/bash
set -u
O=gpurun_out/r02_call20
mkdir -p $O
timeout 1200 python -m pytest tests/test_ppo_agent.py tests/test_emulator_frontend.py tests/test_ppo_heads_fused.py tests/test_graph_manager.py tests/test_data_parallel_gpu.py tests/test_checkpoint.py tests/test_preset_dropin.py -m gpu -q --tb=short 2>&1 | tail -60 > $O/pytest.txt
tail -45 $O/pytest.txt | cut -c1-250
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['value'], d['roofline']['frac'])"
