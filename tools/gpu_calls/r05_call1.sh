#!/bin/bash
# round 5, call 1: the new driver-suite tests (C3 at full size, C2 at L = 1024, profiler overflow, overlap decision on RCCL),
# then the whole GPU suite, the device side of the 10-epoch loss-curve ensemble (8 seeds), and the bench lines the round
# starts from (C2; C2 at L = 1024; c1 / c3 / c4 / c5 with the thread-aware cpu_baseline)
set -u
O=gpurun_out/r05_call1
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_dqn_full_size.py tests/test_ppo_long_episodes.py tests/test_gemm.py::test_kernel_timer_overflow_is_an_error "tests/test_data_parallel_gpu.py::test_rccl_path_world_size_one" "tests/test_data_parallel_gpu.py::test_two_rank_ppo_iteration" -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | tail -60 > $O/pytest_new.txt
tail -40 $O/pytest_new.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_all.txt
tail -5 $O/pytest_all.txt
timeout 600 python tools/loss_curve_c2.py --side hip --hip-seeds 0,1,2,3,4,5,6,7 --dir gpurun_out/lc_ens10 --iterations 49 --epochs 10 --no-init 2>&1 | tail -8
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --episode-length 1024 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c2_L1024.json 2> $O/bench_c2_L1024.err
for w in c1 c3 c4 c5; do timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in c2 c2_L1024 c1 c3 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    r=d.get('roofline', {})
    print('$w', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('by_threads'))
    if '$w' == 'c2': print('   box', d.get('box'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-800:])
PY
done
