#!/bin/bash
# round 4, call 1: the new Clipped-PPO tests (evaluation / forced reset on images, full-size C2 parity), the whole GPU
# suite, the C2 bench line with the in-update roofline + box block, and a kernel trace of the same command.
set -u
O=gpurun_out/r04_call1
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
# canary: some boxes of the pool fault on every kernel of known-good code (torch's own included) — stop at once on those
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 900 python -m pytest tests/test_ppo_eval_reset.py tests/test_ppo_full_size.py -m gpu -q -s --tb=short 2>&1 | tail -60 > $O/new_tests.txt
tail -30 $O/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=5 -x 2>&1 | tail -25 > $O/pytest.txt
tail -8 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 500 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
tail -3 $O/bench_c2.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'warm', r['frac_warm'], 'gemm_us', r['gemm_us_per_update'], r['gemm_us_per_update_warm'])
print(r['update_us_by_family'], r['update_us_sum_of_kernels'], r['update_us_in_epoch_graph'])
for k in r['update_kernels']: print('  %-70s %5.2f x %7.2f = %7.2f' % (k['kernel'][:70], k['launches_per_update'], k['avg_us'], k['us_per_update']))
print(d['box']); print(d['cpu_baseline']['by_threads'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
head -24 $O/c2_kernel_stats.csv | cut -c1-200
