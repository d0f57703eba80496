#!/bin/bash
# final evidence of a round: full GPU suite (no -x), smoke, the five bench lines, C2 / C3 kernel traces, PMC passes
# (traffic, MFMA busy cycles, wave states).  Writes gpurun_out/r03_final/; what is kept goes to profiles/r03_final_*.
set -u
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=8 2>&1 | tail -40 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do timeout 400 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in c2 c1 c3 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('traffic'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/prof_c3.log 2>&1)
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
for w in c4 c5; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/$O/prof_$w.log 2>&1)
f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/ppo_update_once.py > $R/$O/pmc_$c.log 2>&1)
f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_gemm_traffic.json 2>&1 | tail -2 | cut -c1-400
(cd /tmp && REPS=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_eager -- python $R/tools/ppo_update_once.py > $R/$O/kt_eager.log 2>&1)
f=$(find /tmp/kt_eager -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/eager_update_kernel_stats.csv
head -14 $O/c2_kernel_stats.csv | cut -c1-170
