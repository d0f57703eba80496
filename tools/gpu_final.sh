#!/bin/bash
# final evidence of a round: full GPU suite (no -x), smoke, the five bench lines (roofline + box + cpu_baseline), kernel
# traces of the bench commands, PMC passes on one eager update (traffic, MFMA busy cycles).  Writes gpurun_out/r06_final/;
# what is kept goes to profiles/r06_final_*.  Every step runs under its own timeout.
set -u
O=gpurun_out/r06_final5
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 400 python bench.py --shapes > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c3 c4 c5; do timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --force-dist --no-cpu-baseline --no-roofline > $O/bench_c2_force_dist.json 2> $O/bench_c2_force_dist.err
timeout 400 python bench.py --episode-length 1024 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c2_L1024.json 2> $O/bench_c2_L1024.err
timeout 300 python bench.py --fuse-conv 0 --no-cpu-baseline > $O/bench_c2_unfused.json 2> $O/bench_c2_unfused.err
timeout 300 python bench.py --fuse-conv 1 --no-cpu-baseline > $O/bench_c2_pair_only.json 2> $O/bench_c2_pair_only.err
python tools/conv23_timeline.py 2>&1 | grep -v amdgpu.ids > $O/conv23_timeline.txt
python tools/conv23_timeline.py --pair 2>&1 | grep -v amdgpu.ids >> $O/conv23_timeline.txt
for w in c2 c1 c3 c4 c5 c2_force_dist c2_L1024 c2_unfused c2_pair_only; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
    r=d.get('roofline', {})
    print('$w', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'warm', r.get('frac_warm'), 'update_us', r.get('update_us_in_epoch_graph', r.get('update_us')), 'traffic', r.get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('rccl'))
    if '$w' == 'c2': print('   box', d.get('box'))
except Exception as e:
    print('$w', 'ERR', e); print(open('$O/bench_$w.err').read()[-600:])
PY
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1)
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv
tail -1 $O/prof_c2.log | cut -c1-400 > $O/prof_c2_bench_line.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/prof_c3.log 2>&1)
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
for w in c4 c5; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/$O/prof_$w.log 2>&1)
f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/ppo_update_once.py > $R/$O/pmc_$c.log 2>&1)
f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_gemm_traffic.json 2>&1 | tail -2 | cut -c1-400
python tools/pmc_summary.py --mfma $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv $O/pmc_GRBM_GUI_ACTIVE.csv $O/pmc_mfma_util.json 2>&1 | head -3 | cut -c1-300
(cd /tmp && REPS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_eager -- python $R/tools/ppo_update_once.py > $R/$O/kt_eager.log 2>&1)
f=$(find /tmp/kt_eager -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/eager_update_kernel_stats.csv
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=1
for set in "$A" "$B"; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_$i -- python $R/tools/ppo_update_once.py > $R/$O/sq_$i.log 2>&1)
f=$(find /tmp/sq_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/sq_$i.csv
i=$((i+1))
done
python tools/pmc_wave_states.py $O/sq_1.csv $O/sq_2.csv $O/pmc_wave_states.json 2>&1 | tail -16
head -16 $O/c2_kernel_stats.csv | cut -c1-170
du -sh $O
