#!/bin/bash
# round 4, call 34 (= call 9 on the final tree): wave-state counters of every kernel of one C2 update with the LDS-DMA ring (two --pmc passes over
# tools/ppo_update_once.py), and one plain bench line (box lottery: a slow-class box would give the round's slow-box line)
set -u
O=gpurun_out/r04_call34
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=1
for set in "$A" "$B"; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_$i -- python $R/tools/ppo_update_once.py > $R/$O/sq_$i.log 2>&1)
f=$(find /tmp/sq_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/sq_$i.csv
i=$((i+1))
done
python tools/pmc_wave_states.py $O/sq_1.csv $O/sq_2.csv $O/wave_states.json 2>&1 | tail -24
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'update_us', r['update_us_in_epoch_graph'])
print(d['box'])
PY
