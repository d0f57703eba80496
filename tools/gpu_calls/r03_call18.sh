#!/bin/bash
# round 3, call 18: wave-state counters of the update's kernels; tile-size thresholds A/B
set -u
O=gpurun_out/r03_call18
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=0
for set in "$A" "$B"; do
i=$((i+1))
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_$i -- python $R/tools/ppo_update_once.py > $R/$O/sq_$i.log 2>&1)
f=$(find /tmp/sq_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/sq_$i.csv
tail -2 $O/sq_$i.log
done
python tools/pmc_wave_states.py $O/sq_1.csv $O/sq_2.csv $O/wave_states.json 2>&1 | tail -24
timeout 600 python tools/ab_c2.py 3 2>/dev/null | tail -1 | tee $O/ab_c2.json
