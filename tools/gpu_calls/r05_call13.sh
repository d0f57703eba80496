#!/bin/bash
# round 5, call 13: wave states of the large-tile GEMM on the 4096^3 calibration product
set -u
O=gpurun_out/r05_call13
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for b in 0 1; do python tools/gemm4096_probe.py --big $b 2>&1 | grep TFLOP; done
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=1
for set in "$A" "$B"; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/g_$i -- python $R/tools/gemm4096_probe.py > $R/$O/g_$i.log 2>&1)
f=$(find /tmp/g_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/g_$i.csv
i=$((i+1))
done
python - <<PY
import csv
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for f in ("$O/g_1.csv","$O/g_2.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_dma" in r["Kernel_Name"]:
            k=r["Kernel_Name"][28:80]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    wc=v["SQ_WAVE_CYCLES"] or 1
    print(k, {c: round(v[c]/wc,3) for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM")})
    mf=v["SQ_INSTS_MFMA"] or 1
    print("   per MFMA: valu %.2f lds %.2f vmem %.3f salu %.2f ; bank conflict frac %.3f" % (v["SQ_INSTS_VALU"]/mf, v["SQ_INSTS_LDS"]/mf, v["SQ_INSTS_VMEM"]/mf, v["SQ_INSTS_SALU"]/mf, v["SQ_LDS_BANK_CONFLICT"]/(v["SQ_LDS_IDX_ACTIVE"] or 1)))
PY
