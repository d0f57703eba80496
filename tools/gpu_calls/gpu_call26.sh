#!/bin/bash
O=gpurun_out/r02_call26; mkdir -p $O
for w in c4 c5 c3; do timeout 300 python tools/update_host_gpu_split.py --workload $w 2>&1 | tail -1 | tee $O/split_$w.json | cut -c1-600; done
