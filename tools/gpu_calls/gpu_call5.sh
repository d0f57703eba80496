#!/bin/bash
set -u
O=gpurun_out/r02_call5
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for mode in fused nofused; do
  if [ $mode = nofused ]; then export RLX_NO_FUSED_MLP=1; else unset RLX_NO_FUSED_MLP; fi
  cd /tmp && rm -rf /tmp/prof_c1 && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_c1 -- python $R/bench.py --workload c1 --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_$mode.log 2>&1
  cd $R
  find /tmp/prof_c1 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/c1_kernel_trace_$mode.csv
  find /tmp/prof_c1 -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $O/c1_memcopy_trace_$mode.csv
done
ls -la $O
