#!/bin/bash
# full GPU suite three times: is anything flaky?
O=gpurun_out/r02_call33; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -X faulthandler -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/run$i.txt 2>&1
  echo "run $i rc=$? : $(tail -1 $O/run$i.txt | cut -c1-120)"
  grep -n "Fatal Python\|Memory access fault\|Aborted\|core dumped\|HIP error\|hipError" $O/run$i.txt | head -5
done
