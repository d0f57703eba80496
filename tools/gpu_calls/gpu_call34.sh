#!/bin/bash
O=gpurun_out/r02_call34; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python -X faulthandler -m pytest tests/test_agent_loops.py tests/test_ppo_agent.py tests/test_ac_nets.py -m gpu -v --tb=short -x -p no:cacheprovider > $O/run$i.txt 2>&1
  echo "run $i rc=$? : $(grep -c PASSED $O/run$i.txt) passed; last: $(grep -n '::' $O/run$i.txt | tail -1 | cut -c1-150)"
  grep -n "Fatal Python\|Memory access fault\|Aborted\|core dumped\|HIP error\|hipError\|Segmentation" $O/run$i.txt | head -5
done
