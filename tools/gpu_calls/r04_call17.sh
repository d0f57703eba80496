#!/bin/bash
# round 4, call 17: what clock does the shader run at under MFMA load, and how many cycles is a dependent fp32 MFMA?
set -u
O=gpurun_out/r04_call17
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('canary', float((x * 2).sum().item()))" > $O/canary.txt 2>&1
grep -q "canary 2097152.0" $O/canary.txt || { echo "BAD BOX: torch itself faults"; tail -3 $O/canary.txt; exit 7; }
timeout 120 python tools/probe_mfma_clock.py > $O/probe.txt 2>&1
cat $O/probe.txt | tail -8
rocm-smi --showclocks 2>/dev/null | head -20 > $O/smi.txt; tail -12 $O/smi.txt
