#!/bin/bash
python tools/c1_breakdown.py 2>&1 | grep -v amdgpu.ids
RLX_NO_FUSED_MLP=1 python tools/c1_breakdown.py 2>&1 | grep -v amdgpu.ids
