python tools/ac_fused_phases.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -x -k "td3" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5
python tools/ac_fused_bench.py td3 2>&1 | grep -v "amdgpu.ids"
