for m in "" "-DRLX_CHAIN_ABLATE_LOADS" "-DRLX_CHAIN_ABLATE_MATH"; do echo "== $m"; tools/microbench/build/rowchain_probe$m 400 300 | grep "G=100 reps=8\|G=  1 reps=1"; done
python tools/ac_fused_phases.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -x -k "td3" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -15
python tools/ac_fused_bench.py td3 2>&1 | grep -v "amdgpu.ids"
