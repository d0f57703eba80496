python tools/ac_fused_phases.py 2>&1 | grep -v amdgpu.ids | grep total
python -m pytest tests/test_ac_nets.py -m gpu -q --tb=short -p no:cacheprovider -x -k "td3" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -15
python tools/ac_fused_bench.py td3 2>&1 | grep -v "amdgpu.ids"
mkdir -p gpurun_out
python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; tail -c 600 gpurun_out/bench_c4.err; python - <<PY
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c4', d['value'], d['ms_per_step'], 'update_us', r.get('update_us'), 'calls', r.get('library_calls_per_update'))
PY
