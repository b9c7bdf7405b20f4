# round 4, call T: pipeline tests on the last library build; the driver's command with cfg 5 ahead of the heavy configs
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_c_harness.py -m gpu -x -q ) > gpurun_out/pytest_t.log 2>&1
tail -5 gpurun_out/pytest_t.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_t.json 2> gpurun_out/bench_default_t.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_t.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', round(e['ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3))
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('steps'), v.get('identity'), v.get('error'), round(v.get('wall_s',0),1))
PY
timeout 200 python bench.py --config 5 --steps 40 --warmup 4 --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg5_alone_t.json 2>/dev/null
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_cfg5_alone_t.json') if l.startswith('{')][-1]
print('cfg5 alone', round(d['ms_per_step'],3), d['config']['steps_in_flight'])
PY
