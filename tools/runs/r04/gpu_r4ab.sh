# round 4, call AB: cfg 5 with its new defaults (16 hardware queues, 16 steps in flight), alone and inside the default line; the cfg-5 test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --config 5 --soak-seconds 0 --cpu-budget 6 > gpurun_out/bench_ab_cfg5.json 2> gpurun_out/bench_ab_cfg5.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_ab_cfg5.json') if l.startswith('{')][-1]
print('cfg5 alone', round(d['ms_per_step'],3), round(d['value']/1e6,3), 'M ops/s', d['config']['steps_in_flight'], d['steps'], round(d['int_mac']['frac'],3), d['cpu_baseline']['gpu_results_identical_to_cpu'], {k:round(v,2) for k,v in d['kernel_ms'].items()})
PY
( time timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k cfg5 ) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_ab.json 2> gpurun_out/bench_default_ab.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_ab.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', round(e['ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3))
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('steps'), v.get('identity'), v.get('error'), round(v.get('wall_s',0),1))
PY
