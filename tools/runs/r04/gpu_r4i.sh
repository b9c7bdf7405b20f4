# round 4, call I: the whole GPU suite, then the driver's command
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/pytest_gpu_i.log 2>&1
tail -22 gpurun_out/pytest_gpu_i.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_i.json 2> gpurun_out/bench_default_i.err
tail -4 gpurun_out/bench_default_i.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default_i.json').read().strip().splitlines()[-1])
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step')}, '3callers', e['three_callers'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['threads'])
print('serving', {k:v for k,v in d.get('serving',{}).items() if k!='points'} if isinstance(d.get('serving'),dict) else None)
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('identity'), v.get('error'), round(v.get('wall_s',0),1))
PY
