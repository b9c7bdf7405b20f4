# round 4, call S: final tree -- whole GPU suite (incl. the plain-C caller), smoke, the driver's command
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/pytest_gpu_s.log 2>&1
tail -12 gpurun_out/pytest_gpu_s.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_s.json 2> gpurun_out/bench_default_s.err
tail -4 gpurun_out/bench_default_s.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_s.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step')}, '3callers', e['three_callers'])
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('steps'), v.get('identity'), v.get('error'))
PY
