# round 4, call O: the whole GPU suite and the driver's command on the final tree
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/pytest_gpu_o.log 2>&1
tail -14 gpurun_out/pytest_gpu_o.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_o.json 2> gpurun_out/bench_default_o.err
tail -4 gpurun_out/bench_default_o.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_o.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step')}, '3callers', e['three_callers'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], 'int_mac', d['int_mac']['frac'])
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('identity'), v.get('error'), (v.get('roofline') or {}).get('traffic_source'))
PY
