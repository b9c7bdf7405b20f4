# round 4, call U: the driver's command with the other configs in processes of their own
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_u.json 2> gpurun_out/bench_default_u.err
tail -4 gpurun_out/bench_default_u.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_u.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', round(e['ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3))
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('steps'), v.get('identity'), v.get('error'), round(v.get('wall_s',0),1), v.get('command'))
PY
