# round 4, call Q: the signature-body fuzz through the pipelined host-buffer path; the driver's command on the final tree
export TMPDIR=/tmp
mkdir -p gpurun_out
for seed in 4 5; do
  ( time BFTKV_FUZZ_PIECES=3 timeout 600 python tools/fuzz_bodies.py 12 $seed ) > gpurun_out/fuzz_pieces_seed$seed.log 2>&1
  grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/fuzz_pieces_seed$seed.log | tail -4
done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default_q.json 2> gpurun_out/bench_default_q.err
tail -4 gpurun_out/bench_default_q.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_default_q.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('cfg2', d['value'], d['ms_per_step'], 'e2e', {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step')}, '3callers', e['three_callers'])
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('identity'), v.get('error'))
PY
