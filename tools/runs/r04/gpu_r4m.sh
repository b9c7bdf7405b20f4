# round 4, call M: k_multiexp in one-wave blocks, steps-in-flight sweep; cfg 3 with the longer mailbox wait
export TMPDIR=/tmp
mkdir -p gpurun_out
for f in 1 2 3 4 6; do
  BFTKV_MULTIEXP_BLOCK=64 timeout 200 python bench.py --config 5 --steps 20 --warmup 4 --inflight $f --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg5b64_f$f.json 2> gpurun_out/bench_cfg5b64_f$f.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_cfg5b64_f$f.json') if l.startswith('{')][-1]
    print('block64 inflight=$f', 'ms/step', round(d['ms_per_step'],3), 'ops/s', round(d['value']/1e6,3), 'M', {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'mac', round(d['int_mac']['frac'],3))
except Exception as ex:
    print('inflight=$f failed', ex); print(open('gpurun_out/bench_cfg5b64_f$f.err').read()[-800:])
PY
done
timeout 300 python bench.py --config 5 --steps 20 --warmup 4 --inflight 4 --cpu-budget 4 --soak-seconds 0 > gpurun_out/bench_cfg5_ref.json 2>/dev/null
BFTKV_MULTIEXP_BLOCK=64 timeout 300 python bench.py --config 5 --steps 20 --warmup 4 --inflight 4 --cpu-budget 4 --soak-seconds 0 > gpurun_out/bench_cfg5b64_check.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_cfg5_ref','bench_cfg5b64_check'):
    d=[json.loads(l) for l in open('gpurun_out/%s.json'%n) if l.startswith('{')][-1]
    print(n, round(d['ms_per_step'],3), d['cpu_baseline']['gpu_results_identical_to_cpu'])
PY
timeout 400 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg3_m.json 2> gpurun_out/bench_cfg3_m.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_cfg3_m.json') if l.startswith('{')][-1]
print('cfg3', round(d['ms_per_step'],3), round(d['value']/1e6,2), d['kernel_ms'])
PY
