# round 4, call L: cfg-5 steps-in-flight sweep; rocprofv3 passes for cfg 5 and cfg 3
export TMPDIR=/tmp
mkdir -p gpurun_out
for f in 1 2 3 4 5 6 8; do
  timeout 200 python bench.py --config 5 --steps 20 --warmup 4 --inflight $f --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg5_f$f.json 2> gpurun_out/bench_cfg5_f$f.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_cfg5_f$f.json') if l.startswith('{')][-1]
    print('inflight=$f', 'ms/step', round(d['ms_per_step'],3), 'ops/s', round(d['value']/1e6,3), 'M', {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'mac', round(d['int_mac']['frac'],3))
except Exception as ex:
    print('inflight=$f failed', ex); print(open('gpurun_out/bench_cfg5_f$f.err').read()[-800:])
PY
done
( time bash tools/profile_bench.sh r04_cfg5 5 5 ) > gpurun_out/profile_r04_cfg5.log 2>&1; tail -2 gpurun_out/profile_r04_cfg5.log
( time bash tools/profile_bench.sh r04_cfg3 3 3 ) > gpurun_out/profile_r04_cfg3.log 2>&1; tail -2 gpurun_out/profile_r04_cfg3.log
