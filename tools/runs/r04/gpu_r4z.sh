# round 4, call Z: cfg 5, hardware queues x steps in flight, finer
export TMPDIR=/tmp
mkdir -p gpurun_out
for q in 12 16 24 32; do
  for f in 6 8 12 16; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --config 5 --steps 48 --warmup 4 --inflight $f --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_z5_q${q}_f$f.json 2>/dev/null
    python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_z5_q${q}_f$f.json') if l.startswith('{')][-1]
    print('cfg5 queues=$q inflight=$f', round(d['ms_per_step'],3), round(d['int_mac']['frac'],3))
except Exception as e:
    print('cfg5 queues=$q inflight=$f failed', e)
PY
  done
done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --config 5 --steps 48 --warmup 4 --inflight 8 --cpu-budget 5 --soak-seconds 0 > gpurun_out/bench_z5_check.json 2>/dev/null
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_z5_check.json') if l.startswith('{')][-1]
print('check', round(d['ms_per_step'],3), d['cpu_baseline']['gpu_results_identical_to_cpu'])
PY
