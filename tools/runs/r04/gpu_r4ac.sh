# round 4, call AC: cfg 3 with more hardware queues / steps in flight
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  name=$1; shift; fl=$1; shift
  env "$@" timeout 400 python bench.py --config 3 --steps 20 --warmup 3 --inflight $fl --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_ac_$name.json 2>/dev/null
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_ac_$name.json') if l.startswith('{')][-1]
    print('$name', round(d['ms_per_step'],3), round(d['value']/1e6,2), round(d['int_mac']['frac'],3))
except Exception as e:
    print('$name failed', e)
PY
}
run q4_f3 3
run q16_f3 3 GPU_MAX_HW_QUEUES=16
run q16_f4 4 GPU_MAX_HW_QUEUES=16
run q16_f6 6 GPU_MAX_HW_QUEUES=16
