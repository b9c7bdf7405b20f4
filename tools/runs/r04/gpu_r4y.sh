# round 4, call Y: more hardware queues (GPU_MAX_HW_QUEUES) for the steps / batches in flight of cfg 5 and cfg 2
export TMPDIR=/tmp
mkdir -p gpurun_out
for q in 4 8 16; do
  for f in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --config 5 --steps 40 --warmup 4 --inflight $f --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_y5_q${q}_f$f.json 2>/dev/null
    python - <<PY
import json
d=[json.loads(l) for l in open('gpurun_out/bench_y5_q${q}_f$f.json') if l.startswith('{')][-1]
print('cfg5 queues=$q inflight=$f', round(d['ms_per_step'],3))
PY
  done
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config 2 --steps 100 --warmup 5 --no-serving --no-cpu-baseline --no-end-to-end --soak-seconds 0 > gpurun_out/bench_y2_q$q.json 2>/dev/null
  python - <<PY
import json
d=[json.loads(l) for l in open('gpurun_out/bench_y2_q$q.json') if l.startswith('{')][-1]
print('cfg2 queues=$q', round(d['ms_per_step'],3), round(d['kernel_ms']['k_rsa_modexp'],3))
PY
done
