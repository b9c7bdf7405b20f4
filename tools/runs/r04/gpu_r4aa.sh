# round 4, call AA: the host-buffer pipeline with 16 hardware queues
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --config 2 --steps 40 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_aa_$name.json 2> gpurun_out/bench_aa_$name.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_aa_$name.json') if l.startswith('{')][-1]
    e=d['end_to_end']; t=e['timeline_us']
    print('$name step', round(d['ms_per_step'],3), 'single', round(d['kernel_ms']['single_flight']['total'],3), 'e2e', round(e['ms_per_step'],3), 'median', round(e['ms_per_step_median'],3), 'fresh', round(e['fresh_buffers_ms_per_step'],3), 'ring', round(e['ring_ms_per_step'],3), 'unsplit', round(e['unsplit_ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3))
except Exception as ex:
    print('$name failed', ex); print(open('gpurun_out/bench_aa_$name.err').read()[-800:])
PY
}
run q4
run q16 GPU_MAX_HW_QUEUES=16
run q16_early GPU_MAX_HW_QUEUES=16 BFTKV_HB_EARLY_MIDS=1
run q16_p8 GPU_MAX_HW_QUEUES=16 BFTKV_HB_PIECES=8
run q16_early_p8 GPU_MAX_HW_QUEUES=16 BFTKV_HB_EARLY_MIDS=1 BFTKV_HB_PIECES=8
