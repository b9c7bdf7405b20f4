# round 4, call R: one copier against two (early midstates off), on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_r_$name.json 2> gpurun_out/bench_r_$name.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_r_$name.json') if l.startswith('{')][-1]
    e=d['end_to_end']; t=e['timeline_us']
    print('$name step', round(d['ms_per_step'],3), 'e2e', round(e['ms_per_step'],3), 'median', round(e['ms_per_step_median'],3), 'fresh', round(e['fresh_buffers_ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3),
          'copy drained', round(t['copy_stream_drained_us']), [(round(p['ss_enqueued']), round(p['gpu_modexp_start']), round(p['gpu_end'])) for p in t['per_piece_us']])
except Exception as ex:
    print('$name failed', ex); print(open('gpurun_out/bench_r_$name.err').read()[-800:])
PY
}
run c1
run c2 BFTKV_HB_COPIERS=2
run c1_again
run c2_p6 BFTKV_HB_COPIERS=2 BFTKV_HB_PIECES=6
run c1_p6 BFTKV_HB_PIECES=6
