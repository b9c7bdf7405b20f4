# round 4, call G: uniform ~40 MB pieces, direct copies by default, device-side timeline of the pieces
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or pieces or cfg2_full_size or md5 or text_mode or golden" ) > gpurun_out/pytest_g.log 2>&1
tail -15 gpurun_out/pytest_g.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_g_$name.json 2> gpurun_out/bench_g_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_g_$name.json').read().strip().splitlines()[-1])
    e=d['end_to_end']
    print('$name step', round(d['ms_per_step'],3), 'single', round(d['kernel_ms']['single_flight']['total'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step') or k=='ms_per_step_median'}, '3callers', round(e['three_callers']['ms_per_call'],3))
    for nm in ('timeline_us','ring_timeline_us'):
        t=e[nm]
        print('  ',nm,'done', round(t['done_us']), 'copy drained', round(t['copy_stream_drained_us']), [ (round(p['ss_enqueued']), round(p['enqueued']), '|', round(p['gpu_start']), round(p['gpu_modexp_start']), round(p['gpu_modexp_end']), round(p['gpu_end'])) for p in t['per_piece_us']])
except Exception as ex:
    print('$name failed', ex); print(open('gpurun_out/bench_g_$name.err').read()[-1500:])
PY
}
run auto
run auto_pad16k BFTKV_HB_MODEXP_LDS_PAD=16384
run auto_noprio BFTKV_HB_COPY_NO_PRIORITY=1
run p4 BFTKV_HB_PIECES=4
run p6_pad16k BFTKV_HB_PIECES=6 BFTKV_HB_MODEXP_LDS_PAD=16384
run p8 BFTKV_HB_PIECES=8
