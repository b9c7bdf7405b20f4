# round 4, call D: host-buffer pipeline with the copy plan / small last piece / pinned ring: tests, then cfg 2 with its timeline
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or pieces or cfg2_full_size" ) > gpurun_out/pytest_d.log 2>&1
tail -15 gpurun_out/pytest_d.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for p in 0 3 4 6; do
  BFTKV_HB_PIECES=$p timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_d_p$p.json 2> gpurun_out/bench_d_p$p.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_d_p$p.json').read().strip().splitlines()[-1])
    e=d['end_to_end']
    print('pieces=$p step', round(d['ms_per_step'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k.endswith('ms_per_step') or k=='ms_per_step_median'}, '3callers', round(e['three_callers']['ms_per_call'],3))
    print('   timeline', json.dumps(e['timeline_us']))
except Exception as ex:
    print('pieces=$p failed', ex); print(open('gpurun_out/bench_d_p$p.err').read()[-1500:])
PY
done
