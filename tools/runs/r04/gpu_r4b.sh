# round 4, call B: the pipelined host-buffer path -- its tests, then the cfg-2 line for several piece counts
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or pieces or cfg2_full_size or two_keys" ) > gpurun_out/pytest_pipe.log 2>&1
tail -15 gpurun_out/pytest_pipe.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for p in 0 3 4 5 6 8; do
  BFTKV_HB_PIECES=$p timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_hb_p$p.json 2> gpurun_out/bench_hb_p$p.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_hb_p$p.json').read().strip().splitlines()[-1])
    print('pieces=$p', 'resident ms', round(d['ms_per_step'],3), 'end_to_end', json.dumps(d['end_to_end']))
except Exception as e:
    print('pieces=$p failed', e); print(open('gpurun_out/bench_hb_p$p.err').read()[-1500:])
PY
done
