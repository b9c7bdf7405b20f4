# round 4, call C: certificate verification through the batcher (tests), then the occupancy experiment:
# k_rsa_modexp at 2 waves per SIMD (LDS pad) so that the small kernels of other calls / pieces co-reside
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_protocol.py tests/test_gpu_host_pipeline.py -m gpu -x -q ) > gpurun_out/pytest_c.log 2>&1
tail -15 gpurun_out/pytest_c.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 300 python bench.py --config 2 --steps 30 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 $EXTRA > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_$name.json').read().strip().splitlines()[-1])
    e=d['end_to_end']; k=d['kernel_ms']
    print('$name', 'step', round(d['ms_per_step'],3), 'modexp', round(k['k_rsa_modexp'],3), 'single', {a:round(b,3) for a,b in k['single_flight'].items()}, 'e2e', round(e['ms_per_step'],3), 'median', round(e['ms_per_step_median'],3), 'unsplit', round(e['unsplit_ms_per_step'],3), '3callers', round(e['three_callers']['ms_per_call'],3))
except Exception as ex:
    print('$name failed', ex); print(open('gpurun_out/bench_$name.err').read()[-1200:])
PY
}
EXTRA="--inflight 3" run i3_pad0 BFTKV_HB_PIECES=3
EXTRA="--inflight 3" run i3_pad16k BFTKV_HB_PIECES=3 BFTKV_MODEXP_LDS_PAD=16384 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 1" run i1_pad0 BFTKV_HB_PIECES=3
EXTRA="--inflight 1" run i1_pad16k BFTKV_HB_PIECES=3 BFTKV_MODEXP_LDS_PAD=16384 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 2" run i2_pad16k BFTKV_HB_PIECES=4 BFTKV_MODEXP_LDS_PAD=16384 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 3" run hbpad_p3 BFTKV_HB_PIECES=3 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 3" run hbpad_p4 BFTKV_HB_PIECES=4 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 3" run hbpad_p6 BFTKV_HB_PIECES=6 BFTKV_HB_MODEXP_LDS_PAD=16384
EXTRA="--inflight 3" run hbpad_p8 BFTKV_HB_PIECES=8 BFTKV_HB_MODEXP_LDS_PAD=16384
