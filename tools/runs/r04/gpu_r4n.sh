# round 4, call N: k_dsa_modexp with LDS-DMA rows (BFTKV_DSA_DMA=1): the DSA tests, then cfg 3 A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
( time BFTKV_DSA_DMA=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_protocol.py -m gpu -x -q -k "dsa or golden or negative or cfg3 or mixed or text_mode or read_answers or two_phase" ) > gpurun_out/pytest_n.log 2>&1
tail -8 gpurun_out/pytest_n.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for v in 0 1 0 1; do
  BFTKV_DSA_DMA=$v timeout 400 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg3_dma$v.json 2> gpurun_out/bench_cfg3_dma$v.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('gpurun_out/bench_cfg3_dma$v.json') if l.startswith('{')][-1]
    print('dma=$v cfg3', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M/s', {k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k!='last_call' and k!='measured'}, 'single', {k:round(x,2) for k,x in d['kernel_ms']['last_call'].items()})
except Exception as ex:
    print('dma=$v failed', ex); print(open('gpurun_out/bench_cfg3_dma$v.err').read()[-800:])
PY
done
BFTKV_DSA_DMA=1 timeout 400 python bench.py --config 3 --steps 5 --warmup 2 --inflight 1 --cpu-budget 6 --soak-seconds 0 > gpurun_out/bench_cfg3_dma1_check.json 2>/dev/null
timeout 400 python bench.py --config 3 --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_cfg3_dma0_single.json 2>/dev/null
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_cfg3_dma1_check.json') if l.startswith('{')][-1]
print('dma=1 single-flight', round(d['ms_per_step'],3), d['kernel_ms']['last_call'], d['cpu_baseline']['gpu_verdicts_identical_to_cpu'], d['cpu_baseline']['read_answers_identical_to_oracle'])
d=[json.loads(l) for l in open('gpurun_out/bench_cfg3_dma0_single.json') if l.startswith('{')][-1]
print('dma=0 single-flight', round(d['ms_per_step'],3), d['kernel_ms']['last_call'])
PY
