# round 4, call K: new tests (wire replay, fail-closed pipeline), the rocprofv3 passes of the resident cfg-2 step, the RCCL rehearsal
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_protocol.py -m gpu -x -q -k "pipelin or pieces or wire or cert_verify or server" ) > gpurun_out/pytest_gpu_k.log 2>&1
tail -12 gpurun_out/pytest_gpu_k.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
bash tools/profile_bench.sh r04_cfg2 2 5 > gpurun_out/profile_r04_cfg2.log 2>&1
tail -3 gpurun_out/profile_r04_cfg2.log
# 8-GPU debut rehearsal on one rank: real communicators (three per rank for cfg 2 / 3, one for cfg 4), every config in one process
( time BFTKV_FORCE_RCCL=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-serving --cpu-budget 4 ) > gpurun_out/bench_force_rccl.json 2> gpurun_out/bench_force_rccl.err
tail -3 gpurun_out/bench_force_rccl.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_force_rccl.json').read().strip().splitlines()[-1])
print('force-rccl cfg2', round(d['value']/1e6,1), 'M/s', d['ms_per_step'], d['allgather'], 'e2e', round(d['end_to_end']['ms_per_step'],3))
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('identity'), v.get('error'))
PY
