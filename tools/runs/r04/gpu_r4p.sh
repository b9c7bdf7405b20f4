# round 4, call P: the link turn ends when a call's copies are issued: three host-buffer callers
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "pipelin or pieces or cfg2_full_size" ) > gpurun_out/pytest_p.log 2>&1
tail -6 gpurun_out/pytest_p.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for r in 1 2; do
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-serving --no-cpu-baseline --soak-seconds 0 > gpurun_out/bench_p$r.json 2> gpurun_out/bench_p$r.err
python - <<PY
import json
d=[json.loads(l) for l in open('gpurun_out/bench_p$r.json') if l.startswith('{')][-1]
e=d['end_to_end']
print('run $r step', round(d['ms_per_step'],3), 'e2e', round(e['ms_per_step'],3), 'median', round(e['ms_per_step_median'],3), 'fresh', round(e['fresh_buffers_ms_per_step'],3), 'unsplit', round(e['unsplit_ms_per_step'],3), '3callers', e['three_callers'])
PY
done
